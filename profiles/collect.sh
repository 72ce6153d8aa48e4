set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/r1b_bench.json 2> gpurun_out/r1b_bench.err
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d gpurun_out/r1b_kt -o runc --output-format csv -- $B > gpurun_out/r1b_bench_under_rocprof.json 2> gpurun_out/r1b_kt.log
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r1b_fetch -o runc --output-format csv -- $B > /dev/null 2> gpurun_out/r1b_fetch.log
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r1b_write -o runc --output-format csv -- $B > /dev/null 2> gpurun_out/r1b_write.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/r1b_l2 -o runc --output-format csv -- $B > /dev/null 2> gpurun_out/r1b_l2.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d gpurun_out/r1b_sq -o runc --output-format csv -- $B > /dev/null 2> gpurun_out/r1b_sq.log
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d gpurun_out/r1b_sq2 -o runc --output-format csv -- $B > /dev/null 2> gpurun_out/r1b_sq2.log
ls gpurun_out/r1b_*
