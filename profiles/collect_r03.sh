# Round-3 profile collection (run on the GPU box through gpurun):  bash profiles/collect_r03.sh
# Per workload: one kernel-trace pass (durations, --stats) and the PMC passes the bench's `roofline` block quotes --
# fabric traffic (FETCH_SIZE, WRITE_SIZE: separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots"), L2 requests, and
# the issue-slot counters of the accumulate kernels.  Counters only with --pmc (no trace domains in the same run).
# Summarise afterwards in the repo (needs git for the commit id):  python profiles/summarize_r03.py r03 gpurun_out/r03
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
for w in ${WORKLOADS:-cfg3 cfg3-100pt cfg3-scatter cfg3-bigdb cfg2 cfg4 cfg5 cfg5-td}; do
  B="python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-also"
  rocprofv3 --kernel-trace --stats -d $O/kt_$w -o runc --output-format csv -- $B > $O/bench_$w.json 2> $O/kt_$w.log
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$w -o runc --output-format csv -- $B > /dev/null 2> $O/fetch_$w.log
  rocprofv3 --pmc WRITE_SIZE -d $O/write_$w -o runc --output-format csv -- $B > /dev/null 2> $O/write_$w.log
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/l2_$w -o runc --output-format csv -- $B > /dev/null 2> $O/l2_$w.log
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $O/sq_$w -o runc --output-format csv -- $B > /dev/null 2> $O/sq_$w.log
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU -d $O/sq2_$w -o runc --output-format csv -- $B > /dev/null 2> $O/sq2_$w.log
  # keep the merge small: the kernel-trace CSV of a run is the big file
  find $O/kt_$w -name '*_kernel_trace.csv' -size +8M -delete
done
