# Profile collection (run on the GPU box through gpurun):   TAG=r05 bash profiles/collect.sh
#   WORKLOADS="cfg3 cfg4"  (default: every bench workload)     ARITHS="exact fused"  (default: both arithmetic contracts)
# Per (workload, contract): one kernel-trace pass (durations, --stats) and the PMC passes bench.py's `roofline` block quotes --
# memory-side traffic of the L2s (FETCH_SIZE, WRITE_SIZE: separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots"), L2
# requests, and the issue-slot counters of the accumulate kernels.  Counters only with --pmc (no trace domains in the same run).
# Summarise afterwards in the repo (needs git for the commit id):   python profiles/summarize.py r05 gpurun_out/r05
# (the summary's kernel_sources_sha256 is computed THERE from the comment-stripped sources, and it records the hashes of the raw files)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${TAG:-r05}
O=gpurun_out/$TAG
mkdir -p $O
for w in ${WORKLOADS:-cfg3 cfg3-100pt cfg3-scatter cfg3-w256 cfg3-w600 cfg3-ng8 cfg3-static cfg2 cfg4 cfg4-nukl cfg5 cfg5-td cfg3-bigdb cfg3-bigdb4 cfg3-bigdb4-ordered}; do
 for ar in ${ARITHS:-exact fused}; do
  case "$w:$ar" in cfg4-nukl:fused|cfg3-bigdb*:fused) continue;; esac       # (the HBM regime and the nucleation sweep are measured once)
  k=$w; [ $ar = fused ] && k=$w@fused
  export KIWI_HIP_ARITH=$ar
  B="python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-also"
  rocprofv3 --kernel-trace --stats -d $O/kt_$k -o runc --output-format csv -- $B > $O/bench_$k.json 2> $O/kt_$k.log
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$k -o runc --output-format csv -- $B > /dev/null 2> $O/fetch_$k.log
  rocprofv3 --pmc WRITE_SIZE -d $O/write_$k -o runc --output-format csv -- $B > /dev/null 2> $O/write_$k.log
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/l2_$k -o runc --output-format csv -- $B > /dev/null 2> $O/l2_$k.log
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU -d $O/sq2_$k -o runc --output-format csv -- $B > /dev/null 2> $O/sq2_$k.log
  if [ -n "$SQ_EXTRA" ]; then
   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $O/sq_$k -o runc --output-format csv -- $B > /dev/null 2> $O/sq_$k.log
  fi
  # keep the merge small: the kernel-trace CSV of a run is the big file
  find $O/kt_$k -name '*_kernel_trace.csv' -size +8M -delete
 done
done
# the pure-read ceiling of THIS box, next to the HBM-regime workload (profiles/microbench/hbm_read.hip): wall clock, then the counters
# on a known number of bytes
if [ -z "$NO_MICROBENCH" ]; then
 (cd profiles/microbench && hipcc --offload-arch=gfx950 -O3 -o hbm_read hbm_read.hip)
 ./profiles/microbench/hbm_read 4 20 > $O/hbm_read_4g.json 2> $O/hbm_read.err
 ./profiles/microbench/hbm_read 8 10 > $O/hbm_read_8g.json 2>> $O/hbm_read.err
 for c in "FETCH_SIZE" "TCC_MISS_sum TCC_HIT_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  d=$O/hbmread_pmc_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c -d $d -o run --output-format csv -- ./profiles/microbench/hbm_read 4 3 > /dev/null 2> $d.log
 done
fi
