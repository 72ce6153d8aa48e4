# kernel-trace statistics of the other BASELINE configurations (one pass each, no counters)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in cfg2 cfg4 cfg5; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/r1b_kt_$w -o runc --output-format csv -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r1b_bench_$w.json 2> gpurun_out/r1b_kt_$w.log
done
ls gpurun_out/r1b_kt_cfg*
