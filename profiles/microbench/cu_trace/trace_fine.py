"""Round 5's finer per-CU event trace of accumulate_multi_kernel ("What bounds the accumulate kernels"): one evaluation with
the library mktrace_fine.sh builds; prints the median / mean cycles of the nine phases of a centroid group for waves 0 and 3.
  KIWI_HIP_LIB=$PWD/kiwi_amd/libkiwi_hip_trace.so python profiles/microbench/cu_trace/trace_fine.py cfg3 4096"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from kiwi_amd import synthetic, lib as klib
wlname = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
wl = synthetic.workload(wlname, batch, 0)
p, gf, recv, refs, tapers, ncent = bench.setup_product(0, wl, 4096)
L = klib.load()
ar = os.environ.get("KIWI_HIP_ARITH", "exact")
fn = getattr(L, "kiwi_hip_exp_trace_" + ar)
NB, NG, NT = 1500, 24, 12
out = (C.c_ulonglong * (NB * 2 * NG * NT))()
n = C.c_uint(0)
p.eval(); p.sync()
fn(out, C.byref(n), 1)
p.kernel_ms()
p.eval(); p.sync()
ms, la = p.kernel_ms()
fn(out, C.byref(n), 1)
N = min(n.value, NB)
v = np.frombuffer(out, np.uint64).reshape(NB, 2, NG, NT)[:N].astype(np.int64)
np.save("gpurun_out/tracefine_%s_%s.npy" % (wlname, ar), v)
print(wlname, ar, "acc ms %.1f" % (ms[1] / la[1]), "workgroups traced", N)
names = ["top", "issue loads", "barrier1 (tiles free)", "shadow", "blend + LDS writes", "halo", "next records", "barrier2", "apply"]
for w, wn in ((0, "wave 0"), (1, "wave 3 (halo wave)")):
    g = v[:, w].reshape(-1, NT)
    g = g[(g[:, 0] > 0) & (g[:, 9] > 0)]
    g = g[1:]                       # (skip first)
    d = np.diff(g[:, :10], axis=1)
    print("  %s: groups %d; median ticks per phase: " % (wn, len(g)) + "  ".join("%s %d" % (names[i], np.median(d[:, i])) for i in range(9)) + "   sum %d" % np.median(g[:, 9] - g[:, 0]))
    print("      mean: " + "  ".join("%s %d" % (names[i], np.mean(d[:, i])) for i in range(9)))
# skew between waves 0 and 3 at barrier arrival
a = v[:, 0].reshape(-1, NT); b = v[:, 1].reshape(-1, NT)
ok = (a[:, 0] > 0) & (b[:, 0] > 0) & (a[:, 9] > 0) & (b[:, 9] > 0)
print("  arrival at barrier1: wave3 - wave0 median %d (abs median %d); at barrier2: %d (abs %d)" % (np.median(b[ok, 2] - a[ok, 2]), np.median(np.abs(b[ok, 2] - a[ok, 2])), np.median(b[ok, 7] - a[ok, 7]), np.median(np.abs(b[ok, 7] - a[ok, 7]))))
