"""Per-CU event trace of accumulate_multi_kernel ("Where a group's time goes"): runs one evaluation with the
instrumented library mktrace.sh builds (wave 0 of every workgroup on CU 0 of XCC 0 writes s_memtime at its phase boundaries) and
prints the phase medians.   KIWI_HIP_LIB=$PWD/kiwi_amd/libkiwi_hip_T0.so python3 profiles/microbench/cu_trace/trace.py cfg3 4096
VARIANT=noarith,nolds,nocoef SFX=T5 bash profiles/microbench/cu_trace/mktrace.sh builds the text-removal variants of the apply."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from kiwi_amd import synthetic, lib as klib
wlname = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
tag = sys.argv[3] if len(sys.argv) > 3 else ""
wl = synthetic.workload(wlname, batch, 0)
p, gf, recv, refs, tapers, ncent = bench.setup_product(0, wl, 4096)
L = klib.load()
ar = os.environ.get("KIWI_HIP_ARITH", "exact")
fn = getattr(L, "kiwi_hip_exp_trace_" + ar)
NB = 3000
out = (C.c_ulonglong * (NB * 32 * 6))()
n = C.c_uint(0)
p.eval(); p.sync()
fn(out, C.byref(n), 1)
p.kernel_ms()
p.eval(); p.sync()
ms, la = p.kernel_ms()
fn(out, C.byref(n), 1)
N = min(n.value, NB)
v = np.frombuffer(out, np.uint64).reshape(NB, 32, 6)[:N].astype(np.int64)
np.save("gpurun_out/trace_%s_%s%s.npy" % (wlname, ar, tag), v)
acc = ms[1] / la[1]
hdr = v[:, 30]
ng = hdr[:, 3]
ok = ng > 0
print(wlname, ar, tag, "acc ms %.1f" % acc, "workgroups on the CU", N, "groups per wg", np.median(ng[ok]))
span = hdr[ok, 2].max() - hdr[ok, 0].min()
tpu = span / (acc * 1e3)
life = (hdr[ok, 2] - hdr[ok, 0])
print("  ticks per us %.0f   slot occupancy (sum of lifetimes / 3 span) %.2f" % (tpu, life.sum() / (3.0 * span)))
g = np.concatenate([v[i, :min(ng[i], 30)] for i in np.nonzero(ok)[0]])
top, build, bar, app = g[:, 1] - g[:, 0], g[:, 2] - g[:, 1], g[:, 3] - g[:, 2], g[:, 4] - g[:, 3]
print("  per group (median ticks): top %.0f  build %.0f  barrier2 %.0f  apply %.0f   sum %.0f" % (np.median(top), np.median(build), np.median(bar), np.median(app), np.median(top) + np.median(build) + np.median(bar) + np.median(app)))
first = np.array([v[i, 0, 0] for i in np.nonzero(ok)[0]])
print("  per workgroup (median ticks): entry->first group %.0f   groups %.0f   epilogue %.0f   lifetime %.0f" % (np.median(first - hdr[ok, 0]), np.median(hdr[ok, 1] - first), np.median(hdr[ok, 2] - hdr[ok, 1]), np.median(life)))
# first group of a workgroup against the others
isfirst = np.concatenate([np.arange(min(ng[i], 30)) == 0 for i in np.nonzero(ok)[0]])
print("  first group: top %.0f build %.0f   others: top %.0f build %.0f" % (np.median(top[isfirst]), np.median(build[isfirst]), np.median(top[~isfirst]), np.median(build[~isfirst])))
# how many of the CU's (three) workgroups are in their apply at the same time, and does an apply take longer next to another one?
a_s, a_e = np.sort(g[:, 3]), np.sort(g[:, 4])
lo, hi = g[:, 1].min(), g[:, 4].max()
ts = np.linspace(lo + 0.2 * (hi - lo), lo + 0.8 * (hi - lo), 20000)
napp = np.searchsorted(a_s, ts) - np.searchsorted(a_e, ts)
print("  workgroups of the CU in their apply at the same time: " + "  ".join("%d: %.0f %%" % (k, 100 * np.mean(napp == k)) for k in range(4)))
na0 = np.searchsorted(a_s, g[:, 3], side="left") - np.searchsorted(a_e, g[:, 3], side="right")
print("  apply cycles when it starts next to k other applies: " + "  ".join("%d: %.0f" % (k, np.median(app[na0 == k])) for k in range(3) if np.any(na0 == k)))
