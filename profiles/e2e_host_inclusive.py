"""Host-inclusive rate of the boundary (DESIGN.md "PCIe / host-inclusive rate"): per batch `set_source_params` (host
discretiser + H2D of the centroid tables), `eval`, `get_misfits` (D2H of all misfits), against the resident-input figure
bench.py reports.  Usage: python profiles/e2e_host_inclusive.py [cfg3|cfg4|cfg3-100pt ...] [batch]
For the eikonal workloads the host discretiser is a fast-marching solve per trial source (SURVEY 8f-4 asks what share of a
step it is before an on-GPU discretiser is considered).  Third line: the same trial sources as a list of `nb` batches
through ONE call of kiwi_hip_misfits_for_params (host discretiser of one piece while the device evaluates another)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # noqa: E402
from kiwi_amd import synthetic                 # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else {"cfg4": 128, "cfg4-nukl": 128}.get(name, 256)
wl = synthetic.workload(name, batch, 0)
p, gf, recv, refs, tapers, ncent = bench.setup_product(0, wl, 4096)
tr = wl["trials"]
reps = 3 if name.startswith("cfg4") else 10
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(reps):
        p.set_source_params(wl["sourcetype"], tr)     # host discretisation + upload of centroid tables
        p.eval()
        m, n, g = p.get_misfits()                     # kernels + download of all misfits
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        p.set_source_params(wl["sourcetype"], tr)
    ds = (time.perf_counter() - t0) / reps
    print("%s: end-to-end per %d-source batch: %.2f ms (%.0f evals/s); of which set_source_params (host discretiser, "
          "%d threads at most + H2D) %.2f ms = %.1f %%" % (name, batch, dt * 1e3, batch / dt, p.L.kiwi_hip_effective_cpus(), ds * 1e3,
                                                         100 * ds / dt))
# a trial list of nb batches: one after the other, and through the overlapped call
import ctypes                                  # noqa: E402
import numpy as np                             # noqa: E402


def cache_stats(reset=0):
    """(hits, misses) of the eikonal discretiser's solve cache; reset 3: counters and stored solves"""
    h, m = ctypes.c_longlong(0), ctypes.c_longlong(0)
    p.L.kiwi_hip_eikonal_cache_stats(ctypes.byref(h), ctypes.byref(m), reset)
    return h.value, m.value


nb = 4
big = synthetic.workload(name, batch * nb, 0)["trials"]
for rep in range(2):
    t0 = time.perf_counter()
    seq = []
    for k in range(nb):
        p.set_source_params(wl["sourcetype"], big[k * batch:(k + 1) * batch])
        p.eval()
        seq.append(p.get_misfits())
    dseq = time.perf_counter() - t0
    cache_stats(3)                                  # the sweep starts with an empty solve cache
    t0 = time.perf_counter()
    m, n, g, st = p.misfits_for_params(wl["sourcetype"], big, batch)
    dpipe = time.perf_counter() - t0
    hits, misses = cache_stats()
    same = np.array_equal(m, np.concatenate([x[0] for x in seq])) and np.array_equal(g, np.concatenate([x[2] for x in seq]))
    print("%s: %d trial sources in pieces of %d: one after the other %.1f ms (%.0f evals/s); one overlapped call %.1f ms "
          "(%.0f evals/s); identical results: %s; fast-marching solves of that call: %d computed, %d taken from the cache"
          % (name, len(big), batch, dseq * 1e3, len(big) / dseq, dpipe * 1e3, len(big) / dpipe, same, misses, hits))
