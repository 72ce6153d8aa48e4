import sys, time, numpy as np
sys.path.insert(0, '.')
import bench
from kiwi_amd import synthetic
wl = synthetic.workload('cfg3', 256, 0)
p, gf, recv, refs, tapers, ncent = bench.setup_product(0, wl, 4096)
tr = wl['trials']
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        p.set_source_params('bilateral', tr)      # host discretisation + upload of centroid tables
        p.eval(); m, n, g = p.get_misfits()       # kernels + download of all misfits
    dt = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        p.set_source_params('bilateral', tr)
    ds = (time.perf_counter() - t0) / 10
    print("end-to-end per 256-source batch: %.2f ms (%.0f evals/s); of which set_source_params (host discretiser + H2D) %.2f ms" % (dt * 1e3, 256 / dt, ds * 1e3))
