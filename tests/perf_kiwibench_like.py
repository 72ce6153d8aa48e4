#!/usr/bin/env python3
"""The reference's own benchmark loop (benchmark/kiwibench.py:95-153) on this engine: ten `ned` receivers 3-4 km north of
the source, a near-field database sampled at 0.1 s / 50 m (synthetic stand-in for `gfdb_build benchdb 1 200 200 10 0.1 50 50
50 0`), synthetic references of a bilateral point source, NO misfit tapers, `floating_l1norm` with a shift range of
+-1 s, and a sweep over 3610 strikes; reports misfit evaluations per second ("MPS" there) for the device, and for the CPU
oracle on a sample of the same sources (a fresh oracle engine per source = the device's un-tapered semantics)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kiwi_amd import Engine, synthetic  # noqa: E402

EARTH_R = 6371000.0
gf = synthetic.make_gfdb(nx=100, nz=120, ng=10, L=256, dt=0.1, dx=50.0, dz=50.0, firstx=50.0, firstz=0.0,
                         center=60.0, width=40.0, vel=2300.0)
olat, olon = 30.0, 70.0
dist = np.linspace(3000.0, 4000.0, 10)
lat = olat + np.degrees(dist / EARTH_R)
lon = np.full(10, olon)
comps = ["ned"] * 10
base = np.array([0, 0, 0, 5000, 1e12, 91, 87, 164, 0, 0, 0, 0, 2500, 0.2], np.float32)
strikes = np.linspace(0., 360., 3610)
trials = np.tile(base, (len(strikes), 1))
trials[:, 5] = strikes

p = Engine(0)
p.set_database(gf["dt"], gf["dx"], gf["dz"], gf["firstx"], gf["firstz"], gf["data"], gf["first"], gf["nsamp"])
p.set_effective_dt(0.1)
p.set_local_interpolation("bilinear")
p.set_receivers(lat, lon, np.zeros(10, np.float32), comps)
p.set_source_location(olat, olon, 0.0)
p.set_source_params("bilateral", base[None, :])                 # set_synthetic_reference
p.set_keep_synthetics(1)
p.eval()
refs = {(ir + 1, k + 1): p.get_synthetics(0, ir + 1, k + 1, 1) for ir in range(10) for k in range(3)}
p.set_keep_synthetics(0)
for (ir, k), (lo, d) in refs.items():
    p.set_ref_seismogram(ir, k, lo, d)
p.set_floating_shiftrange(0, -1.0, 1.0)
p.set_misfit_method("floating_l1norm")
p.set_source_params("bilateral", trials)
for _ in range(3):
    p.eval()
p.sync()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    p.set_source_params("bilateral", trials)
    p.eval()
    m, n, g = p.get_misfits()
    sh = p.get_floating_shifts()
dt_dev = (time.perf_counter() - t0) / reps
print("device: %d strikes in %.2f ms -> %.0f misfit evaluations / s (set_source_params + eval + get_misfits + get_floating_shifts)"
      % (len(strikes), dt_dev * 1e3, len(strikes) / dt_dev))

from oracle import ko  # noqa: E402
nx, nz, ng, L = gf["data"].shape
db = ko.Gfdb(nx, nz, ng, gf["dt"], gf["dx"], gf["dz"], gf["firstx"], gf["firstz"])
for ix in range(nx):
    for iz in range(nz):
        for ig in range(ng):
            db.set_trace(ix + 1, iz + 1, ig + 1, int(gf["first"][ix, iz, ig]), gf["data"][ix, iz, ig])
sample = list(range(0, len(strikes), 181))
worst, t_cpu = 0.0, 0.0
for i in sample:
    e = ko.Engine(db)
    e.set_receivers(lat, lon, np.zeros(10, np.float32), comps)
    e.set_source_location(olat, olon, 0.0)
    e.set_effective_dt(0.1)
    e.set_interpolation(True)
    e.set_nthreads(min(os.cpu_count() or 1, 10))
    for (ir, k), (lo, d) in refs.items():
        e.set_reference(ir, k, lo, d)
    for ir in range(10):
        e.set_floating_shiftrange(ir + 1, -10, 10)
    e.set_misfit_method(8)
    t1 = time.perf_counter()
    e.set_source_params(1, trials[i])
    om, on, og = e.get_misfits()
    t_cpu += time.perf_counter() - t1
    worst = max(worst, abs(g[i] - og) / max(og, 1e-30))
    assert np.array_equal(sh[i], np.array([e.floating_shift(ir + 1) * gf["dt"] for ir in range(10)], np.float32)), i
    e.close()
db.close()
print("oracle (CPU, %d threads): %.1f misfit evaluations / s on %d of the strikes; worst relative difference of the global misfit %.2e, "
      "floating shifts identical" % (min(os.cpu_count() or 1, 10), len(sample) / t_cpu, len(sample), worst))
