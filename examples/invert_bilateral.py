#!/usr/bin/env python3
"""End-to-end example on synthetic data: what a `python/tunguska` user does with the reference, on this engine.

  1. a Green's function database and a receiver ring (synthetic stand-ins, kiwi_amd/synthetic.py);
  2. "observed" traces = synthetics of a known bilateral rupture + noise, set as references with misfit tapers;
  3. grid search over strike x dip x slip-rake (MisfitGrid: one batched device evaluation for the whole grid),
     bootstrap over the receivers;
  4. Levenberg-Marquardt refinement from the best grid node (one batched evaluation per Jacobian).

Run on a machine with an MI355X:  python examples/invert_bilateral.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kiwi_amd import Engine, synthetic, gridsearch, lm  # noqa: E402


def main(nrec=24, L=1024, noise=0.05, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    gf = synthetic.make_gfdb(nx=96, nz=6, L=L)
    lat, lon, depth, comps, dist = synthetic.make_receivers(nrec, dmin=120e3, dspan=300e3)
    e = Engine(0)
    e.set_database(gf["dt"], gf["dx"], gf["dz"], gf["firstx"], gf["firstz"], gf["data"], gf["first"], gf["nsamp"])
    e.set_effective_dt(0.5)
    e.set_local_interpolation("bilinear")
    e.set_receivers(lat, lon, depth, comps)
    e.set_source_location(40.0, 30.0, 0.0)
    true = np.array(synthetic.TRUE_BILAT, np.float32)
    # observed data: synthetics of the true source (no references needed for that) + band-limited noise
    e.set_source_params("bilateral", true[None, :])
    e.set_keep_synthetics(1)
    e.eval()
    dt = gf["dt"]
    for ir in range(nrec):
        for k in range(3):
            lo, d = e.get_synthetics(0, ir + 1, k + 1, 1)
            n = rng.standard_normal(len(d)).astype(np.float32)
            n = np.convolve(n, np.hanning(21) / np.hanning(21).sum(), "same")
            e.set_ref_seismogram(ir + 1, k + 1, lo, d + noise * np.abs(d).max() * n)
        e.set_misfit_taper(ir + 1, *synthetic.full_taper(lo, len(d), dt, ramp=8.0))
    e.set_keep_synthetics(0)
    e.set_misfit_method("l2norm")
    # grid search
    start = true.copy()
    start[5:8] += [17.0, -11.0, 23.0]
    grid = gridsearch.MisfitGrid("bilateral", start, param_ranges=[("strike", 60, 120, 3), ("dip", 60, 90, 3),
                                                                   ("slip-rake", 130, 200, 5)])
    t0 = time.perf_counter()
    grid.compute(e)
    grid.postprocess(bootstrap_iterations=200, rng=rng, outer_norm="l2norm")
    t_grid = time.perf_counter() - t0
    best = grid.best_source
    # refinement
    t0 = time.perf_counter()
    res = lm.minimize_lm(e, "bilateral", best, ["strike", "dip", "slip-rake", "depth"])
    t_lm = time.perf_counter() - t0
    if verbose:
        print("grid: %d sources in %.3f s (incl. %d bootstrap draws); best strike/dip/rake %.0f/%.0f/%.0f, misfit %.4f"
              % (len(grid.sources), t_grid, len(grid.bootstrap_sources), best[5], best[6], best[7], grid.get_best_misfit()))
        for name, st in grid.stats.items():
            print("   %-10s best %7.2f   bootstrap mean %7.2f +- %.2f" % (name, st.best, st.mean, st.std))
        print("LM:   %d forward evaluations (Jacobians batched), %.3f s; strike/dip/rake/depth %.2f/%.2f/%.2f/%.0f, misfit %.4f"
              % (res.iterations, t_lm, res.best[5], res.best[6], res.best[7], res.best[3], res.misfit))
        print("true: strike/dip/rake/depth %.2f/%.2f/%.2f/%.0f" % (true[5], true[6], true[7], true[3]))
    e.close()
    return true, grid, res


if __name__ == "__main__":
    main()
