#!/usr/bin/env python3
"""Generates tests/golden/eikonal_vectors_big.npz: a few `mt_eikonal` / `eikonal` ruptures at the resolution BASELINE config 4
runs at (effective dt 0.5 s: a 25 m fine grid, 10^5 .. 10^6 points per fast-marching solve), discretised by the reference's OWN
Fortran modules (oracle/_ref, compiled unmodified; CRUST2.0 tables under /root/reference/aux).  Round 6 rewrote the product's
march and grid passes for exactly these sizes; the small cases of eikonal_vectors.npz (50 .. 100 m grids) do not reach the heap
depths and front lengths of a 30 km rupture.  Run in the dev container:

    python tests/golden/make_golden_eikonal_big.py

Inputs (parameters, looked-up profiles, constraints) and the reference's centroid tables only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_eikonal as small  # noqa: E402   (runs the small generator's module body: its helpers; it rewrites its own file identically)

out = {"rupture_profile": small.out["rupture_profile"], "origin_profile": small.out["origin_profile"]}
cases = [
    # type, limit, [time, north, east, depth, moment, strike, dip, (slip-rake,) bord-shift-x, -y, radius, nukl-x, -y, rel-velocity, ...]
    (5, 0.0, [0.0, 0.0, 0.0, 11000.0, 1.0, 91.0, 90.0, 0.0, 0.0, 15000.0, 2000.0, 0.0, 0.9, 1e20, 2e19, -3e19, 1e19, 5e19, -2e19, 1.0]),
    (5, 16000.0, [0.3, 400.0, -800.0, 9000.0, 1.0, 30.0, 60.0, 500.0, -300.0, 12000.0, -5000.0, 1500.0, 0.75, 1e18, -2e18, 1e18, 3e17, 0.0, 5e17, 0.0]),
    (4, 0.0, [-0.2, 100.0, 50.0, 14000.0, 3e19, -120.0, 45.0, 170.0, 0.0, 0.0, 8000.0, 0.0, 0.0, 1.0, 2.0]),
    (5, 12000.0, [0.0, 0.0, 0.0, 7000.0, 1.0, 200.0, 80.0, 0.0, 0.0, 13000.0, 12500.0, -2000.0, 0.6, 0.0, 1e18, -1e18, 0.0, 0.0, 0.0, 0.5]),
]
k = 0
for st, limit, p in cases:
    nc, cent, mo, ri, cp, cn = small.ref_eikonal(st, p, 0.5, small.lat_rad, small.lon_rad, limit)
    assert nc > 0, (k, nc)
    out["e%d_type" % k] = np.array(st)
    out["e%d_params" % k] = np.asarray(p, np.float32)
    out["e%d_edt" % k] = np.float32(0.5)
    out["e%d_limit" % k] = np.float32(limit)
    out["e%d_con" % k] = np.stack([cp, cn])
    out["e%d_cent" % k] = cent
    out["e%d_mr" % k] = np.array([mo, ri], np.float32)
    k += 1
out["n"] = np.array(k)
path = os.path.join(HERE, "eikonal_vectors_big.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes;", k, "cases; centroids:", [len(out["e%d_cent" % i]) for i in range(k)])
