#!/usr/bin/env python3
"""Generates tests/golden/eikonal_vectors.npz from the reference's OWN Fortran modules (oracle/_ref:
source_eikonal, source_mt_eikonal, eikonal, heap, geometry, crust2x2 compiled unmodified) and the
CRUST2.0 tables under /root/reference/aux.  Run in the dev container:

    python tests/golden/make_golden_eikonal.py

The file holds inputs (parameters, the looked-up 1-D crust profiles, constraints) and the reference's
centroid tables only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ko  # noqa: E402

R = ko.ref()
assert R is not None, "build oracle/_ref first (make -C oracle ref)"
AUX = "/root/reference/aux/crust2x2"
fp = ko._fp


def ref_profile(lat, lon):
    vp, vs, rho, th = (np.zeros(8, np.float32), np.zeros(8, np.float32), np.zeros(8, np.float32), np.zeros(7, np.float32))
    ok = C.c_int()
    d = AUX.encode()
    R.ref_crust_profile(d, C.c_int(len(d)), C.c_double(lat), C.c_double(lon), fp(vp), fp(vs), fp(rho), fp(th), C.byref(ok))
    assert ok.value == 1
    return np.concatenate([vp, vs, rho, th])


def ref_eikonal(st, params, edt, lat_rad, lon_rad, limit=0.0):
    params = np.ascontiguousarray(params, np.float32)
    maxc = 200000
    cent = np.zeros((maxc, 10), np.float32)
    nc, gs = C.c_int(), (C.c_int * 2)()
    mo, ri = C.c_float(), C.c_float()
    cp, cn = np.zeros((2, 3), np.float32), np.zeros((2, 3), np.float32)
    R.ref_discretize_eikonal(C.c_int(st), C.c_int(len(params)), fp(params), C.c_float(edt), C.c_double(lat_rad),
                             C.c_double(lon_rad), C.c_float(limit), C.c_int(maxc), C.byref(nc), fp(cent), C.byref(mo),
                             C.byref(ri), gs, fp(cp), fp(cn))
    return nc.value, cent[:max(nc.value, 0)].copy(), mo.value, ri.value, cp, cn


out = {}
rng = np.random.default_rng(20261003)
lat, lon = 40.75, 29.86                       # Izmit; any CRUST2.0 cell does
lat_rad, lon_rad = float(np.deg2rad(lat)), float(np.deg2rad(lon))
# the two look-ups of the reference: radians for the rupture speeds (source_eikonal.f90:472),
# degrees for the thickness (parameterized_source.f90:215)
out["rupture_profile"] = ref_profile(lat_rad, lon_rad)
out["origin_profile"] = ref_profile(lat, lon)
k = 0
for st in (4, 5):
    for limit in (0.0, 0.0, 0.0, 0.0, 12000.0, 20000.0):
        for _ in range(50):
            common = [rng.uniform(-1, 1), rng.uniform(-3e3, 3e3), rng.uniform(-3e3, 3e3), rng.uniform(4e3, 2.0e4)]
            strike, dip = rng.uniform(-180, 180), rng.uniform(5, 90)
            bord = [rng.uniform(-2e3, 2e3), rng.uniform(-2e3, 2e3), rng.uniform(2e3, 9e3)]
            nukl = [rng.uniform(-1, 1) * 0.5 * bord[2], rng.uniform(-1, 1) * 0.3 * bord[2]]
            relv, rise = rng.uniform(0.6, 1.0), rng.uniform(0, 3)
            if st == 5:
                p = common + [1.0, strike, dip] + bord + nukl + [relv] + list(rng.standard_normal(6) * 1e18) + [rise]
            else:
                p = common + [10 ** rng.uniform(17, 19), strike, dip, rng.uniform(-180, 180)] + bord + nukl + [relv, rise]
            edt = float(rng.choice([1.0, 2.0]))
            nc, cent, mo, ri, cp, cn = ref_eikonal(st, p, edt, lat_rad, lon_rad, limit)
            if nc > 0:
                break
        else:
            raise SystemExit("no valid case found")
        out["e%d_type" % k] = np.array(st)
        out["e%d_params" % k] = np.asarray(p, np.float32)
        out["e%d_edt" % k] = np.float32(edt)
        out["e%d_limit" % k] = np.float32(limit)
        out["e%d_con" % k] = np.stack([cp, cn])
        out["e%d_cent" % k] = cent
        out["e%d_mr" % k] = np.array([mo, ri], np.float32)
        k += 1
# failure cases: rupture plane entirely above the surface constraint, nucleation point outside the circle
p = [0, 0, 0, 500., 1.0, 30., 0.5, 0, 0, 300., 0, 0, 0.9, 1e18, -1e18, 0, 0, 0, 0, 1.0]
nc, *_ = ref_eikonal(5, p, 1.0, lat_rad, lon_rad)
assert nc == -1, nc
out["fail_empty_params"] = np.asarray(p, np.float32)
out["n"] = np.array(k)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eikonal_vectors.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes;", k, "cases; centroids:", [len(out["e%d_cent" % i]) for i in range(k)])
