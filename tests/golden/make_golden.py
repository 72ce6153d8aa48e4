#!/usr/bin/env python3
"""Generates tests/golden/ref_vectors.npz from the reference's OWN Fortran modules (oracle/_ref,
built unmodified from /root/reference by `make -C oracle ref`).  Run in the dev container:

    python tests/golden/make_golden.py

The file holds inputs and the reference's outputs only (data, no reference source).  Seeds fixed."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ko  # noqa: E402

R = ko.ref()
assert R is not None, "build oracle/_ref first (make -C oracle ref)"
fp, ip = ko._fp, ko._ip
out = {}
rng = np.random.default_rng(20261002)


def rnd_trace(n):
    d = rng.standard_normal(n).astype(np.float32)
    for _ in range(rng.integers(0, 3)):
        a = rng.integers(0, n)
        d[a:a + rng.integers(6, 12)] = 0
    if rng.random() < 0.4:
        d[-rng.integers(1, 5):] = 0
    return d


# ---- trace_multiply_add (sparse_trace.f90:597): 60 cases, mode 2 (fractional shift) mostly
cases = []
for i in range(60):
    n = int(rng.integers(3, 90))
    t = rnd_trace(n)
    tlo = int(rng.integers(-20, 40))
    has_s = rng.random() < 0.7
    sn = int(rng.integers(1, 120))
    s = rng.standard_normal(sn).astype(np.float32)
    slo = int(rng.integers(-30, 60))
    factor = np.float32(rng.standard_normal() * 5)
    mode = int(rng.choice([0, 1, 2, 2, 2]))
    ish = int(rng.integers(-10, 10))
    rsh = np.float32(rng.uniform(-10, 10))
    o = np.zeros(2048, np.float32)
    olo, on = C.c_int(), C.c_int()
    R.ref_multiply_add(C.c_int(tlo), C.c_int(n), fp(t), C.c_int(int(has_s)), C.c_int(slo), C.c_int(sn), fp(s),
                       C.c_float(factor), C.c_int(mode), C.c_int(ish), C.c_float(rsh), C.c_int(2048),
                       C.byref(olo), C.byref(on), fp(o))
    out["ma%d_t" % i] = t
    out["ma%d_s" % i] = s
    out["ma%d_p" % i] = np.array([tlo, int(has_s), slo, mode, ish, olo.value], np.int32)
    out["ma%d_f" % i] = np.array([factor, rsh], np.float32)
    out["ma%d_o" % i] = o[:on.value].copy()
out["ma_n"] = np.array(60)

# ---- gfdb blend (gfdb.f90:944-949 on trace_multiply_add_nogrow): 20 cases
for i in range(20):
    nmax = 100
    lo = rng.integers(0, 20, 4).astype(np.int32)
    n = rng.integers(5, nmax, 4).astype(np.int32)
    data = np.zeros((4, nmax), np.float32)
    for k in range(4):
        data[k, :n[k]] = rnd_trace(int(n[k]))
    dix, diz = np.float32(rng.random()), np.float32(rng.random())
    o = np.zeros(512, np.float32)
    olo, on = C.c_int(), C.c_int()
    R.ref_blend4(ip(lo), ip(n), C.c_int(nmax), fp(data), C.c_float(dix), C.c_float(diz), C.byref(olo), C.byref(on), fp(o))
    out["bl%d_lo" % i] = lo
    out["bl%d_n" % i] = n
    out["bl%d_d" % i] = data
    out["bl%d_w" % i] = np.array([dix, diz], np.float32)
    out["bl%d_o" % i] = o[:on.value].copy()
    out["bl%d_olo" % i] = np.array(olo.value)
out["bl_n"] = np.array(20)

# ---- orthodrome (orthodrome.f90:77,193,245): 200 receiver/centroid pairs
n = 200
alat = np.deg2rad(rng.uniform(-70, 70, n))
alon = np.deg2rad(rng.uniform(-180, 180, n))
blat = alat + rng.uniform(-0.1, 0.1, n)
blon = alon + rng.uniform(-0.1, 0.1, n)
dx = rng.uniform(-2e4, 2e4, n).astype(np.float32)
dy = rng.uniform(-2e4, 2e4, n).astype(np.float32)
dx[:5] = 0
dy[:5] = 0
res = np.zeros((n, 6))
for i in range(n):
    a, b, d = C.c_double(), C.c_double(), C.c_double()
    R.ref_azibazi_dist(C.c_double(alat[i]), C.c_double(alon[i]), C.c_double(blat[i]), C.c_double(blon[i]),
                       C.byref(a), C.byref(b), C.byref(d))
    na, nb, nd = C.c_double(), C.c_double(), C.c_double()
    R.ref_approx_differential_azidist(C.c_float(dx[i]), C.c_float(dy[i]), a, b, d, C.byref(na), C.byref(nb), C.byref(nd))
    res[i] = [a.value, b.value, d.value, na.value, nb.value, nd.value]
out["or_in"] = np.stack([alat, alon, blat, blon], 1)
out["or_dxy"] = np.stack([dx, dy], 1)
out["or_out"] = res

# ---- source discretisers (source_bilat.f90:241, source_circular.f90:235, source_moment_tensor.f90:205, source_point_lp.f90:237)
k = 0
for st, plist in ((1, [[0., 0., 0., 10000., 1e20, 91., 87., 164., 0., 4000., 2000., 4000., 3000., 2.],
                       [0.5, 1000., -2000., 8000., 3e19, -40., 45., 90., 30., 3000., 0., 2000., 2500., 0.5],
                       [0., 0., 0., 5000., 1e18, 10., 60., -20., 0., 0., 0., 0., 3000., 1.]]),
                  (2, [[0., 0., 0., 10000., 5e19, 80., 70., 100., 3000., 3000., 1.5],
                       [1., 500., 500., 6000., 1e19, 200., 30., -60., 1200., 2800., 0.3]]),
                  (6, [[0., 0., 0., 10000., 1e18, -1e18, 0., 3e17, 0., -2e17, 1.],
                       [2., 100., 200., 3000., 1., 2., -3., 0.5, 0.25, -0.75, 3.2]]),
                  (3, [[0., 0., 0., 10000., 7e18, 1., 0., -1., 1., 1., 1., 20., 10.],          # point_lp, source_point_lp.f90:237-337
                       [1.5, -300., 800., 4000., 2e17, 0.3, -0.8, 0.5, 0.1, -0.2, 0.7, 7.5, 3.0]])):
    for p in plist:
        for edt in (0.5, 1.0):
            par = np.array(p, np.float32)
            cent = np.zeros((20000, 10), np.float32)
            nc, gs = C.c_int(), (C.c_int * 3)()
            mo, rt = C.c_float(), C.c_float()
            R.ref_discretize(C.c_int(st), C.c_int(len(par)), fp(par), C.c_float(edt), C.c_int(20000), C.byref(nc),
                             fp(cent), C.byref(mo), C.byref(rt), gs)
            out["ds%d_in" % k] = np.concatenate([[st, edt], par]).astype(np.float32)
            out["ds%d_c" % k] = cent[:nc.value].copy()
            out["ds%d_mr" % k] = np.array([mo.value, rt.value], np.float32)
            k += 1
out["ds_n"] = np.array(k)

# ---- P / T axes of bilateral sources (psm_update_dep_params_bilat, source_bilat.f90:216-239)
pa_rng = np.random.default_rng(77)
pa_in = np.zeros((40, 14), np.float32)
pa_in[:, 5] = pa_rng.uniform(-360, 360, 40)
pa_in[:, 6] = pa_rng.uniform(0, 90, 40)
pa_in[:, 7] = pa_rng.uniform(-180, 180, 40)
pa_in[:4, 5:8] = [[0, 90, 0], [90, 45, 90], [30, 0, -90], [180, 90, 180]]
pa_out = np.zeros((40, 4), np.float32)
for i in range(40):
    pax, tax = np.zeros(2, np.float32), np.zeros(2, np.float32)
    R.ref_principal_axes_bilat(fp(pa_in[i]), fp(pax), fp(tax))
    pa_out[i] = [pax[0], pax[1], tax[0], tax[1]]
out["pa_in"], out["pa_out"] = pa_in, pa_out

# ---- strip_fold (sparse_trace.f90:379): 20 cases
for i in range(20):
    n = int(rng.integers(5, 80))
    d = rng.standard_normal(n).astype(np.float32)
    if rng.random() < 0.5:
        d[-int(rng.integers(1, 8)):] = d[-1]
    if rng.random() < 0.5:
        d[:int(rng.integers(1, 6))] = 0
    lo = int(rng.integers(-10, 10))
    ns = 1 + 2 * int(rng.integers(0, 5))
    sh = (np.arange(ns) - (ns - 1) / 2).astype(np.float32)
    am = rng.random(ns).astype(np.float32)
    am /= am.sum()
    o = np.zeros(512, np.float32)
    olo, on = C.c_int(), C.c_int()
    R.ref_strip_fold(C.c_int(lo), C.c_int(n), fp(d), C.c_int(ns), fp(sh), fp(am), C.c_int(512), C.byref(olo),
                     C.byref(on), fp(o))
    out["sf%d_d" % i] = d
    out["sf%d_p" % i] = np.array([lo, olo.value], np.int32)
    out["sf%d_sh" % i] = sh
    out["sf%d_am" % i] = am
    out["sf%d_o" % i] = o[:on.value].copy()
out["sf_n"] = np.array(20)

# ---- taper weights (piecewise_linear_function.f90:195, ip_cos): 10 cases
for i in range(10):
    x = np.sort(rng.uniform(0, 60, 4)).astype(np.float32)
    y = np.array([0, 1, 1, 0], np.float32)
    lo, hi = int(rng.integers(-5, 20)), int(rng.integers(100, 140))
    dx_ = np.float32(rng.choice([0.5, 0.25, 1.0]))
    arr = np.ones(hi - lo + 1, np.float32)
    R.ref_plf_taper_array_r(C.c_int(4), fp(x), fp(y), C.c_int(lo), C.c_int(hi), fp(arr), C.c_float(dx_), C.c_int(0))
    out["tp%d_x" % i] = x
    out["tp%d_p" % i] = np.array([lo, hi], np.int32)
    out["tp%d_dx" % i] = np.array(dx_)
    out["tp%d_o" % i] = arr
out["tp_n"] = np.array(10)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vectors.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
