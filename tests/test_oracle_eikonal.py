"""Eikonal sources (source_eikonal.f90, source_mt_eikonal.f90, eikonal.f90, heap.f90, geometry.f90):
oracle restatement against the reference's own modules, bit for bit.  Needs oracle/_ref and the
CRUST2.0 tables under /root/reference/aux (dev container only); golden vectors cover other boxes."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import ko

R = ko.LazyRef()
AUX = "/root/reference/aux/crust2x2"
pytestmark = pytest.mark.skipif(not ko.ref_available() or not os.path.isdir(AUX), reason="needs oracle/_ref and the reference's aux data")
fp = ko._fp


def ref_profile(lat, lon):
    vp, vs, rho, th = (np.zeros(8, np.float32), np.zeros(8, np.float32), np.zeros(8, np.float32), np.zeros(7, np.float32))
    ok = C.c_int()
    d = AUX.encode()
    R.ref_crust_profile(d, C.c_int(len(d)), C.c_double(lat), C.c_double(lon), fp(vp), fp(vs), fp(rho), fp(th), C.byref(ok))
    assert ok.value == 1
    return ko.crust_profile(vp, vs, rho, th)


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
    return nc.value, cent[:max(nc.value, 0)].copy(), mo.value, ri.value, (gs[0], gs[1]), cp, cn


def test_fast_marching_bitexact():
    """Speed fields as psm_make_*_grid builds them (source_mt_eikonal.f90:487-519): layered in the
    down-dip direction, an outside region at half the minimum speed.  (With arbitrary rough random
    fields the reference's own updateheap can walk off its heap array and crash; not exercised.)"""
    rng = np.random.default_rng(11)
    L = ko.lib()
    for _ in range(40):
        nx, ny = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        layers = np.sort(rng.choice([2100., 2600., 3200., 3500., 3900.], 3)) * np.float32(rng.uniform(0.6, 1.0))
        speed = np.zeros((ny, nx), np.float32)
        for iy in range(ny):
            speed[iy, :] = layers[min(2, (3 * iy) // max(ny, 1))]
        yy, xx = np.mgrid[0:ny, 0:nx]
        outside = (xx - nx / 2.) ** 2 / max(nx / 2., 1) ** 2 + (yy - ny / 2.) ** 2 / max(ny / 2., 1) ** 2 > 1.0
        speed[outside] = np.float32(speed.min() * np.float32(0.5))
        origin = rng.uniform(-5000, 0, 2).astype(np.float32)
        delta = rng.uniform(100, 900, 2).astype(np.float32)
        ip = (origin + rng.uniform(0.3, 0.7, 2) * delta * [nx, ny]).astype(np.float32)
        t1 = np.zeros((ny, nx), np.float32)
        t2 = np.zeros((ny, nx), np.float32)
        R.ref_eikonal_fmm(C.c_int(nx), C.c_int(ny), fp(speed), fp(origin), fp(delta), fp(ip), fp(t1))
        L.ko_eikonal_solver_fmm(fp(speed), C.c_int(nx), C.c_int(ny), fp(origin), fp(delta), fp(ip), fp(t2))
        assert np.array_equal(t1.view(np.uint32), t2.view(np.uint32))


def test_default_constraints_from_crust():
    lat, lon = 40.75, 29.86
    prof = ref_profile(lat, lon)
    nc, cent, mo, ri, gs, cp, cn = ref_eikonal(5, [0, 0, 0, 10000, 1, 100, 60, 0, 0, 6000, 0, 0, 0.9, 1e18, -1e18, 0, 2e17, 0, 0, 1.0],
                                               1.0, np.deg2rad(lat), np.deg2rad(lon))
    assert np.array_equal(cp[0], [0, 0, 1500]) and np.array_equal(cn[0], [0, 0, -1])
    assert np.array_equal(cn[1], [0, 0, 1])
    assert cp[1][2] == np.float32(ko.crust_thickness(prof))


@pytest.mark.parametrize("st", [4, 5])
def test_eikonal_discretisers_bitexact(st):
    rng = np.random.default_rng(20 + st)
    lat, lon = 40.75, 29.86
    lat_rad, lon_rad = float(np.deg2rad(lat)), float(np.deg2rad(lon))
    # psm_make_*_grid looks the profile up with the origin in RADIANS (source_mt_eikonal.f90:480): reproduce
    prof_speed = ref_profile(lat_rad, lon_rad)
    ncase = 0
    for _ in range(12):
        common = [rng.uniform(-1, 1), rng.uniform(-3e3, 3e3), rng.uniform(-3e3, 3e3), rng.uniform(4e3, 2.5e4)]
        strike, dip = rng.uniform(-180, 180), rng.uniform(5, 90)
        bord = [rng.uniform(-2e3, 2e3), rng.uniform(-2e3, 2e3), rng.uniform(2e3, 1.2e4)]
        nukl = [rng.uniform(-1, 1) * 0.5 * bord[2], rng.uniform(-1, 1) * 0.3 * bord[2]]
        relv, rise = rng.uniform(0.6, 1.0), rng.uniform(0, 3)
        if st == 5:
            p = common + [1.0, strike, dip] + bord + nukl + [relv] + list(rng.standard_normal(6) * 1e18) + [rise]
        else:
            p = common + [10 ** rng.uniform(17, 19), strike, dip, rng.uniform(-180, 180)] + bord + nukl + [relv, rise]
        edt = float(rng.choice([0.5, 1.0, 2.0]))
        nc, cent, mo, ri, gs, cp, cn = ref_eikonal(st, p, edt, lat_rad, lon_rad)
        try:
            a, amo, ari, ags = ko.discretize_eikonal(st, p, edt, prof_speed, cp, cn)
        except ValueError:
            assert nc == -1
            continue
        assert nc == len(a) and gs == ags
        assert amo == mo and ari == ri
        assert np.array_equal(a.view(np.uint32), cent.view(np.uint32))
        ncase += 1
    assert ncase >= 6
