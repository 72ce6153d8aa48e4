"""Oracle (oracle/libko.so, C restatement) against the reference's own Fortran modules
(oracle/_ref/libkiwi_ref.so, built unmodified from /root/reference by `make -C oracle ref`).
Bit-for-bit on seeded random inputs.  Skipped when the reference build is absent."""
import ctypes as C

import numpy as np
import pytest

from oracle import ko

R = ko.LazyRef()
pytestmark = pytest.mark.skipif(not ko.ref_available(), reason="oracle/_ref/libkiwi_ref.so not built")

fp = ko._fp


def rnd_trace(rng, n, gaps=True):
    d = rng.standard_normal(n).astype(np.float32)
    if gaps:
        for _ in range(rng.integers(0, 4)):
            a = rng.integers(0, n)
            d[a:a + rng.integers(1, 12)] = 0
    if rng.random() < 0.3:
        d[-rng.integers(1, 5):] = 0
    if rng.random() < 0.3:
        d[:rng.integers(1, 5)] = 0
    return d


def ref_multiply_add(tlo, tdata, strip, factor, mode, ishift, rshift):
    omax = 4096
    out = np.zeros(omax, np.float32)
    olo, on = C.c_int(), C.c_int()
    has_s = strip is not None
    slo, sdata = strip if has_s else (0, np.zeros(1, np.float32))
    sdata = np.ascontiguousarray(sdata, np.float32)
    R.ref_multiply_add(C.c_int(tlo), C.c_int(len(tdata)), fp(tdata), C.c_int(has_s), C.c_int(slo),
                       C.c_int(len(sdata)), fp(sdata), C.c_float(factor), C.c_int(mode), C.c_int(ishift),
                       C.c_float(rshift), C.c_int(omax), C.byref(olo), C.byref(on), fp(out))
    return olo.value, out[:on.value].copy()


def test_trace_pack_spans():
    rng = np.random.default_rng(1)
    for _ in range(200):
        n = int(rng.integers(1, 80))
        d = rnd_trace(rng, n)
        if rng.random() < 0.05:
            d[:] = 0
        lo = int(rng.integers(-50, 50))
        spans, tspan = ko.trace_pack_spans(lo, d)
        ns = C.c_int()
        rs = np.zeros((64, 2), np.int32)
        ts = (C.c_int * 2)()
        R.ref_trace_pack(C.c_int(lo), C.c_int(n), fp(d), C.c_int(64), C.byref(ns), ko._ip(rs), ts)
        assert ns.value == len(spans)
        assert [tuple(x) for x in rs[:ns.value]] == spans
        assert (ts[0], ts[1]) == tspan


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_trace_multiply_add_bitexact(mode):
    rng = np.random.default_rng(10 + mode)
    for _ in range(300):
        n = int(rng.integers(1, 120))
        d = rnd_trace(rng, n)
        tlo = int(rng.integers(-30, 60))
        strip = None
        if rng.random() < 0.7:
            sn = int(rng.integers(1, 150))
            strip = (int(rng.integers(-40, 80)), rng.standard_normal(sn).astype(np.float32))
        factor = np.float32(rng.standard_normal() * 10)
        ishift = int(rng.integers(-20, 20))
        rshift = np.float32(rng.uniform(-20, 20))
        if rng.random() < 0.2:
            rshift = np.float32(np.round(rshift))
        a = ko.multiply_add(tlo, d, strip, factor, mode, ishift, rshift)
        b = ref_multiply_add(tlo, d, strip, factor, mode, ishift, rshift)
        assert a[0] == b[0]
        assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def test_blend4_bitexact():
    """gfdb_get_trace_bilin's summation: oracle gfdb blend vs the reference primitive."""
    rng = np.random.default_rng(5)
    L = ko.lib()
    for _ in range(100):
        nmax = 160
        lo = rng.integers(0, 30, 4).astype(np.int32)
        n = rng.integers(5, nmax, 4).astype(np.int32)
        data = np.zeros((4, nmax), np.float32)
        for i in range(4):
            data[i, :n[i]] = rnd_trace(rng, int(n[i]))
        dix, diz = np.float32(rng.random()), np.float32(rng.random())
        out = np.zeros(1024, np.float32)
        olo, on = C.c_int(), C.c_int()
        R.ref_blend4(ko._ip(lo), ko._ip(n), C.c_int(nmax), fp(data), C.c_float(dix), C.c_float(diz),
                     C.byref(olo), C.byref(on), fp(out))
        # oracle: 2x2 node gfdb, one component
        db = ko.Gfdb(2, 2, 1, 1.0, 1.0, 1.0, 0.0, 0.0)
        k = 0
        for ix in (1, 2):
            for iz in (1, 2):          # order t00,t01,t10,t11
                db.set_trace(ix, iz, 1, int(lo[k]), data[k, :n[k]])
                k += 1
        L.ko_gfdb_get_trace_bilin.restype = C.POINTER(ko.Trace)
        L.ko_gfdb_get_trace_bilin.argtypes = [C.c_void_p, ko.c_int_p, ko.c_int_p, C.c_int, C.c_float, C.c_float]
        ix = np.array([1, 2], np.int32)
        iz = np.array([1, 2], np.int32)
        t = L.ko_gfdb_get_trace_bilin(db.h, ko._ip(ix), ko._ip(iz), 1, dix, diz).contents
        assert t.nstrips == 1
        got = np.ctypeslib.as_array(t.strips[0].d, (t.strips[0].n,)).copy()
        assert t.strips[0].lo == olo.value and len(got) == on.value
        assert np.array_equal(got.view(np.uint32), out[:on.value].view(np.uint32))
        db.close()


def test_strip_dataspan_and_fold():
    rng = np.random.default_rng(7)
    for _ in range(200):
        n = int(rng.integers(1, 100))
        d = rnd_trace(rng, n, gaps=False)
        if rng.random() < 0.5:
            d[-int(rng.integers(1, 10)):] = d[-1]
        lo = int(rng.integers(-20, 20))
        ds = (C.c_int * 2)()
        R.ref_strip_dataspan(C.c_int(lo), C.c_int(n), fp(d), ds)
        assert ko.strip_dataspan(lo, d) == (ds[0], ds[1])
        ns = 1 + 2 * int(rng.integers(0, 5))
        shifts = (np.arange(ns) - (ns - 1) / 2).astype(np.float32)
        amps = rng.random(ns).astype(np.float32)
        amps /= amps.sum()
        out = np.zeros(1024, np.float32)
        olo, on = C.c_int(), C.c_int()
        R.ref_strip_fold(C.c_int(lo), C.c_int(n), fp(d), C.c_int(ns), fp(shifts), fp(amps), C.c_int(1024),
                         C.byref(olo), C.byref(on), fp(out))
        a = ko.strip_fold(lo, d, shifts, amps)
        assert a[0] == olo.value
        assert np.array_equal(a[1].view(np.uint32), out[:on.value].view(np.uint32))
    # the corner cases the device's span of an un-tapered synthetic rests on (kiwi_misfit.hpp synspan_kernel): all zeros (the fold
    # leaves a strip of more than one sample alone: its data span is inverted), one sample, one constant value
    for d in (np.zeros(5, np.float32), np.zeros(1, np.float32), np.full(4, 2.5, np.float32), np.array([0, 0, 1, 3, 3, 3], np.float32),
              np.array([0, 1, 0, 0], np.float32)):
        lo, n = 7, len(d)
        ds = (C.c_int * 2)()
        R.ref_strip_dataspan(C.c_int(lo), C.c_int(n), fp(d), ds)
        assert ko.strip_dataspan(lo, d) == (ds[0], ds[1])
        shifts = np.array([-1, 0, 1], np.float32)
        amps = np.array([0.25, 0.5, 0.25], np.float32)
        out = np.zeros(64, np.float32)
        olo, on = C.c_int(), C.c_int()
        R.ref_strip_fold(C.c_int(lo), C.c_int(n), fp(d), C.c_int(3), fp(shifts), fp(amps), C.c_int(64), C.byref(olo), C.byref(on), fp(out))
        a = ko.strip_fold(lo, d, shifts, amps)
        assert a[0] == olo.value and len(a[1]) == on.value
        assert np.array_equal(a[1].view(np.uint32), out[:on.value].view(np.uint32))
        # the span rule the device uses: extent [lo, lo + n - 1] joined with [d1 - 1, d2 + 1 + 1] when the data span is not inverted
        d1, d2 = ds[0], ds[1]
        want = (lo, lo + n - 1) if d2 < d1 else (min(lo, d1 - 1), max(lo + n - 1, d2 + 2))
        assert (olo.value, olo.value + on.value - 1) == want, (d, want, olo.value, on.value)


def test_d2r():
    rng = np.random.default_rng(3)
    L = ko.lib()
    for x in rng.uniform(-360, 360, 500):
        assert L.ko_d2r_d(x) == R.ref_d2r_d(x)
        assert L.ko_d2r_r(np.float32(x)) == R.ref_d2r_r(np.float32(x))


def test_orthodrome_bitexact():
    rng = np.random.default_rng(4)
    L = ko.lib()
    for _ in range(500):
        alat, alon = np.deg2rad(rng.uniform(-80, 80)), np.deg2rad(rng.uniform(-180, 180))
        blat, blon = alat + rng.uniform(-0.2, 0.2), alon + rng.uniform(-0.2, 0.2)
        ra, rb, rd = C.c_double(), C.c_double(), C.c_double()
        R.ref_azibazi_dist(C.c_double(alat), C.c_double(alon), C.c_double(blat), C.c_double(blon),
                           C.byref(ra), C.byref(rb), C.byref(rd))
        a, b, d = geo(alat, alon, blat, blon)
        assert (a, b, d) == (ra.value, rb.value, rd.value)
        dx, dy = np.float32(rng.uniform(-30000, 30000)), np.float32(rng.uniform(-30000, 30000))
        if rng.random() < 0.05:
            dx = dy = np.float32(0)
        na, nb, nd = C.c_double(), C.c_double(), C.c_double()
        R.ref_approx_differential_azidist(C.c_float(dx), C.c_float(dy), ra, rb, rd,
                                          C.byref(na), C.byref(nb), C.byref(nd))
        oa, ob, od = C.c_double(), C.c_double(), C.c_double()
        L.ko_approx_differential_azidist(C.c_float(dx), C.c_float(dy), ra, rb, rd,
                                         C.byref(oa), C.byref(ob), C.byref(od))
        assert (oa.value, ob.value, od.value) == (na.value, nb.value, nd.value)


class Geo(C.Structure):
    _fields_ = [("lat", C.c_double), ("lon", C.c_double)]


def geo(alat, alon, blat, blon):
    L = ko.lib()
    L.ko_azibazi.argtypes = [Geo, Geo, ko.c_double_p, ko.c_double_p]
    L.ko_distance_accurate50m.argtypes = [Geo, Geo]
    a, b = C.c_double(), C.c_double()
    L.ko_azibazi(Geo(alat, alon), Geo(blat, blon), C.byref(a), C.byref(b))
    return a.value, b.value, L.ko_distance_accurate50m(Geo(alat, alon), Geo(blat, blon))


def test_euler_and_plf():
    rng = np.random.default_rng(6)
    L = ko.lib()
    for _ in range(200):
        al, be, ga = [np.float32(v) for v in rng.uniform(-7, 7, 3)]
        m = np.zeros((3, 3), np.float32)
        R.ref_init_euler(C.c_float(al), C.c_float(be), C.c_float(ga), fp(m))     # column-major
        o = np.zeros((3, 3), np.float32)
        L.ko_init_euler(C.c_float(al), C.c_float(be), C.c_float(ga), fp(o))       # o[row][col]
        assert np.array_equal(o.view(np.uint32), m.T.view(np.uint32))
    for _ in range(300):
        n = int(rng.integers(2, 6))
        x = np.sort(rng.uniform(-5, 5, n)).astype(np.float32)
        if rng.random() < 0.3 and n >= 4:
            x[1] = x[0]
            x[-1] = x[-2]
        y = rng.uniform(0, 2, n).astype(np.float32)
        a, b = sorted(np.float32(v) for v in rng.uniform(-6, 6, 2))
        p = ko.make_plf(x, y)
        ar, ce = C.c_float(), C.c_float()
        L.ko_plf_integrate_and_centroid(C.byref(p), C.c_float(a), C.c_float(b), C.byref(ar), C.byref(ce))
        rar, rce = C.c_float(), C.c_float()
        R.ref_plf_integrate_and_centroid(C.c_int(n), fp(x), fp(y), C.c_float(a), C.c_float(b),
                                         C.byref(rar), C.byref(rce))
        assert np.float32(ar.value).view(np.uint32) == np.float32(rar.value).view(np.uint32)
        same_c = np.float32(ce.value).view(np.uint32) == np.float32(rce.value).view(np.uint32)
        assert same_c or (np.isnan(ce.value) and np.isnan(rce.value))
        for ip in (0, 1, 2):
            lo, hi = int(rng.integers(-40, 0)), int(rng.integers(1, 40))
            arr = rng.standard_normal(hi - lo + 1).astype(np.float32)
            arr2 = arr.copy()
            dx = np.float32(rng.choice([0.25, 0.5, 0.3, 1.0]))
            L.ko_plf_taper_array_r(C.byref(p), fp(arr), C.c_int(lo), C.c_int(hi), C.c_float(dx), C.c_int(ip))
            R.ref_plf_taper_array_r(C.c_int(n), fp(x), fp(y), C.c_int(lo), C.c_int(hi), fp(arr2),
                                    C.c_float(dx), C.c_int(ip))
            assert np.array_equal(arr.view(np.uint32), arr2.view(np.uint32))


def ref_discretize(st, params, edt):
    params = np.ascontiguousarray(params, np.float32)
    maxc = 100000
    cent = np.zeros((maxc, 10), np.float32)
    nc, gs = C.c_int(), (C.c_int * 3)()
    mo, rt = C.c_float(), C.c_float()
    R.ref_discretize(C.c_int(st), C.c_int(len(params)), fp(params), C.c_float(edt), C.c_int(maxc),
                     C.byref(nc), fp(cent), C.byref(mo), C.byref(rt), gs)
    return cent[:nc.value].copy(), mo.value, rt.value, tuple(gs)


def test_discretize_bitexact():
    rng = np.random.default_rng(8)
    for _ in range(60):
        # bilateral (source_bilat.f90:93-106)
        p = [rng.uniform(-2, 2), rng.uniform(-5e3, 5e3), rng.uniform(-5e3, 5e3), rng.uniform(2e3, 3e4),
             10 ** rng.uniform(17, 20), rng.uniform(-180, 180), rng.uniform(0, 90), rng.uniform(-180, 180),
             rng.uniform(-180, 180), rng.uniform(0, 2e4), rng.uniform(0, 1e4), rng.uniform(0, 1e4),
             rng.uniform(1500, 4000), rng.uniform(0, 3)]
        if rng.random() < 0.1:
            p[9] = p[10] = 0.0
        if rng.random() < 0.1:
            p[11] = 0.0
        edt = float(rng.choice([0.5, 1.0, 2.0]))
        a = ko.discretize(1, p, edt)
        b = ref_discretize(1, p, edt)
        assert a[3] == b[3]
        assert a[1] == b[1] and a[2] == b[2]
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        # circular (source_circular.f90:92-102)
        q = p[:8] + [rng.uniform(0, 1.5e4), p[12], p[13]]
        a = ko.discretize(2, q, edt)
        b = ref_discretize(2, q, edt)
        assert a[3] == b[3] and a[0].shape == b[0].shape
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        # moment tensor (source_moment_tensor.f90:90-100)
        mt = p[:4] + list(rng.standard_normal(6) * 1e18) + [rng.uniform(0.01, 4)]
        a = ko.discretize(6, mt, edt)
        b = ref_discretize(6, mt, edt)
        assert a[3][0] == b[3][0]
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        # point_lp (source_point_lp.f90:237-337; default-real exp / sin in the source time function)
        lp = p[:5] + list(rng.standard_normal(6)) + [rng.uniform(1, 40), rng.uniform(2, 30)]
        a = ko.discretize(3, lp, edt)
        b = ref_discretize(3, lp, edt)
        assert a[3] == b[3] and a[1] == b[1] and a[2] == b[2]
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))

