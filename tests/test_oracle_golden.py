"""Oracle against committed golden vectors: outputs of the reference's own Fortran modules
(tests/golden/make_golden.py, generated in the dev container from oracle/_ref).  Runs anywhere."""
import ctypes as C
import os

import numpy as np

from oracle import ko

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"))
fp = ko._fp


def biteq(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def test_trace_multiply_add_golden():
    for i in range(int(G["ma_n"])):
        tlo, has_s, slo, mode, ish, olo = [int(v) for v in G["ma%d_p" % i]]
        factor, rsh = G["ma%d_f" % i]
        strip = (slo, G["ma%d_s" % i]) if has_s else None
        lo, o = ko.multiply_add(tlo, G["ma%d_t" % i], strip, factor, mode, ish, rsh)
        assert lo == olo and biteq(o, G["ma%d_o" % i]), i


def test_blend_golden():
    L = ko.lib()
    L.ko_gfdb_get_trace_bilin.restype = C.POINTER(ko.Trace)
    L.ko_gfdb_get_trace_bilin.argtypes = [C.c_void_p, ko.c_int_p, ko.c_int_p, C.c_int, C.c_float, C.c_float]
    for i in range(int(G["bl_n"])):
        lo, n, data = G["bl%d_lo" % i], G["bl%d_n" % i], G["bl%d_d" % i]
        dix, diz = G["bl%d_w" % i]
        db = ko.Gfdb(2, 2, 1, 1.0, 1.0, 1.0, 0.0, 0.0)
        k = 0
        for ix in (1, 2):
            for iz in (1, 2):
                db.set_trace(ix, iz, 1, int(lo[k]), data[k, :n[k]])
                k += 1
        ix = np.array([1, 2], np.int32)
        iz = np.array([1, 2], np.int32)
        t = L.ko_gfdb_get_trace_bilin(db.h, ko._ip(ix), ko._ip(iz), 1, dix, diz).contents
        got = np.ctypeslib.as_array(t.strips[0].d, (t.strips[0].n,)).copy()
        assert t.strips[0].lo == int(G["bl%d_olo" % i]) and biteq(got, G["bl%d_o" % i]), i
        db.close()


def test_orthodrome_golden():
    class Geo(C.Structure):
        _fields_ = [("lat", C.c_double), ("lon", C.c_double)]
    L = ko.lib()
    L.ko_azibazi.argtypes = [Geo, Geo, ko.c_double_p, ko.c_double_p]
    L.ko_distance_accurate50m.argtypes = [Geo, Geo]
    for (alat, alon, blat, blon), (dx, dy), want in zip(G["or_in"], G["or_dxy"], G["or_out"]):
        a, b = C.c_double(), C.c_double()
        L.ko_azibazi(Geo(alat, alon), Geo(blat, blon), C.byref(a), C.byref(b))
        d = L.ko_distance_accurate50m(Geo(alat, alon), Geo(blat, blon))
        na, nb, nd = C.c_double(), C.c_double(), C.c_double()
        L.ko_approx_differential_azidist(C.c_float(dx), C.c_float(dy), a, b, C.c_double(d), C.byref(na), C.byref(nb),
                                         C.byref(nd))
        got = np.array([a.value, b.value, d, na.value, nb.value, nd.value])
        # glibc's real*8 libm selects FMA builds on CPUs that have FMA: allow 4 ulp of fp64 between boxes
        assert np.allclose(got, want, rtol=1e-15, atol=1e-15), (got, want)


def test_discretise_golden():
    for k in range(int(G["ds_n"])):
        v = G["ds%d_in" % k]
        st, edt, par = int(v[0]), float(v[1]), v[2:]
        cent, mo, ri, _ = ko.discretize(st, par, edt)
        assert biteq(cent, G["ds%d_c" % k]), k
        assert biteq(np.array([mo, ri], np.float32), G["ds%d_mr" % k])


def test_strip_fold_golden():
    for i in range(int(G["sf_n"])):
        lo, olo = [int(v) for v in G["sf%d_p" % i]]
        a = ko.strip_fold(lo, G["sf%d_d" % i], G["sf%d_sh" % i], G["sf%d_am" % i])
        assert a[0] == olo and biteq(a[1], G["sf%d_o" % i]), i


def test_taper_weights_golden():
    L = ko.lib()
    for i in range(int(G["tp_n"])):
        x = G["tp%d_x" % i]
        lo, hi = [int(v) for v in G["tp%d_p" % i]]
        p = ko.make_plf(x, [0., 1., 1., 0.])
        arr = np.ones(hi - lo + 1, np.float32)
        L.ko_plf_taper_array_r(C.byref(p), fp(arr), C.c_int(lo), C.c_int(hi), C.c_float(float(G["tp%d_dx" % i])), C.c_int(0))
        assert biteq(arr, G["tp%d_o" % i]), i


def test_principal_axes_golden():
    """psm%pax / psm%tax of bilateral sources (source_bilat.f90:216-239): oracle and product host, bit for bit."""
    from kiwi_amd import engine
    L = ko.lib()
    L.ko_principal_axes_bilat.restype = None
    for p, want in zip(G["pa_in"], G["pa_out"]):
        pax, tax = np.zeros(2, np.float32), np.zeros(2, np.float32)
        L.ko_principal_axes_bilat(fp(np.ascontiguousarray(p)), fp(pax), fp(tax))
        assert biteq(np.concatenate([pax, tax]), want)
        ppax, ptax = engine.principal_axes("bilateral", p)
        assert biteq(np.concatenate([ppax, ptax]), want)
    import pytest
    from kiwi_amd import KiwiHipError
    with pytest.raises(KiwiHipError):
        engine.principal_axes("circular", np.zeros(11, np.float32))
