"""The reference's own unit-test vectors, re-expressed as data and run
through the oracle.  Each case cites the reference test file:line it comes from."""
import ctypes as C

import numpy as np

from oracle import ko

fp = ko._fp
f32 = lambda *v: np.array(v, np.float32)


# ---------------------------------------------------------------- test_sparse_trace.f90
def test_pack_gap_rule():
    # test_sparse_trace.f90:32-46: 20-sample strip at (21,40) packs into (24,27),(33,40)
    d = f32(0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0)
    spans, tspan = ko.trace_pack_spans(21, d)
    assert spans == [(24, 27), (33, 40)]
    assert tspan == (24, 40)
    # :76-81 no trimming when data starts/ends non-zero
    spans, _ = ko.trace_pack_spans(1, f32(3, 1, 1, 99))
    assert spans == [(1, 4)]


def test_multiply_add_order_independent():
    # test_sparse_trace.f90:48-73: join/unpack == multiply-add in either order
    d1 = f32(0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0)
    a = ko.multiply_add(21, d1, None)
    a = ko.multiply_add(51, f32(6), a)
    b = ko.multiply_add(51, f32(6), None)
    b = ko.multiply_add(21, d1, b)
    expect = np.zeros(51 - 24 + 1, np.float32)
    expect[0:4] = d1[3:7]
    expect[33 - 24:40 - 24 + 1] = d1[12:20]
    expect[-1] = 6
    for lo, got in (a, b):
        assert lo == 24 and np.array_equal(got, expect)


def test_integer_shift():
    # test_sparse_trace.f90:84-90
    lo, got = ko.multiply_add(2, f32(1, 1), (1, f32(1, 1)), 1.0, mode=1, ishift=-1)
    assert lo == 1 and np.array_equal(got, f32(2, 2))


def test_fractional_shift_kat():
    # test_sparse_trace.f90:105-110: (2,4)=(1,1,0) shifted by -0.25 onto (1,1)=(0) -> (0.25,1,0.75,0)
    lo, got = ko.multiply_add(2, f32(1, 1, 0), (1, f32(0)), 1.0, mode=2, rshift=-0.25)
    assert lo == 1 and np.array_equal(got, f32(0.25, 1, 0.75, 0))


def test_strip_dataspan_kats():
    # test_sparse_trace.f90:113-123
    assert ko.strip_dataspan(-2, f32(0, 0, 1, 2, 2, 2, 2, 2)) == (0, 1)
    assert ko.strip_dataspan(-2, f32(1, 1, 1, 2, 2, 2, 2, 3)) == (-2, 5)
    assert ko.strip_dataspan(-2, f32(0, 0, 0)) == (0, -2)


# ---------------------------------------------------------------- test_comparator.f90
class Probes:
    def __init__(self, dt=1.0):
        self.L = ko.lib()
        self.a = (C.c_char * 4096)()
        self.b = (C.c_char * 4096)()
        self.L.ko_probe_init(self.a, C.c_float(dt))
        self.L.ko_probe_init(self.b, C.c_float(dt))

    def set(self, p, lo, data):
        s = ko.strip_from(lo, data)
        self.L.ko_probe_set_array(p, C.byref(s), C.c_float(1.0))
        self.L.ko_strip_destroy(C.byref(s))

    def norm(self, method):
        return self.L.ko_probes_norm(self.a, self.b, C.c_int(method))


def test_comparator_time_domain_kats():
    P = Probes()
    eps = 1e-6
    # test_comparator.f90:40-47  equal up to zero-left / constant-right extension
    P.set(P.a, -1, f32(0, 0, 5, 1)); P.set(P.b, 1, f32(5, 1, 1, 1))
    assert P.norm(1) == 0.0 and P.norm(2) == 0.0
    # :49-56
    P.set(P.a, -1, f32(0, 0, 0, 1)); P.set(P.b, 1, f32(0, 1, 1, 1))
    assert P.norm(1) == 0.0 and P.norm(2) == 0.0
    # :60-67 sqrt(2) / 2
    P.set(P.a, -4, f32(1, 0, 0, 0)); P.set(P.b, 1, f32(1, 0, 0, 0))
    assert abs(P.norm(1) - np.sqrt(2.)) < eps and abs(P.norm(2) - 2.) < eps
    # :69-76 sqrt(3) / 3
    P.set(P.a, 0, f32(1, 2, 1, 0)); P.set(P.b, 1, f32(1, 1, 0, 1))
    assert abs(P.norm(1) - np.sqrt(3.)) < eps and abs(P.norm(2) - 3.) < eps
    # :78-85 1 / 1
    P.set(P.a, 0, f32(0, 1, 2, 1)); P.set(P.b, 1, f32(1, 2))
    assert abs(P.norm(1) - 1.) < eps and abs(P.norm(2) - 1.) < eps


def test_comparator_ampspec_shift_invariant():
    # test_comparator.f90:88-94: amplitude spectra of shifted copies agree
    P = Probes()
    P.set(P.a, 0, f32(0, 1, 2, 1, 0)); P.set(P.b, 10, f32(0, 1, 2, 1, 0))
    assert abs(P.norm(3)) < 1e-6


def test_comparator_cross_correlation_kats():
    # test_comparator.f90:96-113
    P = Probes()
    P.set(P.a, 1, f32(0, 1, 2, 1, 0)); P.set(P.b, 2, f32(0, 1, 2, 1, 0))
    cc = np.zeros(11, np.float32)
    P.L.ko_probes_windowed_cross_corr(P.a, P.b, C.c_int(-5), C.c_int(5), fp(cc))
    assert np.array_equal(cc, f32(0, 0, 1, 4, 6, 4, 1, 0, 0, 0, 0))
    taper = ko.make_plf([2.5, 3.5], [1., 1.])
    P.L.ko_probe_set_taper(P.a, C.byref(taper))
    P.L.ko_probe_set_taper(P.b, C.byref(taper))
    P.L.ko_probes_windowed_cross_corr(P.a, P.b, C.c_int(-5), C.c_int(5), fp(cc))
    assert np.array_equal(cc, f32(0, 0, 0, 2, 4, 2, 0, 0, 0, 0, 0))


def test_next_power_of_two():
    # comparator.f90:1111-1118 as the flang-built reference evaluates it
    L = ko.lib()
    for e in range(1, 17):
        assert L.ko_next_power_of_two(2 ** e) == 2 ** e
        assert L.ko_next_power_of_two(2 ** e + 1) == 2 ** (e + 1)


# ---------------------------------------------------------------- test_source_bilat.f90
def _bilat_case(strike, dip, rake):
    # parameter vectors of test_source_bilat.f90:47-60 and :96-109, effective dt 0.5 (:62,:111)
    p = [0., 0., 0., 1000., 1., strike, dip, rake, 0., 2000., 0., 1000., 2000., 1.]
    cent, moment, risetime, grid = ko.discretize(1, p, 0.5)
    return cent[:, 4:10], len(cent)


def test_source_bilat_thrust():
    # test_source_bilat.f90:64-92: strike 90, dip 45, rake 90 => every centroid has mxx = -mzz < 0,
    # all other elements 0 (|.| < 1/n/100), and sum(mxx) = -1 +- 0.01
    m, n = _bilat_case(90., 45., 90.)
    epsm = 1. / n / 100
    mxx, myy, mzz, mxy, mxz, myz = m.T
    assert np.all(np.abs(mxx) >= epsm) and np.all(np.abs(mzz) >= epsm)
    assert np.all(np.abs(mxx + mzz) < epsm) and np.all(mxx <= 0)
    for other in (myy, mxy, mxz, myz):
        assert np.all(np.abs(other) < epsm)
    assert abs(-1. - np.float32(mxx.sum(dtype=np.float32))) < 0.01


def test_source_bilat_strike_slip():
    # test_source_bilat.f90:113-141: strike 45, dip 90, rake 0 => mxx = -myy < 0, others 0, sum(mxx) = -1
    m, n = _bilat_case(45., 90., 0.)
    epsm = 1. / n / 100
    mxx, myy, mzz, mxy, mxz, myz = m.T
    assert np.all(np.abs(mxx) >= epsm) and np.all(np.abs(myy) >= epsm)
    assert np.all(np.abs(mxx + myy) < epsm) and np.all(mxx <= 0)
    for other in (mzz, mxy, mxz, myz):
        assert np.all(np.abs(other) < epsm)
    assert abs(-1. - np.float32(mxx.sum(dtype=np.float32))) < 0.01


# ---------------------------------------------------------------- test_orthodrome.f90
def test_hamburg_munich_distance():
    # test_orthodrome.f90:87-96: 612.59 km +- 50 m
    class Geo(C.Structure):
        _fields_ = [("lat", C.c_double), ("lon", C.c_double)]
    L = ko.lib()
    L.ko_distance_accurate50m.argtypes = [Geo, Geo]
    hh = Geo(L.ko_d2r_d(53.556867), L.ko_d2r_d(9.994622))
    mu = Geo(L.ko_d2r_d(48.139743), L.ko_d2r_d(11.560050))
    d = L.ko_distance_accurate50m(hh, mu)
    assert abs(d / 1000. - 612.59) < 0.05


# ---------------------------------------------------------------- test_piecewise_linear_function.f90
def test_plf_integrals():
    # test_piecewise_linear_function.f90:29-59 exact integrals of the trapezoid (0,0)(1,1)(2,1)(3,0)
    L = ko.lib()
    L.ko_plf_integrate.argtypes = [C.c_void_p, C.c_float, C.c_float]
    func = ko.make_plf([0., 1., 2., 3.], [0., 1., 1., 0.])
    integ = lambda a, b: L.ko_plf_integrate(C.byref(func), a, b)
    assert integ(-1., 3.) == 2.
    assert integ(-1., 1.5) == 1.
    assert integ(0.5, 2.5) == np.float32(6. / 8. + 1.)
    assert integ(2., 2.5) == np.float32(3. / 8.)
    assert integ(2., 2.) == 0.
    assert integ(2.5, 2.5) == 0.
    assert integ(2.5, 2.75) == np.float32(1. / 8. - 1. / 32.)
    assert integ(1., 2.) == 1.


def test_plf_integrate_and_centroid():
    # test_piecewise_linear_function.f90:61-74 (the reference only fails if BOTH differ; check both)
    L = ko.lib()
    func = ko.make_plf([0., 1., 2., 3.], [0., 1., 1., 0.])

    def ic(a, b):
        ar, ce = C.c_float(), C.c_float()
        L.ko_plf_integrate_and_centroid(C.byref(func), C.c_float(a), C.c_float(b), C.byref(ar), C.byref(ce))
        return ar.value, ce.value
    a, c = ic(-1., 6.)
    assert a == 2. and abs(c - 1.5) < 1e-6
    a, c = ic(0., 0.5)
    assert a == np.float32(1. / 8.) and abs(c - 1. / 3.) < 1e-6
    a, c = ic(0., 2.)
    assert a == np.float32(3. / 2.) and abs(c - (1. + 2. / 9.)) < 1e-6


# ---------------------------------------------------------------- test_eikonal.f90:33-60
def test_eikonal_uniform_speed():
    import ctypes as C
    L = ko.lib()
    nx, ny = 500, 1000
    speed = np.full((ny, nx), 2.0, np.float32)
    delta = np.array([50. / nx, 50. / ny], np.float32)
    origin = np.zeros(2, np.float32)
    start = np.array([0., 25.], np.float32)
    t = np.zeros((ny, nx), np.float32)
    L.ko_eikonal_solver_fmm(ko._fp(speed), C.c_int(nx), C.c_int(ny), ko._fp(origin), ko._fp(delta), ko._fp(start), ko._fp(t))
    eps = float(delta.max()) / 2.0
    assert abs(t[0, 0] - 12.5) < eps and abs(t[ny - 1, 0] - 12.5) < eps           # times(1,1), times(1,ny)
    assert abs(t[0, nx - 1] - 27.95) < eps and abs(t[ny - 1, nx - 1] - 27.95) < eps  # times(nx,1), times(nx,ny)


# ---------------------------------------------------------------- test_euler.f90:25-62
def test_euler_quarter_turns():
    import ctypes as C
    L = ko.lib()
    L.ko_init_euler.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    pi = np.float32(3.14159265358979)
    want = {"alpha": [[1, 0, 0], [0, 0, 1], [0, -1, 0]], "beta": [[0, 1, 0], [-1, 0, 0], [0, 0, 1]],
            "gamma": [[0, 1, 0], [-1, 0, 0], [0, 0, 1]]}                            # columns = images of the unit vectors
    for name, args in (("alpha", (pi / 2, 0, 0)), ("beta", (0, pi / 2, 0)), ("gamma", (0, 0, pi / 2))):
        rot = np.zeros((3, 3), np.float32)
        L.ko_init_euler(C.c_float(args[0]), C.c_float(args[1]), C.c_float(args[2]), ko._fp(rot))
        rotsys = np.stack([rot @ np.eye(3, dtype=np.float32)[:, i] for i in range(3)])   # rotsys(:,i) as rows
        assert np.all(np.abs(rotsys - np.array(want[name], np.float32)) < 1e-3), name


# ---------------------------------------------------------------- test_geometry.f90:25-92
def test_geometry_halfspace_piercing_and_trimmed_circle():
    import ctypes as C
    L = ko.lib()
    fp = ko._fp
    hp, hn = np.array([0, 0, -1], np.float32), np.array([0, -1, -1], np.float32)
    pts = np.array([[0, 2, -1], [0, -2, -1], [0, 0, -1], [0, 0, 0]], np.float32)
    assert [bool(L.ko_point_in_halfspace(fp(p), fp(hp), fp(hn))) for p in pts] == [True, False, True, True]

    def pierce(a, b):
        a, b = np.array(a, np.float32), np.array(b, np.float32)
        pp = np.zeros(3, np.float32)
        bt, par = C.c_int(), C.c_int()
        L.ko_get_piercingpoint(fp(a), fp(b), fp(hp), fp(hn), fp(pp), C.byref(bt), C.byref(par))
        return pp, bool(bt.value), bool(par.value)

    pp, bt, par = pierce(pts[0], pts[1])
    assert np.array_equal(pp, [0, 0, -1]) and bt and not par
    pp, bt, par = pierce([0, 2, -1], [0, 1, -2])
    assert np.array_equal(pp, [0, 1, -2]) and not bt and not par
    pp, bt, par = pierce([0, 2, 5], [0, 1, 1])
    assert np.all(np.abs(pp - [0, 0.4, -1.4]) < 1e-4) and not bt and not par
    pp, bt, par = pierce([0, 1, 0], [0, 2, -1.0001])
    assert np.array_equal(pp, [0, 0, 0]) and not bt and par
    # circle of radius 3, dip = strike = 45 degrees, centre (0,0,1), 7 points, cut by z <= 0 ... expected_circ (:36-40)
    d2r = np.float32(2.) / np.float32(360.) * np.float32(3.14159265358979)
    rot = np.zeros((3, 3), np.float32)
    L.ko_init_euler.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    L.ko_init_euler(C.c_float(d2r * 45), C.c_float(d2r * 45), C.c_float(0), fp(rot))
    tr = np.ascontiguousarray(rot * np.float32(3.))
    out = np.zeros((16, 3), np.float32)
    L.ko_trim_circle.restype = C.c_int
    n = L.ko_trim_circle(fp(np.array([0, 0, 1], np.float32)), fp(tr), C.c_int(7), fp(np.zeros(3, np.float32)),
                         fp(np.array([0, 0, -1], np.float32)), fp(out), C.c_int(16))
    expected = np.array([0.14987442, 2.4953687, 2.6585152, -1.9344299, 0.9903534, 3.0681345, -2.5620692, -1.2604182,
                         1.9204066, -1.2604178, -2.5620692, 0.07959348, -1.1043297, -2.5185432, 0., 2.3468528,
                         0.9326396, 0., 2.12132, 2.1213207, 1.0000004], np.float32).reshape(7, 3)
    assert n == 7 and np.all(np.abs(out[:7] - expected) < 1e-5)
