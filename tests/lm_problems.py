"""Least-squares test problems for the Levenberg-Marquardt pin (tests/test_lm_minpack.py, tests/golden/make_golden_lm.py).

The classic MINPACK test set (More, Garbow, Hillstrom 1981), restricted to the problems whose residuals need only
+ - * / so that they evaluate to the same float32 bits on every machine, plus a clamping residual that modifies its
argument in place the way lm_forward_step does (minimizer_engine.f90:820-842).  Each problem is
(name, m, n, x0, f) with f(x[n] float32) -> fvec[m] float32; f may change x."""
import numpy as np

F = np.float32


def _rosenbrock(x):
    return np.array([F(10) * (x[1] - x[0] * x[0]), F(1) - x[0]], F)


def _powell_singular(x):
    return np.array([x[0] + F(10) * x[1], F(np.sqrt(F(5))) * (x[2] - x[3]), (x[1] - F(2) * x[2]) * (x[1] - F(2) * x[2]),
                     F(np.sqrt(F(10))) * (x[0] - x[3]) * (x[0] - x[3])], F)


def _freudenstein_roth(x):
    return np.array([F(-13) + x[0] + ((F(5) - x[1]) * x[1] - F(2)) * x[1],
                     F(-29) + x[0] + ((F(1) + x[1]) * x[1] - F(14)) * x[1]], F)


_BARD_Y = np.array([0.14, 0.18, 0.22, 0.25, 0.29, 0.32, 0.35, 0.39, 0.37, 0.58, 0.73, 0.96, 1.34, 2.10, 4.39], F)


def _bard(x):
    out = np.zeros(15, F)
    for i in range(15):
        u, v = F(i + 1), F(15 - i)
        w = min(u, v)
        out[i] = _BARD_Y[i] - (x[0] + u / (v * x[1] + w * x[2]))
    return out


_KO_Y = np.array([0.1957, 0.1947, 0.1735, 0.1600, 0.0844, 0.0627, 0.0456, 0.0342, 0.0323, 0.0235, 0.0246], F)
_KO_U = np.array([4.0, 2.0, 1.0, 0.5, 0.25, 0.167, 0.125, 0.1, 0.0833, 0.0714, 0.0625], F)


def _kowalik_osborne(x):
    u = _KO_U
    return (_KO_Y - x[0] * (u * u + u * x[1]) / (u * u + u * x[2] + x[3])).astype(F)


def _wood(x):
    return np.array([F(10) * (x[1] - x[0] * x[0]), F(1) - x[0], F(np.sqrt(F(90))) * (x[3] - x[2] * x[2]), F(1) - x[2],
                     F(np.sqrt(F(10))) * (x[1] + x[3] - F(2)), (x[1] - x[3]) / F(np.sqrt(F(10)))], F)


def _beale(x):
    y = np.array([1.5, 2.25, 2.625], F)
    out = np.zeros(3, F)
    p = x[1]
    for i in range(3):
        out[i] = y[i] - x[0] * (F(1) - p)
        p = p * x[1]
    return out


def _watson(x, m=31):
    n = len(x)
    out = np.zeros(m, F)
    for i in range(29):
        t = F(i + 1) / F(29)
        s1, dx = F(0), F(1)
        for j in range(1, n):
            s1 = s1 + F(j) * dx * x[j]
            dx = dx * t
        s2, dx = F(0), F(1)
        for j in range(n):
            s2 = s2 + dx * x[j]
            dx = dx * t
        out[i] = s1 - s2 * s2 - F(1)
    out[29] = x[0]
    out[30] = x[1] - x[0] * x[0] - F(1)
    return out


def _linear_full_rank(x, m=10):
    n = len(x)
    s = F(0)
    for j in range(n):
        s = s + x[j]
    t = F(2) * s / F(m) + F(1)
    out = np.full(m, -t, F)
    out[:n] = x - t
    return out.astype(F)


def _linear_rank1(x, m=10):
    n = len(x)
    s = F(0)
    for j in range(n):
        s = s + F(j + 1) * x[j]
    return np.array([F(i + 1) * s - F(1) for i in range(m)], F)


def _brown_almost_linear(x):
    n = len(x)
    s = F(-(n + 1))
    prod = F(1)
    for j in range(n):
        s = s + x[j]
        prod = prod * x[j]
    out = (x + s).astype(F)
    out[n - 1] = prod - F(1)
    return out


def _chebyquad(x, m=8):
    n = len(x)
    out = np.zeros(m, F)
    for j in range(n):
        t1, t2 = F(1), F(2) * x[j] - F(1)
        t = F(2) * t2
        for i in range(m):
            out[i] = out[i] + t2
            th = t * t2 - t1
            t1, t2 = t2, th
    dx = F(1) / F(n)
    iev = -1
    for i in range(m):
        out[i] = dx * out[i]
        if iev > 0:
            out[i] = out[i] + F(1) / (F(i + 1) * F(i + 1) - F(1))
        iev = -iev
    return out


def _clamped_rosenbrock(x):
    """limits as lm_forward_step applies them: x is moved back inside IN PLACE, the residuals carry the penalty"""
    lo, hi = np.array([-1.5, 0.2], F), np.array([0.8, 3.0], F)
    penalty = F(0)
    for i in range(2):
        if x[i] < lo[i]:
            penalty = penalty + abs(x[i] - lo[i]) / abs(hi[i] - lo[i])
            x[i] = lo[i]
        if x[i] > hi[i]:
            penalty = penalty + abs(x[i] - hi[i]) / abs(hi[i] - lo[i])
            x[i] = hi[i]
    return (_rosenbrock(x) * (F(1) + penalty)).astype(F)


def _scaled(f, scale):
    return lambda x: f((x * scale).astype(F))


PROBLEMS = [
    ("rosenbrock", 2, 2, [-1.2, 1.0], _rosenbrock),
    ("rosenbrock_x10", 2, 2, [-12.0, 10.0], _rosenbrock),
    ("powell_singular", 4, 4, [3.0, -1.0, 0.0, 1.0], _powell_singular),
    ("freudenstein_roth", 2, 2, [0.5, -2.0], _freudenstein_roth),
    ("bard", 15, 3, [1.0, 1.0, 1.0], _bard),
    ("kowalik_osborne", 11, 4, [0.25, 0.39, 0.415, 0.39], _kowalik_osborne),
    ("wood", 6, 4, [-3.0, -1.0, -3.0, -1.0], _wood),
    ("beale", 3, 2, [1.0, 1.0], _beale),
    ("watson6", 31, 6, [0.0] * 6, _watson),
    ("watson9", 31, 9, [0.0] * 9, _watson),
    ("linear_full_rank", 10, 5, [1.0] * 5, _linear_full_rank),
    ("linear_rank1", 10, 5, [1.0] * 5, _linear_rank1),
    ("brown_almost_linear", 7, 7, [0.5] * 7, _brown_almost_linear),
    ("chebyquad", 8, 8, [(j + 1) / 9.0 for j in range(8)], _chebyquad),
    ("clamped_rosenbrock", 2, 2, [-1.2, 1.0], _clamped_rosenbrock),
    ("zero_start", 3, 2, [0.0, 0.0], _beale),
]

# (ftol, xtol, gtol, maxfev factor, epsfcn, mode, factor): minimize_lm's settings (minimizer_engine.f90:778-790) and
# lmdif1's defaults (sminpack/lmdif1.f)
SETTINGS = {
    "minimize_lm": dict(gtol=0.0, maxfev=500, epsfcn=0.0, mode=2, factor=0.01),
    "lmdif1": dict(gtol=0.0, maxfev=200, epsfcn=0.0, mode=1, factor=100.0),
}
TOL = float(np.sqrt(F(1.192091e-07)))


def run_product(L, klib, name, m, n, x0, f, st):
    """kiwi_hip_lmdif (batched callback) -> (x, fvec, info, nfev)"""
    import ctypes as C
    x = np.array(x0, F)
    fvec = np.zeros(m, F)
    diag = np.ones(n, F)

    def cb(user, k, m_, n_, xs, fv):
        xa = np.ctypeslib.as_array(xs, (k, n_))
        fa = np.ctypeslib.as_array(fv, (k, m_))
        for i in range(k):
            fa[i] = f(xa[i])
        return 0

    info, nfev = C.c_int(), C.c_int()
    fp = lambda a: a.ctypes.data_as(klib.c_float_p)
    rc = L.kiwi_hip_lmdif(klib.RESIDUAL_FN(cb), None, m, n, fp(x), fp(fvec), TOL, TOL, st["gtol"], st["maxfev"] * (n + 1),
                          st["epsfcn"], fp(diag), st["mode"], st["factor"], C.byref(info), C.byref(nfev))
    assert rc == 0
    return x, fvec, info.value, nfev.value


def run_reference(R, name, m, n, x0, f, st):
    """the reference's sminpack lmdif through oracle/_ref (one point per callback)"""
    import ctypes as C
    x = np.array(x0, F)
    fvec = np.zeros(m, F)
    diag = np.ones(n, F)
    CB = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))

    def cb(m_, n_, xp, fv):
        xa = np.ctypeslib.as_array(xp, (n_,))
        np.ctypeslib.as_array(fv, (m_,))[:] = f(xa)
        return 0

    info, nfev = C.c_int(), C.c_int()
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    cbk = CB(cb)
    R.ref_lmdif.restype = None
    R.ref_lmdif.argtypes = [CB, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float,
                            C.c_int, C.c_float, C.POINTER(C.c_float), C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    R.ref_lmdif(cbk, m, n, fp(x), fp(fvec), TOL, TOL, st["gtol"], st["maxfev"] * (n + 1), st["epsfcn"], fp(diag), st["mode"],
                st["factor"], C.byref(info), C.byref(nfev))
    return x, fvec, info.value, nfev.value
