"""`minimize_lm` (minimizer_engine.f90:728-874) on top of the batched engine: Levenberg-Marquardt over the masked,
normalised source parameters with the per-receiver-component misfits as residual vector.

The reference calls `lmdif` of its single-precision MINPACK (`sminpack/`), whose forward-difference Jacobian costs
one forward evaluation per free parameter, one after the other.  The product library carries that algorithm in fp32,
operation for operation (`kiwi_amd/csrc/kiwi_host_lm.hpp`; pinned against the reference's own sminpack build on the
MINPACK test problems, tests/test_lm_minpack.py), with one change: the n perturbed sources of a Jacobian are ONE
batched device evaluation.  This module is the thin Python face of `kiwi_hip_minimize_lm` and
`kiwi_hip_lmdif`."""
import ctypes as C

import numpy as np

from . import lib as klib
from .engine import SOURCE_TYPES
from .gridsearch import SOURCE_PARAMS
from .lib import KiwiHipError

# psm_params_norm_* (source_bilat.f90:45-46, source_circular.f90:44-45, source_moment_tensor.f90:42-43,
# source_eikonal.f90:48-49, source_mt_eikonal.f90:48-50)
PARAMS_NORM = {
    "bilateral": [1., 10000., 10000., 10000., 7e18, 360., 90., 360., 360., 10000., 10000., 10000., 3000., 1.],
    "circular": [1., 10000., 10000., 10000., 7e18, 360., 90., 360., 10000., 3000., 1.],
    "moment_tensor": [1., 10000., 10000., 10000., 7e18, 7e18, 7e18, 7e18, 7e18, 7e18, 1.],
    "point_lp": [1., 10000., 10000., 10000., 7e18, 1., 0., -1., 1., 1., 1., 20., 1.],        # source_point_lp.f90:54-55 (as is)
    "eikonal": [1., 10000., 10000., 10000., 7e18, 360., 90., 360., 10000., 10000., 10000., 360., 10000., 1., 1.],
    "mt_eikonal": [1., 10000., 10000., 10000., 7e18, 360., 90., 10000., 10000., 10000., 360., 10000., 1., 7e18, 7e18, 7e18, 7e18,
                   7e18, 7e18, 1.],
}


class LmResult:
    """params: the source the engine is left with (the LAST forward step, as the reference's psm after minimize_lm);
    best: lmdif's accepted iterate; misfit: global misfit of `params`; iterations: forward steps; info: lmdif's."""
    def __init__(self, params, best, misfit, info, iterations):
        self.params, self.best, self.misfit, self.info, self.iterations = params, best, misfit, info, iterations


def minimize_lm(engine, sourcetype, params, mask, mins=None, maxs=None):
    """engine: kiwi_amd.Engine set up for the inversion; params: start values (wire order); mask: which parameters
    are free (`set_source_params_mask`, names or booleans); mins / maxs: optional limits of the FREE parameters in
    physical units (`set_source_subparams_limits`: outside them the residuals are multiplied by 1 + penalty and the
    parameter is clamped, minimizer_engine.f90:820-842)."""
    names = SOURCE_PARAMS[sourcetype]
    if sourcetype not in SOURCE_TYPES:
        raise KiwiHipError("unknown source type")
    p = np.ascontiguousarray(params, np.float32).copy()
    if len(p) != len(names):
        raise KiwiHipError("wrong number of source parameters")
    if len(mask) and isinstance(mask[0], str):
        free = np.array([n in set(mask) for n in names], np.int32)
    else:
        free = np.ascontiguousarray(np.asarray(mask, bool), np.int32)
    if len(free) != len(names):
        raise KiwiHipError("wrong number of elements in mask")
    n = int(free.sum())
    lo = hi = None
    if mins is not None or maxs is not None:
        lo, hi = np.ascontiguousarray(mins, np.float32), np.ascontiguousarray(maxs, np.float32)
        if len(lo) != n or len(hi) != n:
            raise KiwiHipError("wrong number of subparam_mins / subparam_maxs")
    fp = lambda a: None if a is None else a.ctypes.data_as(klib.c_float_p)
    info, it, mis = C.c_int(), C.c_int(), C.c_float()
    best = np.zeros_like(p)
    engine._ck(engine.L.kiwi_hip_minimize_lm(engine.h, SOURCE_TYPES[sourcetype], fp(p), free.ctypes.data_as(klib.c_int_p),
                                             fp(lo), fp(hi), C.byref(info), C.byref(it), C.byref(mis), fp(best)), "minimize_lm")
    return LmResult(p, best, float(mis.value), int(info.value), int(it.value))


def lmdif(residuals, x0, m, ftol=None, xtol=None, gtol=0.0, maxfev=None, epsfcn=0.0, diag=None, factor=100.0):
    """The optimiser by itself (sminpack/lmdif.f in fp32, batched forward differences) for any residual function:
    `residuals(xs[k, n]) -> fv[k, m]` (float32; xs may be modified in place).  Returns (x, fvec, info, nfev)."""
    L = klib.load()
    x = np.ascontiguousarray(x0, np.float32).copy()
    n = len(x)
    fvec = np.zeros(m, np.float32)
    tol = float(np.sqrt(np.float32(1.192091e-07)))
    state = {}

    def cb(user, k, m_, n_, xs, fv):
        try:
            xa = np.ctypeslib.as_array(xs, (k, n_))
            np.ctypeslib.as_array(fv, (k, m_))[:] = np.asarray(residuals(xa), np.float32).reshape(k, m_)
            return 0
        except Exception as exc:        # noqa: BLE001 -- must not unwind through the C frames
            state["exc"] = exc
            return -1

    mode = 1 if diag is None else 2
    d = np.ones(n, np.float32) if diag is None else np.ascontiguousarray(diag, np.float32).copy()
    info, nfev = C.c_int(), C.c_int()
    fp = lambda a: a.ctypes.data_as(klib.c_float_p)
    rc = L.kiwi_hip_lmdif(klib.RESIDUAL_FN(cb), None, m, n, fp(x), fp(fvec), tol if ftol is None else ftol,
                          tol if xtol is None else xtol, gtol, maxfev or 200 * (n + 1), epsfcn, fp(d), mode, factor,
                          C.byref(info), C.byref(nfev))
    if "exc" in state:
        raise state["exc"]
    if rc:
        raise KiwiHipError("lmdif failed")
    return x, fvec, int(info.value), int(nfev.value)
