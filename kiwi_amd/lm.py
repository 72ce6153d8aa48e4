"""`minimize_lm` (minimizer_engine.f90:722-874) on top of the batched engine: Levenberg-Marquardt over the masked,
normalised source parameters with the per-receiver-component misfits as residual vector.

The reference calls MINPACK's `lmdif` (single precision `sminpack`), whose forward-difference Jacobian costs one
forward evaluation per free parameter, one after the other.  Here MINPACK's `lmder` (through scipy, double precision)
gets the same forward-difference Jacobian from ONE batched device evaluation of the n + 1 perturbed sources
(SURVEY.md 8f item 4).  Same algorithm, tolerances and step rule (`fdjac2`: h = sqrt(eps) * |x_j|, or sqrt(eps) if
x_j = 0), so the iterates follow the reference's up to floating-point precision; they are not bit-identical to it."""
import numpy as np

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
EPS32 = float(np.finfo(np.float32).eps)          # spmpar(1) of the reference's single-precision MINPACK


class LmResult:
    def __init__(self, params, misfit, info, iterations, nbatches):
        self.params, self.misfit, self.info, self.iterations, self.nbatches = params, misfit, info, iterations, nbatches


def minimize_lm(engine, sourcetype, params, mask, mins=None, maxs=None, maxfev=None):
    """engine: kiwi_amd.Engine set up for the inversion; params: start values (wire order); mask: which parameters
    are free (`set_source_params_mask`, names or booleans); mins / maxs: optional bounds of the FREE parameters in
    physical units (`set_source_subparams_range`: outside them the residuals are multiplied by 1 + penalty and the
    parameter is clamped, minimizer_engine.f90:820-842).  Returns LmResult (params in physical units)."""
    from scipy.optimize import leastsq
    names = SOURCE_PARAMS[sourcetype]
    if sourcetype not in SOURCE_TYPES:
        raise KiwiHipError("unknown source type")
    p0 = np.asarray(params, np.float64).copy()
    if len(p0) != len(names):
        raise KiwiHipError("wrong number of source parameters")
    if len(mask) and isinstance(mask[0], str):
        free = np.array([n in set(mask) for n in names])
    else:
        free = np.asarray(mask, bool)
    idx = np.flatnonzero(free)
    n = len(idx)
    norm = np.asarray(PARAMS_NORM[sourcetype], np.float64)[idx]
    if n == 0:
        raise KiwiHipError("no free parameters")
    state = {"iterations": 0, "nbatches": 0}

    def clamp(sub):                                   # lm_forward_step, minimizer_engine.f90:820-842
        sub = np.array(sub, np.float64)
        penalty = 0.0
        if mins is not None and maxs is not None:
            lo, hi = np.asarray(mins, np.float64), np.asarray(maxs, np.float64)
            phys = sub * norm
            below, above = phys < lo, phys > hi
            penalty += np.sum(np.abs(phys - lo)[below] / np.abs(hi - lo)[below])
            penalty += np.sum(np.abs(phys - hi)[above] / np.abs(hi - lo)[above])
            sub = np.where(below, lo / norm, np.where(above, hi / norm, sub))
        return sub, penalty

    def residuals_batch(subs):
        """misfits[(len(subs), nmisfits)] of several normalised sub-parameter vectors in one device evaluation"""
        rows, pens = [], []
        for sub in subs:
            sub, pen = clamp(sub)
            p = p0.copy()
            p[idx] = sub * norm
            rows.append(p)
            pens.append(pen)
        engine.set_source_params(sourcetype, np.array(rows, np.float32))
        engine.eval()
        m, _, _ = engine.get_misfits()
        state["nbatches"] += 1
        state["iterations"] += len(subs)              # `iterations` counts forward evaluations (:866)
        return m.astype(np.float64) * (1.0 + np.array(pens))[:, None]

    def func(sub):
        return residuals_batch([sub])[0]

    def jac(sub):                                     # fdjac2 (MINPACK), all columns in one batch
        h = np.sqrt(EPS32) * np.abs(sub)
        h[h == 0.0] = np.sqrt(EPS32)
        subs = [sub] + [sub + h[j] * np.eye(n)[j] for j in range(n)]
        r = residuals_batch(subs)
        return ((r[1:] - r[0]) / h[:, None]).T        # (nmisfits, n)

    tol = float(np.sqrt(EPS32))
    nmis = engine.nmisfits()
    if nmis < n:
        raise KiwiHipError("fewer misfits than free parameters")
    x, _, infodict, _, info = leastsq(func, p0[idx] / norm, Dfun=jac, full_output=True, ftol=tol, xtol=tol, gtol=0.0,
                                      maxfev=maxfev or 500 * (n + 1), factor=0.01, diag=np.ones(n))
    if info == 8:
        info = 4                                      # minimizer_engine.f90:796
    x, _ = clamp(x)
    best = p0.copy()
    best[idx] = x * norm
    engine.set_source_params(sourcetype, best[None, :].astype(np.float32))
    engine.eval()
    _, _, g = engine.get_misfits()
    return LmResult(best.astype(np.float32), float(g[0]), int(info), state["iterations"], state["nbatches"])
