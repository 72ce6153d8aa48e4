"""Trial-source grids and the brute-force grid search on top of the batched engine: the Python-3
counterpart of python/tunguska/source.py:119-164 (`Source.grid`) and gridsearch.py:19-23,111-289
(`mimainc_to_gvals`, `MisfitGrid` with bootstrap).  The ordering defined here is the shard ordering of
kiwi_amd/shard.py: the first grid parameter varies slowest."""
import numpy as np

from .engine import SOURCE_TYPES, make_global_misfits
from .lib import KiwiHipError

# wire order of set_source_params (source_bilat.f90:93-106, source_circular.f90:92-102, source_eikonal.f90:97-114,
# source_mt_eikonal.f90:102-124, source_moment_tensor.f90:90-100)
SOURCE_PARAMS = {
    "bilateral": ["time", "north-shift", "east-shift", "depth", "moment", "strike", "dip", "slip-rake", "rupture-rake",
                  "length-a", "length-b", "width", "rupture-velocity", "rise-time"],
    "circular": ["time", "north-shift", "east-shift", "depth", "moment", "strike", "dip", "slip-rake", "radius",
                 "rupture-velocity", "rise-time"],
    "eikonal": ["time", "north-shift", "east-shift", "depth", "moment", "strike", "dip", "slip-rake", "bord-shift-x",
                "bord-shift-y", "bord-radius", "nukl-shift-x", "nukl-shift-y", "rel-rupture-velocity", "rise-time"],
    "mt_eikonal": ["time", "north-shift", "east-shift", "depth", "moment-factor", "strike", "dip", "bord-shift-x",
                   "bord-shift-y", "bord-radius", "nukl-shift-x", "nukl-shift-y", "rel-rupture-velocity", "mxx", "myy",
                   "mzz", "mxy", "mxz", "myz", "rise-time"],
    "moment_tensor": ["time", "north-shift", "east-shift", "depth", "mxx", "myy", "mzz", "mxy", "mxz", "myz", "rise-time"],
    "point_lp": ["time", "north-shift", "east-shift", "depth", "moment", "m_xx", "m_yy", "m_zz", "m_xy", "m_xz", "m_yz",
                 "excitation-time", "main-period"],                                     # source_point_lp.f90:110-122
}


def mimainc_to_gvals(mi, ma, inc):
    """gridsearch.py:19-23: n = round((max-min)/inc)+1 equally spaced values including both ends."""
    vmin, vmax, vinc = float(mi), float(ma), float(inc)
    n = int(round((vmax - vmin) / vinc)) + 1
    if n == 1:
        return np.array([vmin])
    vinc = (vmax - vmin) / (n - 1)
    return np.array([vmin + i * vinc for i in range(n)], np.float64)


def source_grid(sourcetype, base_params, grid_definition, source_constraints=None):
    """source.py:119-164: all combinations of [(parameter, values), ...] applied to the base source, first
    parameter slowest; `source_constraints(dict)` may switch grid nodes off.  Returns params[n, nparams]
    (float32, wire order)."""
    names = SOURCE_PARAMS[sourcetype]
    base = np.asarray(base_params, np.float64)
    if base.shape != (len(names),):
        raise KiwiHipError("%s takes %d parameters" % (sourcetype, len(names)))
    cols = []
    for key, _ in grid_definition:
        if key not in names:
            raise KiwiHipError("unknown parameter '%s' for source type %s" % (key, sourcetype))
        cols.append(names.index(key))
    vals = [np.asarray(v, np.float64).ravel() for _, v in grid_definition]
    if not vals:
        return np.zeros((0, len(names)), np.float32)
    mesh = np.meshgrid(*vals, indexing="ij")                 # first parameter slowest
    n = mesh[0].size
    out = np.tile(base, (n, 1))
    for c, mg in zip(cols, mesh):
        out[:, c] = mg.ravel()
    if source_constraints is not None:
        keep = [bool(source_constraints(dict(zip(names, row)))) for row in out]
        out = out[np.array(keep, bool)]
    return out.astype(np.float32)


class MisfitGridStats:
    """gridsearch.py:45-105 (the numbers, not the plots)."""

    def __init__(self, paramname, best, distribution, tested_values=None):
        self.paramname, self.best, self.tested_values = paramname, best, tested_values
        self.distribution = np.asarray(distribution, np.float64)
        self.mean = float(self.distribution.mean()) if self.distribution.size else float("nan")
        self.std = float(self.distribution.std()) if self.distribution.size else float("nan")

    def converged(self):
        return self.distribution.size > 0 and np.all(self.distribution == self.distribution[0])


class MisfitGrid:
    """Brute force grid search minimizer with builtin bootstrapping (gridsearch.py:111-289)."""

    def __init__(self, sourcetype, base_params, param_ranges=None, param_values=None, source_constraints=None,
                 ref_params=None):
        self.sourcetype = sourcetype
        self.base_params = np.asarray(base_params, np.float32)
        self.ref_params = self.base_params if ref_params is None else np.asarray(ref_params, np.float32)
        if param_values is not None:
            self.param_values = [(p, np.asarray(v, np.float64)) for p, v in param_values]
        else:
            self.param_values = [(p, mimainc_to_gvals(mi, ma, inc)) for p, mi, ma, inc in param_ranges]
        self.sources = source_grid(sourcetype, self.base_params, self.param_values, source_constraints)
        self.sourceparams = [p for p, _ in self.param_values]
        self.misfits_by_src = self.norms_by_src = None
        self.failings = []
        self.best_source = self.misfits_by_s = self.misfits_by_r = self.variability_by_r = None
        self.bootstrap_sources = self.stats = None

    def compute(self, engine, dist=None, device=0):
        """Trace misfits for every grid node (and the reference source), `engine` = kiwi_amd.Engine set up for
        the inversion.  With a torch.distributed group the grid is sharded over the ranks (kiwi_amd/shard.py)."""
        self.receiver_mask = np.array(engine.enabled, bool)
        self.nreceivers = len(engine.components)
        if len(self.sources):
            if dist is not None:
                from .shard import sharded_misfits_for_sources
                self.misfits_by_src, self.norms_by_src, self.failings = sharded_misfits_for_sources(
                    engine, self.sourcetype, self.sources, dist, device)
            else:
                self.misfits_by_src, self.norms_by_src, self.failings = engine.make_misfits_for_sources(self.sourcetype,
                                                                                                      self.sources)
        # sources the engine rejected keep zero misfits AND zero norms: their global misfit comes out as NaN and
        # nanargmin passes over them (gridsearch.py:172-178 drops `failings` the same way)
        self.ref_misfits_by_src, self.ref_norms_by_src, _ = engine.make_misfits_for_sources(self.sourcetype,
                                                                                          self.ref_params[None, :])
        self.best_source = None

    def _best_source(self, **cfg):
        g, g_sr = make_global_misfits(self.misfits_by_src, self.norms_by_src, receiver_mask=self.receiver_mask, **cfg)
        ibest = int(np.nanargmin(g)) if np.any(np.isfinite(g)) else 0
        return ibest, g, g_sr

    def postprocess(self, bootstrap_iterations=1000, rng=None, **outer_misfit_config):
        """Global misfits, best source, bootstrap distribution of the best source (gridsearch.py:199-289)."""
        g, g_sr = make_global_misfits(self.ref_misfits_by_src, self.ref_norms_by_src, receiver_mask=self.receiver_mask,
                                      **outer_misfit_config)
        self.ref_misfit, self.ref_misfits_by_r = g[0], g_sr[0]
        if len(self.sources) == 0:
            self.best_source, self.misfits_by_s, self.misfits_by_r, self.variability_by_r = self.base_params, [], [], []
            self.bootstrap_sources, self.stats = [], {}
            return
        ibest, g, g_sr = self._best_source(**outer_misfit_config)
        self.ibest, self.best_source, self.misfits_by_s = ibest, self.sources[ibest], g
        self.misfits_by_r, self.variability_by_r = g_sr[ibest], np.std(g_sr, 0)
        rng = np.random.default_rng() if rng is None else rng
        self.bootstrap_sources = [self.sources[self._best_source(bootstrap=True, rng=rng, **outer_misfit_config)[0]]
                                  for _ in range(bootstrap_iterations)]
        names = SOURCE_PARAMS[self.sourcetype]
        self.stats = {}
        for param, gvalues in self.param_values:
            k = names.index(param)
            self.stats[param] = MisfitGridStats(param, float(self.best_source[k]),
                                                [s[k] for s in self.bootstrap_sources], tested_values=gvalues)

    def get_best_misfit(self):
        return self.ref_misfit if len(self.misfits_by_s) == 0 else float(np.nanmin(self.misfits_by_s))
