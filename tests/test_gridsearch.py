"""P2/P3: grid construction order, outer norms with anarchy / bootstrap, MisfitGrid
bookkeeping -- pinned with hand-derived vectors (the reference's Python 2 modules cannot be imported)."""
import numpy as np
import pytest

from kiwi_amd import gridsearch as gs
from kiwi_amd.engine import make_global_misfits
from kiwi_amd.lib import KiwiHipError

BASE = [0., 0., 0., 10000., 1e20, 91., 87., 164., 0., 4000., 2000., 4000., 3000., 2.]


def test_mimainc_to_gvals():
    # gridsearch.py:19-23: n = round((max-min)/inc)+1, increment re-derived so that both ends are hit
    assert np.array_equal(gs.mimainc_to_gvals(0, 10, 5), [0., 5., 10.])
    v = gs.mimainc_to_gvals(0, 1, 0.3)
    assert len(v) == 4 and v[0] == 0 and v[-1] == 1 and np.allclose(np.diff(v), 1 / 3)
    assert np.array_equal(gs.mimainc_to_gvals(2, 2.1, 1), [2.])


def test_source_grid_first_parameter_slowest():
    g = gs.source_grid("bilateral", BASE, [("strike", [10., 20.]), ("dip", [1., 2., 3.])])
    assert g.shape == (6, 14) and g.dtype == np.float32
    assert list(g[:, 5]) == [10, 10, 10, 20, 20, 20]           # source.py:143-160: outer loop = first tuple
    assert list(g[:, 6]) == [1, 2, 3, 1, 2, 3]
    assert np.array_equal(g[:, [0, 1, 2, 3, 4, 7, 8, 9, 10, 11, 12, 13]],
                          np.tile(np.array(BASE, np.float32)[[0, 1, 2, 3, 4, 7, 8, 9, 10, 11, 12, 13]], (6, 1)))
    # constraints switch nodes off
    g2 = gs.source_grid("bilateral", BASE, [("strike", [10., 20.]), ("dip", [1., 2., 3.])],
                        source_constraints=lambda s: s["dip"] != 2.)
    assert list(g2[:, 6]) == [1, 3, 1, 3]
    with pytest.raises(KiwiHipError):
        gs.source_grid("bilateral", BASE, [("radius", [1.])])
    with pytest.raises(KiwiHipError):
        gs.source_grid("circular", BASE, [("radius", [1.])])
    assert gs.source_grid("bilateral", BASE, []).shape == (0, 14)


def test_outer_norms_mask_anarchy_bootstrap():
    m = np.array([[[3., 4.], [1., 0.], [0., 2.]],
                  [[0., 0.], [2., 2.], [1., 1.]]])
    n = np.array([[[6., 8.], [2., 0.], [0., 4.]],
                  [[6., 8.], [2., 0.], [0., 4.]]])
    # anarchy, l2 (seismosizer.py:896-899): weight_r = 1 / n_sr  ->  m_sr/n_sr = .5,.5,.5 ; every n_sr -> 1
    g, msr = make_global_misfits(m, n, "l2norm", anarchy=True)
    assert np.allclose(msr[0], [0.5, 0.5, 0.5]) and np.isclose(g[0], np.sqrt(0.75 / 3))
    assert np.allclose(msr[1], [0., np.sqrt(8) / 2, np.sqrt(2) / 4])
    # anarchy, l1 (:879-883)
    g1, msr1 = make_global_misfits(m, n, "l1norm", anarchy=True)
    assert np.allclose(msr1[0], [7 / 14, 1 / 2, 2 / 4]) and np.isclose(g1[0], 1.5 / 3)
    # bootstrap: counts over the enabled receivers only, sqrt for l2 (:901-902)
    mask = np.array([True, False, True])
    rng = np.random.default_rng(3)
    draw = np.random.default_rng(3).integers(0, 2, 2)
    bw = np.bincount(np.array([0, 2])[draw], minlength=3)
    assert bw.sum() == 2 and bw[1] == 0
    g2, msr2 = make_global_misfits(m, n, "l2norm", receiver_mask=mask, bootstrap=True, rng=rng)
    m_sr = np.sqrt((m ** 2).sum(2)) * np.sqrt(bw)
    n_sr = np.sqrt((n ** 2).sum(2)) * np.sqrt(bw)
    assert np.allclose(msr2, m_sr) and np.allclose(g2, np.sqrt((m_sr ** 2).sum(1) / (n_sr ** 2).sum(1)))
    g3, msr3 = make_global_misfits(m, n, "l1norm", receiver_mask=mask, bootstrap=True, rng=np.random.default_rng(3))
    assert np.allclose(msr3, m.sum(2) * bw)
    # a zero receiver weight also removes the receiver from the draw (:850-861)
    for seed in range(5):
        _, msr4 = make_global_misfits(m, n, "l1norm", receiver_weights=np.array([1., 1., 0.]), bootstrap=True,
                                      rng=np.random.default_rng(seed))
        assert np.all(msr4[:, 2] == 0)
    with pytest.raises(KiwiHipError):
        make_global_misfits(m, n, "l3norm")


class FakeEngine:
    """Stands in for kiwi_amd.Engine: misfit = |strike - 30| + |dip - 2| per receiver-component."""
    components = ["ned", "d", "ne"]
    enabled = [True, False, True]
    reject = False

    def make_misfits_for_sources(self, sourcetype, params):
        p = np.atleast_2d(params)
        base = np.abs(p[:, 5] - 30.) + np.abs(p[:, 6] - 2.)
        m = np.zeros((len(p), 3, 3))
        n = np.zeros((len(p), 3, 3))
        for ir, k in ((0, 3), (2, 2)):
            m[:, ir, :k] = base[:, None] * (1 + 0.1 * ir)
            n[:, ir, :k] = 10.
        fails = [int(i) for i in np.nonzero(p[:, 5] == 20.)[0]] if self.reject else []   # "Empty rupture area" stand-ins
        m[fails] = 0.
        n[fails] = 0.
        return m, n, fails


def test_misfit_grid_finds_minimum_and_bootstraps():
    mg = gs.MisfitGrid("bilateral", BASE, param_ranges=[("strike", 10, 50, 10), ("dip", 1, 3, 1)])
    assert len(mg.sources) == 15 and mg.sourceparams == ["strike", "dip"]
    mg.compute(FakeEngine())
    assert mg.misfits_by_src.shape == (15, 3, 3) and list(mg.receiver_mask) == [True, False, True]
    mg.postprocess(bootstrap_iterations=50, rng=np.random.default_rng(1), outer_norm="l2norm")
    assert mg.best_source[5] == 30. and mg.best_source[6] == 2. and mg.ibest == 7
    assert mg.get_best_misfit() == 0.0 and len(mg.bootstrap_sources) == 50
    assert mg.stats["strike"].best == 30. and mg.stats["strike"].converged() and mg.stats["dip"].mean == 2.
    assert mg.misfits_by_r.shape == (3,) and mg.variability_by_r.shape == (3,)
    assert mg.failings == []
    # trial sources the engine rejects (seismosizer.py:703-720): listed, zero rows, NaN global misfit, never the best
    fe = FakeEngine()
    fe.reject = True
    mf = gs.MisfitGrid("bilateral", BASE, param_ranges=[("strike", 10, 50, 10), ("dip", 1, 3, 1)])
    mf.compute(fe)
    mf.postprocess(bootstrap_iterations=5, rng=np.random.default_rng(1), outer_norm="l2norm")
    assert mf.failings == [3, 4, 5] and np.all(mf.misfits_by_src[3:6] == 0) and np.all(np.isnan(mf.misfits_by_s[3:6]))
    assert mf.ibest == 7
    # reference source misfit (gridsearch.py:266-271): base strike 91, dip 87
    assert mg.ref_misfit > 1.
    # empty grid falls back to the base source (gridsearch.py:204-209)
    e = gs.MisfitGrid("bilateral", BASE, param_values=[])
    e.compute(FakeEngine())
    e.postprocess()
    assert np.array_equal(e.best_source, np.array(BASE, np.float32)) and e.get_best_misfit() == e.ref_misfit


def test_pieces_of_the_one_call_evaluation_cover_the_list():
    """kiwi_amd.engine._pieces mirrors how kiwi_hip_misfits_for_params cuts a trial list (kiwi_hip.hip): pieces of `piece` sources in
    list order; for the eikonal types (4, 5) the last piece as half, a quarter, an eighth and an eighth of it -- the device starts
    after an eighth of a piece's fast-marching solves.  Whatever the cut: the pieces tile [0, n) in order, the head piece is
    [0, piece), no piece is empty, and the closed-form types are never ramped."""
    from kiwi_amd.engine import _pieces
    for st in (1, 4, 5, 6):
        for n in (1, 2, 7, 8, 9, 127, 128, 129, 200, 512, 1350):
            for piece in (1, 2, 8, 9, 16, 128, 2048):
                ps = _pieces(n, piece, st)
                assert ps[0][0] == 0 and ps[0][1] == min(piece, n)
                assert all(c > 0 for _, c in ps)
                assert all(a[0] + a[1] == b[0] for a, b in zip(ps, ps[1:])) and ps[-1][0] + ps[-1][1] == n
                if st not in (4, 5) or n <= piece:
                    assert ps == [(s0, min(piece, n - s0)) for s0 in range(0, n, piece)]
    assert _pieces(512, 128, 5) == [(0, 128), (128, 128), (256, 128), (384, 64), (448, 32), (480, 16), (496, 16)]
    assert _pieces(512, 128, 1) == [(0, 128), (128, 128), (256, 128), (384, 128)]
