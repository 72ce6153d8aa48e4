"""End-to-end behaviour of the oracle engine on a small synthetic setup (CPU)."""
import numpy as np
import pytest

from kiwi_amd import synthetic
from tests.common import Scenario, oracle_misfits


@pytest.mark.parametrize("bilinear", [False, True])
def test_true_source_has_zero_misfit_and_perturbations_grow(bilinear):
    sc = Scenario(bilinear=bilinear)
    e = sc.oracle()
    sc.make_references(e)
    sc.apply_setup(e, True)
    trials = np.vstack([sc.true_params[None], synthetic.bilat_strike_sweep(4, step=1.0)])
    m, n, g = oracle_misfits(e, 1, trials)
    assert m.shape == (5, sc.nrec * 3)
    assert np.all(m[0] == 0.0) and g[0] == 0.0
    assert np.all(n[0] > 0)
    assert np.all(np.diff(g) > 0), g          # misfit grows with the strike perturbation
    # global misfit formula, minimizer_engine.f90:936-942
    expect = np.sqrt((m.astype(np.float64) ** 2).sum(1)) / np.sqrt((n.astype(np.float64) ** 2).sum(1))
    assert np.allclose(g, expect, rtol=1e-5)


def test_threads_do_not_change_results():
    sc = Scenario()
    e1 = sc.oracle(1)
    sc.make_references(e1)
    sc.apply_setup(e1, True)
    trials = synthetic.bilat_strike_sweep(3, step=2.0)
    a = oracle_misfits(e1, 1, trials)
    e4 = sc.oracle(4)
    sc.apply_setup(e4, True)
    b = oracle_misfits(e4, 1, trials)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_disabled_receiver_is_skipped():
    sc = Scenario()
    e = sc.oracle()
    sc.make_references(e)
    sc.apply_setup(e, True)
    e.set_source_params(1, synthetic.bilat_strike_sweep(1, 3.0)[0])
    m0, n0, _ = e.get_misfits()
    e.switch_receiver(2, False)
    m1, n1, _ = e.get_misfits()
    assert len(m1) == len(m0) - 3
    assert np.array_equal(m1, np.delete(m0, [3, 4, 5]))


def test_pack_gfdb_is_the_oracles_trace_pack():
    """kiwi_amd.synthetic.pack_gfdb (what bench.py hands the product) == the spans and samples the oracle's trace_pack
    restatement stores for the same dense array (first / last non-zero sample plus one zero, all-zero traces, interior gaps)."""
    import numpy as np
    from kiwi_amd import synthetic
    from tests.common import Scenario
    for variant in ("probe", "static"):
        sc = Scenario(variant=variant, L=300)
        sc.gf["data"][2, 1, 3, :] = 0.0
        sc.gf["data"][3, 1, 4, :7] = 0.0
        sc.gf["data"][4, 1, 4, -9:] = 0.0
        sc.oracle()
        f, n, d = sc.odb.dense_tables()
        g = synthetic.pack_gfdb(sc.gf)
        assert np.array_equal(f, g["first"]) and np.array_equal(n, g["nsamp"]) and np.array_equal(d, g["data"])
        assert n.min() == 1 and n.max() == 300
