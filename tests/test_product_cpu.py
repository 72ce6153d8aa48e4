"""CPU-side checks of the product: the C-ABI library loads, exports every symbol the public header
declares, fails loudly without a GPU, and its HOST-side pieces (source discretisers) agree bit for
bit with the oracle.  No device compute here."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from kiwi_amd import lib as klib
from kiwi_amd import engine as keng
from oracle import ko


def test_library_builds_and_exports_declared_symbols():
    klib.build()
    assert os.path.exists(klib.LIB_PATH)
    L = C.CDLL(klib.LIB_PATH)
    names = klib.declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(L, name), "missing export " + name


def test_loader_signatures_cover_header():
    L = klib.load()
    for name in klib.declared_symbols():
        assert getattr(L, name).argtypes is not None, name


@pytest.mark.skipif(torch.cuda.is_available(), reason="this box has a GPU")
def test_init_fails_loudly_without_gpu():
    with pytest.raises(klib.KiwiHipError):
        keng.Engine(0)


def test_effective_cpus_follows_the_cgroup_quota():
    """The discretiser's team size: allowed hardware threads cut to the container's CPU quota (cgroup v2 cpu.max or v1
    cfs quota) -- a container sees every hardware thread of its host but is throttled at the quota."""
    n = klib.load().kiwi_hip_effective_cpus()
    allowed = len(os.sched_getaffinity(0))
    assert 1 <= n <= allowed
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except OSError:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except OSError:
            pass
    if quota is not None:
        assert n == max(1, min(allowed, int(quota + 0.5)))
    else:
        assert n == allowed


def test_loader_puts_torch_first():
    """One HIP runtime per process: the library is loaded after torch (whose libamdhip64 it then binds to); see lib.load."""
    import importlib.util
    import os
    import sys
    klib.load()
    if importlib.util.find_spec("torch") is not None and os.environ.get("KIWI_HIP_WITHOUT_TORCH", "0") != "1":
        assert "torch" in sys.modules                    # (a machine without torch has nothing to order)


def test_source_nparams():
    L = klib.load()
    assert L.kiwi_hip_source_nparams(1) == 14
    assert L.kiwi_hip_source_nparams(2) == 11
    assert L.kiwi_hip_source_nparams(6) == 11
    assert L.kiwi_hip_source_nparams(99) < 0


def test_host_discretisers_match_oracle_bitwise():
    rng = np.random.default_rng(21)
    for _ in range(80):
        p = [rng.uniform(-2, 2), rng.uniform(-5e3, 5e3), rng.uniform(-5e3, 5e3), rng.uniform(2e3, 3e4),
             10 ** rng.uniform(17, 20), rng.uniform(-180, 180), rng.uniform(0, 90), rng.uniform(-180, 180),
             rng.uniform(-180, 180), rng.uniform(0, 2e4), rng.uniform(0, 1e4), rng.uniform(0, 1e4),
             rng.uniform(1500, 4000), rng.uniform(0, 3)]
        edt = float(rng.choice([0.5, 1.0, 2.0]))
        cases = [(1, p), (2, p[:8] + [rng.uniform(0, 1.5e4), p[12], p[13]]),
                 (6, p[:4] + list(rng.standard_normal(6) * 1e18) + [rng.uniform(0.01, 4)]),
                 (3, p[:5] + list(rng.standard_normal(6)) + [rng.uniform(1, 40), rng.uniform(2, 30)])]
        for st, par in cases:
            a, mo_a, ri_a = keng.discretize(st, par, edt)
            b, mo_b, ri_b, _ = ko.discretize(st, par, edt)
            assert a.shape == b.shape
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
            assert mo_a == mo_b and ri_a == ri_b


def test_discretize_rejects_bad_input():
    with pytest.raises(klib.KiwiHipError):
        keng.discretize(1, [0.0] * 5, 0.5)
    with pytest.raises(klib.KiwiHipError):
        keng.discretize(3, [0.0] * 12, 0.5)
    with pytest.raises(klib.KiwiHipError):
        keng.discretize(7, [0.0] * 13, 0.5)


def test_make_global_misfits_hand_vectors():
    # seismosizer.py:843-922, hand-derived: 2 sources x 3 receivers x 2 components
    m = np.array([[[3., 4.], [0., 0.], [1., 0.]], [[0., 0.], [6., 8.], [0., 2.]]])
    n = np.array([[[5., 0.], [1., 0.], [0., 1.]], [[5., 0.], [1., 0.], [0., 1.]]])
    g, msr = keng.make_global_misfits(m, n, "l2norm")
    assert np.allclose(msr, [[5., 0., 1.], [0., 10., 2.]])
    assert np.allclose(g, [np.sqrt(26. / 27.), np.sqrt(104. / 27.)])
    g1, msr1 = keng.make_global_misfits(m, n, "l1norm")
    assert np.allclose(msr1, [[7., 0., 1.], [0., 14., 2.]])
    assert np.allclose(g1, [8. / 7., 16. / 7.])
    gw, _ = keng.make_global_misfits(m, n, "l2norm", receiver_weights=np.array([1., 0., 2.]))
    assert np.allclose(gw, [np.sqrt((25. + 4.) / (25. + 4.)), np.sqrt(16. / 29.)])
    gz, _ = keng.make_global_misfits(m, n * 0, "l2norm")
    assert np.all(np.isnan(gz))


def test_this_image_runs_every_conditional_branch():
    """The HDF5 database reader (libkiwi_gfdb.so) and the Fortran side of the boundary (binding smoke, protocol host) are
    built only where their toolchains exist; on the build image and the GPU box they do, and their tests must have run
    there -- a missing toolchain is a failure here unless KIWI_TEST_ALLOW_MISSING=1 says it is expected."""
    import os
    from tests.common import HAVE_HDF5, HAVE_FLANG
    if os.environ.get("KIWI_TEST_ALLOW_MISSING"):
        return
    assert HAVE_HDF5, "HDF5 C headers missing: tests/test_gfdb_hdf5.py and the protocol host's HDF5 set_database did not run"
    assert HAVE_FLANG, "amdflang missing: tests/test_fortran_binding.py and tests/test_protocol_host.py did not run"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import __graft_entry__  # noqa: F401  (build() is what makes these)
    for f in ("kiwi_amd/libkiwi_hip.so", "kiwi_amd/libkiwi_gfdb.so", "kiwi_amd/fortran/minimizer_hip", "kiwi_amd/fortran/binding_smoke",
              "oracle/libko.so"):
        assert os.path.exists(os.path.join(root, f)), f + " not built: run `python -c 'import __graft_entry__ as g; g.build()'`"
