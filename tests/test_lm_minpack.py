"""Levenberg-Marquardt pin: the product's fp32 lmdif with batched forward differences (kiwi_amd/csrc/kiwi_host_lm.hpp,
C-ABI kiwi_hip_lmdif) against the reference's own sminpack -- committed outputs (tests/golden/lm_vectors.npz, made by
tests/golden/make_golden_lm.py) and, where oracle/_ref is present, the live library.  Bit for bit: x, residuals, info
and the number of function evaluations.  No device involved."""
import os
import warnings

import numpy as np
import pytest

import lm_problems as P
from kiwi_amd import lib as klib
from kiwi_amd import lm

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "lm_vectors.npz"))
CASES = [(s, p[0]) for s in P.SETTINGS for p in P.PROBLEMS]
BYNAME = {p[0]: p for p in P.PROBLEMS}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("sname,pname", CASES)
def test_lmdif_matches_the_reference_minpack_golden(sname, pname):
    warnings.simplefilter("ignore")
    name, m, n, x0, f = BYNAME[pname]
    x, fvec, info, nfev = P.run_product(klib.load(), klib, name, m, n, x0, f, P.SETTINGS[sname])
    key = sname + "/" + pname
    assert [info, nfev] == list(G[key + "/info_nfev"])
    assert np.array_equal(bits(x), bits(G[key + "/x"]))
    assert np.array_equal(bits(fvec), bits(G[key + "/fvec"]))


def test_lmdif_matches_the_live_reference_build():
    from oracle import ko
    R = ko.ref()
    if R is None or not hasattr(R, "ref_lmdif"):
        pytest.skip("oracle/_ref not built")
    warnings.simplefilter("ignore")
    rng = np.random.default_rng(5)
    for sname, st in P.SETTINGS.items():
        for name, m, n, x0, f in P.PROBLEMS:
            for trial in range(3):                     # the documented start and two random ones
                start = np.array(x0, np.float32) if trial == 0 else (np.array(x0, np.float32) + rng.normal(0, 0.3, n)).astype(np.float32)
                a = P.run_product(klib.load(), klib, name, m, n, start, f, st)
                b = P.run_reference(R, name, m, n, start, f, st)
                assert a[2:] == b[2:], (sname, name, trial)
                assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[1]), bits(b[1])), (sname, name, trial)


def test_python_face_and_error_paths():
    x, fvec, info, nfev = lm.lmdif(lambda xs: np.stack([BYNAME["rosenbrock"][4](x) for x in xs]), [-1.2, 1.0], 2)
    assert info in (1, 2, 3, 4) and np.allclose(x, [1, 1], atol=1e-5) and nfev > 5
    with pytest.raises(ZeroDivisionError):             # an exception in the residual aborts the run and resurfaces
        lm.lmdif(lambda xs: 1 / 0, [1.0, 1.0], 3)
    # improper input (m < n): info 0, nothing evaluated
    x, fvec, info, nfev = lm.lmdif(lambda xs: np.zeros((len(xs), 1), np.float32), [1.0, 2.0], 1)
    assert info == 0 and nfev == 0
