import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "exact_only: a statement about the bit-exact arithmetic contract only (not run under KIWI_HIP_ARITH=fused)")


# Every GPU test runs under BOTH arithmetic contracts of the accumulate kernels (include/kiwi_hip.h KIWI_ARITH_*): `exact`
# (the default: bit-identical to the oracle given equal geometry) and `fused` (multiply + add contracted into one FMA:
# tolerance class).  The mode reaches the library -- and the Fortran protocol host the tests start -- through the environment
# (KIWI_HIP_ARITH, read by kiwi_hip_init); tests/common.py's comparison helpers read it back: bit-identity statements become
# "within 2e-6 of the trace maximum" / "within 1e-6 of max(misfit, norm factor)" under `fused`.
def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("gpu") is None:
        return
    modes = ["exact"] if metafunc.definition.get_closest_marker("exact_only") else ["exact", "fused"]
    if "_arith" not in metafunc.fixturenames:
        metafunc.fixturenames.append("_arith")
    metafunc.parametrize("_arith", modes, indirect=True)


@pytest.fixture
def _arith(request, monkeypatch):
    mode = getattr(request, "param", "exact")
    monkeypatch.setenv("KIWI_HIP_ARITH", mode)
    return mode


@pytest.fixture(autouse=True)
def _arith_default(request, monkeypatch):
    if "_arith" not in request.fixturenames:
        monkeypatch.setenv("KIWI_HIP_ARITH", "exact")


@pytest.fixture(scope="session")
def ko():
    from oracle import ko as _ko
    _ko.lib()
    return _ko
