"""A fixed slice of the randomised parity sweep (tests/fuzz_gpu_parity.py) under pytest, so that the driver's GPU run sees
it: 300 seeded cases -- random databases (8 / 10 components, 96-2300 samples, gaps and static end values), receiver sets (all
component letters, depths), nearest / bilinear interpolation and under-sampling, all six source types, traces missing from
the database, tapers, every norm (time-domain, floating, amplitude-spectrum, frequency-filtered), the synthetics factor,
repeated trials and the one-call in random pieces -- each against the CPU oracle at the tolerances of tests/common.py /
tests/test_gpu_parity.py.  Replay one case: python tests/fuzz_gpu_parity.py case <seed> <index>."""
import numpy as np
import pytest

from tests import fuzz_gpu_parity as fz

pytestmark = pytest.mark.gpu
SEED = 20261003


@pytest.mark.parametrize("block", range(6))
def test_fuzz_slice(block):
    bad = [n for n in range(50 * block, 50 * block + 50) if not fz.one_case(np.random.default_rng([SEED, n]), verbose=False)]
    assert not bad, "replay with: python tests/fuzz_gpu_parity.py case %d <index>, indices %s" % (SEED, bad)
