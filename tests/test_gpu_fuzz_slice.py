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


# cases the long runs have found (seed, index) -- each failed on the library of its day:
#   5202 / 463, 5608, 6585 (round 5, `fused`, ng = 8): accumulate_multi_kernel's compiled apply started its rotated sums from the first
#   product, (c1 x) + (c2 y), which the contracted build may fuse either way -- a batch evaluated in pieces of 1 (another kernel) no longer
#   gave the bits of the whole batch
FOUND = [(5202, 463), (5202, 5608), (5202, 6585)]


@pytest.mark.parametrize("seed,index", FOUND)
def test_fuzz_found_cases(seed, index):
    assert fz.one_case(np.random.default_rng([seed, index]), verbose=False), "replay with: python tests/fuzz_gpu_parity.py case %d %d" % (seed, index)
