"""The spectral tolerances have a derivation (tests/common.py fft_roundoff_bound) instead of fixed figures; this test pins the
derivation on the CPU: the oracle evaluates the same trial under the same spectral norm / frequency filter twice -- with its
fp64 transform (an exact DFT rounded once) and with a textbook fp32 radix-2 FFT (oracle/ko_comparator.c fft_c32, the second
checker) -- and the difference, which is nothing but the round-off of an fp32 transform of that length, has to stay inside the
bound with the constant the GPU tests use.  The device's transforms (in-LDS radix-4, hipFFT) and FFTW's single-precision
library in the reference are fp32 transforms of the same kind; the GPU tests compare the device with the fp64 oracle under
that same bound (+ 1e-6 of the norm factor for everything that is not transform round-off)."""
import numpy as np
import pytest

from kiwi_amd import synthetic
from oracle import ko
from tests.common import Scenario, slot_scales, fft_roundoff_bound, FFT_ROUNDOFF_C

METHODS = {"l2norm": 1, "l1norm": 2, "ampspec_l2norm": 3, "ampspec_l1norm": 4, "peak": 6}


def evaluate(sc, mid, filt, trial, bits):
    ko.set_fft_precision(bits)
    try:
        e = sc.oracle()
        sc.apply_setup(e, True)
        e.set_misfit_method(mid)
        if filt is not None:
            for ir in range(sc.nrec):
                e.set_filter(ir + 1, *filt)
        e.set_source_params(1, trial)
        m, n, g = e.get_misfits()
        scales = slot_scales(e, sc.comps, sc.gf["dt"]) if bits == 64 else None
        e.close()
        return np.array(m, np.float64), np.array(n, np.float64), scales
    finally:
        ko.set_fft_precision(64)


@pytest.mark.parametrize("L", [200, 700, 2300])
def test_textbook_fp32_transform_stays_inside_the_round_off_bound(L):
    rng = np.random.default_rng(20261004 + L)
    worst = {}
    for case in range(6):
        sc = Scenario(nx=10, nz=5, ng=10, L=L, nrec=4, variant=str(rng.choice(["probe", "static"])),
                      comps_list=["ned", "ar", "d", "ne"], taper_ramp=float(rng.uniform(2, 12)))
        e0 = sc.oracle()
        sc.make_references(e0)
        e0.close()
        trial = np.array(synthetic.TRUE_BILAT, np.float32)
        trial[5] += rng.uniform(-20, 20); trial[6] -= rng.uniform(0, 20); trial[3] += rng.uniform(-1500, 1500); trial[0] += rng.uniform(-2, 2)
        f0 = rng.uniform(0.01, 0.05)
        filt = ([f0, 2 * f0, 6 * f0, 9 * f0], [0., 1., 1., 0.])
        for method, mid in METHODS.items():
            spectral = mid in (3, 4)
            for use_filter in ((False, True) if spectral else (True,)):
                m64, n64, scales = evaluate(sc, mid, filt if use_filter else None, trial, 64)
                m32, n32, _ = evaluate(sc, mid, filt if use_filter else None, trial, 32)
                nt, wl, na, nb = scales
                b1 = fft_roundoff_bound(method, sc.gf["dt"], nt, wl, na, nb, c=1.0)
                bn = fft_roundoff_bound(method, sc.gf["dt"], nt, wl, na, 0.0 * na, c=1.0)
                r = max(float(np.max(np.abs(m32 - m64) / b1)), float(np.max(np.abs(n32 - n64) / bn)))
                worst[(method, use_filter)] = max(worst.get((method, use_filter), 0.0), r)
                assert r <= FFT_ROUNDOFF_C, (method, use_filter, case, r)
                assert np.any(m32 != m64)             # (the fp32 transform does change the figures: the bound is not vacuous)
    print("L = %d: worst |m32 - m64| / bound(c = 1): %s" % (L, {k: round(v, 3) for k, v in worst.items()}))
