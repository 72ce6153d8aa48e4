"""get_peak_amplitudes / get_arias_intensities in the oracle (comparator.f90:519-625,700-766,824-859,1012-1058;
receiver.f90:512-594) against a direct numpy evaluation of the same formulas on the oracle's own probe arrays."""
import ctypes as C

import numpy as np
import pytest

from oracle import ko
from tests.common import Scenario

F = np.float32
PI = F(3.14159265358979)


def component_ids(comps):
    ids = ["wsulc?ardne".index(ch) - 5 for ch in comps]
    ih1 = ih2 = iver = 0
    for k, ct in enumerate(ids):
        if abs(ct) == 1: ih1 = k + 1
        if abs(ct) == 2: ih2 = k + 1
        if abs(ct) == 3: iver = k + 1
    if ih1 == 0 or ih2 == 0:
        for k, ct in enumerate(ids):
            if abs(ct) == 4: ih1 = k + 1
            if abs(ct) == 5: ih2 = k + 1
    if ih1 == 0 or ih2 == 0:
        ih1 = ih2 = 0
    return iver, ih1, ih2


def expected(e, sc, ir, kind, tapered, factor=1.0):
    iver, ih1, ih2 = component_ids(sc.comps[ir - 1])
    if kind != 3:
        used = [k for k in (iver, ih1, ih2) if k]
    elif iver and ih1 and ih2:
        used = [iver, ih1, ih2]
    elif ih1 and ih2:
        used = [ih1, ih2]
    else:
        used = [iver] if iver else []
    if not used:
        return 0.0
    arrs = []
    for k in used:
        lo, d = e.synthetic(ir, k, 2 if tapered else 1)
        arrs.append((lo, d))
    lo = min(a[0] for a in arrs)
    hi = max(a[0] + len(a[1]) - 1 for a in arrs)
    x = []
    for l0, d in arrs:                      # probe arrays: zeros before the data, the last value after it
        full = np.zeros(hi - lo + 1, F)
        full[l0 - lo:l0 - lo + len(d)] = d
        full[l0 - lo + len(d):] = d[-1]
        x.append(full)
    dt = F(sc.gf["dt"])
    f2 = np.float64(F(factor) * F(factor))
    if kind == 1:
        v = sum(f2 * (a[:-1] - a[1:]).astype(np.float64) ** 2 for a in x)
        return float(F(np.sqrt(v.max()) / np.float64(dt)))
    v = sum(f2 * ((a[:-2] - F(2.0) * a[1:-1]) + a[2:]).astype(np.float64) ** 2 for a in x)
    if kind == 2:
        return float(F(np.sqrt(v.max()) / np.float64(dt * dt)))
    return float(F(np.float64(PI / (F(2.) * F(9.81)) * dt) * v.sum() / np.float64(dt * dt)))


@pytest.mark.parametrize("tapered", [True, False])
def test_shake_values_match_numpy(tapered):
    comps = ["ned", "d", "ar", "nd", "une", "rau"]
    sc = Scenario(nrec=6, comps_list=comps)
    e = sc.oracle()
    sc.make_references(e)
    if not tapered:
        sc.tapers = {}
    sc.apply_setup(e, True)
    e.set_synthetics_factor(0.7)
    p = sc.true_params.copy()
    p[5] += 7.0
    e.set_source_params(1, p)
    for kind, got in ((1, e.peak_amplitudes(1)), (2, e.peak_amplitudes(2)), (3, e.arias_intensities())):
        assert len(got) == 6
        for ir in range(1, 7):
            want = expected(e, sc, ir, kind, tapered, 0.7)
            assert abs(got[ir - 1] - want) <= 2e-6 * abs(want), (kind, ir, got[ir - 1], want)
    assert got[3] > 0                     # 'nd': incomplete horizontals, the vertical alone counts
    e.switch_receiver(2, False)
    assert len(e.arias_intensities()) == 5
    e.close()
