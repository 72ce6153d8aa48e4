"""Parity of the HIP path (through the C-ABI) with the CPU oracle on the same seeded inputs.

Tolerances (stated once, used everywhere):
  * synthetics: |hip - oracle| <= SYN_RTOL * max|oracle| per trace.  The accumulate kernel itself
    is bit-exact given identical geometry records (test_accumulate_bitexact_given_geometry); the
    only source of differences is device libm (fp64 sin/cos/acos/asin/atan2 in the geometry
    kernel) vs glibc on the host, which moves an fp32 weight by at most an ulp or two.
  * misfits: relative 1e-6 (BASELINE.json north_star), norm factors bit-exact (host computed).
"""
import numpy as np
import pytest

from kiwi_amd import synthetic, KiwiHipError
from tests.common import (Scenario, oracle_misfits, misfit_close, same_bits, arith, MISFIT_RTOL, SYN_RTOL, slot_scales,
                          spectral_close, fft_roundoff_bound, FFT_ROUNDOFF_C)

pytestmark = pytest.mark.gpu


def build(sc, method="l2norm"):
    e = sc.oracle()
    sc.make_references(e)
    sc.apply_setup(e, True)
    p = sc.product()
    sc.apply_setup(p, False)
    mid = {"l2norm": 1, "l1norm": 2, "scalar_product": 5, "peak": 6}[method]
    e.set_misfit_method(mid)
    p.set_misfit_method(method)
    return e, p


@pytest.mark.parametrize("bilinear", [False, True])
@pytest.mark.parametrize("variant", ["probe", "static"])
def test_bilateral_misfits_match_oracle(bilinear, variant):
    sc = Scenario(bilinear=bilinear, variant=variant)
    e, p = build(sc)
    trials = np.vstack([sc.true_params[None], synthetic.bilat_strike_sweep(12, step=1.5)])
    trials[5:, 6] -= 7.0          # also move dip, depth and rake
    trials[8:, 3] += 1500.0
    trials[10:, 7] += 20.0
    m, n, g = oracle_misfits(e, 1, trials)
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert pm.shape == m.shape
    assert np.array_equal(pn[0], n[0])                     # norm factors: host-side, bit exact
    assert np.all(pm[0] <= 1e-6 * pn[0])                   # the true source reproduces its references
    assert misfit_close(pm[1:], m[1:], pn[1:]), np.max(np.abs(pm[1:] - m[1:]) / np.abs(m[1:]))
    assert misfit_close(pg[1:], g[1:], glob=True)


@pytest.mark.parametrize("method", ["l1norm", "scalar_product", "peak"])
def test_other_time_domain_norms(method):
    sc = Scenario()
    e, p = build(sc, method)
    trials = synthetic.bilat_strike_sweep(5, step=2.0)
    m, n, g = oracle_misfits(e, 1, trials)
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert np.array_equal(pn[0], n[0])
    assert np.allclose(pm, m, rtol=2e-6, atol=0)
    assert np.allclose(pg, g, rtol=2e-6, atol=0)


def test_synthetics_sample_for_sample():
    sc = Scenario(variant="static")
    e, p = build(sc)
    trial = synthetic.bilat_strike_sweep(1, step=4.0)
    p.set_source_params("bilateral", trial)
    p.eval()
    e.set_source_params(1, trial[0])
    e.get_misfits()
    for ir in range(1, sc.nrec + 1):
        for k in range(1, 4):
            for which in (1, 2):
                lo_o, so = e.synthetic(ir, k, which)
                lo_p, sp = p.get_synthetics(0, ir, k, which)
                # the product returns the taper window; the oracle its data span (plain) or
                # taper span /\ data span (tapered): compare on the overlap
                a = max(lo_o, lo_p)
                b = min(lo_o + len(so), lo_p + len(sp))
                assert b - a > 200
                xo = so[a - lo_o:b - lo_o]
                xp = sp[a - lo_p:b - lo_p]
                assert np.max(np.abs(xo - xp)) <= SYN_RTOL * np.max(np.abs(so))


def test_geometry_records_match_oracle():
    """Device geometry kernel vs the oracle's per-centroid quantities, field by field.  Integer
    fields (GF rows, shift, flags) must be identical; float fields are bit-identical except where
    the device's fp64 libm differs from glibc's in the last place and that flips an fp32 rounding."""
    from kiwi_amd.engine import GEOREC
    sc = Scenario(variant="static", nrec=8)
    e, p = build(sc)
    trial = synthetic.bilat_strike_sweep(1, step=3.0)
    p.set_source_params("bilateral", trial)
    p.eval()
    e.set_source_params(1, trial[0])
    e.get_misfits()
    nf = nbad = 0
    for ir in range(1, sc.nrec + 1):
        g = p.get_geometry(0, ir)
        o = e.centroid_geometry(ir, len(g), GEOREC)
        for name in ("row", "ishift"):
            assert np.array_equal(g[name], o[name]), name
        assert np.array_equal(g["flags"] & 3, o["flags"])        # bit2 (same point as predecessor) is device-only
        for name in ("w", "wfrac", "f", "cl", "sl"):
            d = g[name].view(np.int32).astype(np.int64) - o[name].view(np.int32).astype(np.int64)
            nf += d.size
            nbad += np.count_nonzero(d)
            assert np.allclose(g[name], o[name], rtol=1e-5, atol=1e-12), name
    assert nbad <= 0.002 * nf, (nbad, nf)


@pytest.mark.exact_only
def test_accumulate_bitexact_given_geometry():
    """Where the device geometry records are bit-identical to the host's, the synthetics must be
    bit-identical to the oracle's: the accumulate kernel reproduces the reference's fp32 operation
    order exactly (shift, interpolation, blend, rotation, repeated end points)."""
    from kiwi_amd.engine import GEOREC
    sc = Scenario(variant="static", nrec=8)
    e, p = build(sc)
    trial = synthetic.bilat_strike_sweep(1, step=3.0)
    p.set_source_params("bilateral", trial)
    p.eval()
    e.set_source_params(1, trial[0])
    e.get_misfits()
    checked = 0
    for ir in range(1, sc.nrec + 1):
        g = p.get_geometry(0, ir)
        o = e.centroid_geometry(ir, len(g), GEOREC)
        g["flags"] &= 3
        g["pad"] = 0                   # device-only group hint
        if g.tobytes() != o.tobytes():
            continue
        checked += 1
        for k in (1, 2, 3):
            lo_o, so = e.synthetic(ir, k, 1)
            lo_p, sp = p.get_synthetics(0, ir, k, 1)
            a = max(lo_o, lo_p)
            b = min(lo_o + len(so), lo_p + len(sp))
            assert np.array_equal(so[a - lo_o:b - lo_o].view(np.uint32), sp[a - lo_p:b - lo_p].view(np.uint32)), (ir, k)
    assert checked >= sc.nrec // 2, checked


def test_moment_tensor_and_circular_sources():
    sc = Scenario()
    e, p = build(sc)
    mt = synthetic.mt_sdr_grid(step=60, depth=9000.0)[:20]
    m, n, g = oracle_misfits(e, 6, mt)
    p.set_source_params("moment_tensor", mt)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert misfit_close(pm, m, pn) and misfit_close(pg, g, glob=True)
    circ = np.tile(np.array([0., 0., 0., 10000., 5e19, 80., 70., 100., 3000., 3000., 1.5], np.float32), (4, 1))
    circ[:, 8] = [1500., 2500., 3500., 4500.]
    m, n, g = oracle_misfits(e, 2, circ)
    p.set_source_params("circular", circ)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert misfit_close(pm, m, pn) and misfit_close(pg, g, glob=True)


def test_risetime_fold_and_synthetics_factor():
    sc = Scenario()
    e, p = build(sc)
    trial = synthetic.bilat_strike_sweep(3, step=2.0)
    tabs, moms = [], []
    from oracle import ko
    for t in trial:
        c, mo, _, _ = ko.discretize(1, t, sc.effective_dt)
        tabs.append(c)
        moms.append(mo)
    rise = [0.0, 1.5, 3.0]
    ms = []
    for c, mo, ri in zip(tabs, moms, rise):
        e.set_centroids(c, mo, ri)
        e.set_synthetics_factor(1.25)
        ms.append(e.get_misfits())
    p.set_sources(tabs, moms, rise)
    p.set_synthetics_factor(1.25)
    p.eval()
    pm, pn, pg = p.get_misfits()
    for i in range(3):
        assert misfit_close(pm[i], ms[i][0], pn[i]), i
        assert abs(pg[i] - ms[i][2]) <= MISFIT_RTOL * ms[i][2]


def test_edge_cases_components_depth_disabled_outofrange():
    comps = ["ned", "d", "ar", "une", "arn", "e"]
    sc = Scenario(comps_list=comps, depths=[0., 0., 500., 0., 0., 250.])
    e, p = build(sc)
    e.switch_receiver(4, False)
    p.switch_receiver(4, False)
    trials = synthetic.bilat_strike_sweep(4, step=2.5)
    trials[3, 3] = 2000.0          # depth above the first GF depth: bilinear lower node out of range -> centroids skipped
    m, n, g = oracle_misfits(e, 1, trials)
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert pm.shape == m.shape == (4, 3 + 1 + 2 + 3 + 1)
    assert np.array_equal(pn[0], n[0])
    assert misfit_close(pm, m, pn)
    assert misfit_close(pg, g, glob=True)
    mis, nor, failings = p.make_misfits_for_sources()
    assert mis.shape == (4, 6, 3) and np.all(mis[:, 3] == 0) and failings == []


def test_chunked_eval_equals_single_launch():
    sc = Scenario()
    e, p = build(sc)
    trials = synthetic.bilat_strike_sweep(9, step=0.7)
    p.set_source_params("bilateral", trials)
    p.eval()
    a = p.get_misfits()
    p.eval(0, 4)
    p.eval(4, 5)
    b = p.get_misfits()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_errors_are_reported_not_fatal():
    from kiwi_amd import KiwiHipError
    sc = Scenario()
    e = sc.oracle()
    sc.make_references(e)
    p = sc.product()
    p.set_source_params("bilateral", synthetic.bilat_strike_sweep(1))
    p.eval()                      # synthesis works without references (that is how references get made) ...
    with pytest.raises(KiwiHipError, match="misfits need a reference seismogram and a misfit taper"):
        p.get_misfits()           # ... misfits do not
    with pytest.raises(KiwiHipError):
        p.switch_receiver(99, True)
    with pytest.raises(KiwiHipError):
        p.set_misfit_method("nonsense")
    sc.apply_setup(p, False)
    p.eval()                      # recovers after the setup is completed
    m, n, g = p.get_misfits()
    assert np.all(np.isfinite(m))


@pytest.mark.parametrize("bilinear", [False, True])
def test_long_windows_span_several_tiles(bilinear):
    """Windows longer than one workgroup tile (the grouped kernel tiles time in 256..1024-sample
    pieces with an LDS halo): 1400-sample traces, every tile boundary must be seamless."""
    sc = Scenario(L=1400, nrec=4, bilinear=bilinear, variant="static")
    e, p = build(sc)
    trials = synthetic.bilat_strike_sweep(5, step=2.0)
    trials[3:, 3] += 1200.0
    m, n, g = oracle_misfits(e, 1, trials)
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert np.array_equal(pn[0], n[0])
    assert misfit_close(pm, m, pn), np.max(np.abs(pm - m) / np.abs(m))
    assert misfit_close(pg, g, glob=True)
    e.set_source_params(1, trials[1])
    e.get_misfits()
    for ir in (1, 3):
        for k in (1, 2, 3):
            lo_o, so = e.synthetic(ir, k, 1)
            lo_p, sp = p.get_synthetics(1, ir, k, 1)
            a = max(lo_o, lo_p)
            b = min(lo_o + len(so), lo_p + len(sp))
            assert b - a > 1300
            assert np.max(np.abs(so[a - lo_o:b - lo_o] - sp[a - lo_p:b - lo_p])) <= SYN_RTOL * np.max(np.abs(so))


@pytest.mark.parametrize("ng", [8, 10])
@pytest.mark.parametrize("us", [(1, 1), (2, 2)])
def test_far_field_db_and_spatial_undersampling(ng, us):
    """ng = 8 (far-field only database, gfdb.f90:57: the f6 / near-field terms drop out) and
    set_spacial_undersampling (gfdb.f90:794-815: bilinear nodes xus / zus grid steps apart)."""
    sc = Scenario(ng=ng, nx=12, nz=6)
    e, p = build(sc)
    e.set_interpolation(True, us[0], us[1])
    p.set_spacial_undersampling(us[0], us[1])
    trials = synthetic.bilat_strike_sweep(4, step=2.0)
    m, n, g = oracle_misfits(e, 1, trials)
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert misfit_close(pm, m, pn), np.max(np.abs(pm - m) / np.abs(m))
    assert misfit_close(pg, g, glob=True)


SPEC_RTOL = 1e-5      # ampspec / filtered norms: FFT libraries differ (hipFFT fp32 vs the oracle's fp64 DFT rounded to fp32)


@pytest.mark.parametrize("method", ["ampspec_l2norm", "ampspec_l1norm"])
@pytest.mark.parametrize("with_filter", [False, True])
def test_spectral_norms(method, with_filter):
    """comparator.f90:861-886,1186-1231 on hipFFT: amplitude-spectrum norms, optional frequency filter."""
    sc = Scenario()
    e, p = build(sc)
    mid = {"ampspec_l2norm": 3, "ampspec_l1norm": 4}[method]
    e.set_misfit_method(mid)
    p.set_misfit_method(method)
    if with_filter:
        fx, fy = [0.01, 0.03, 0.25, 0.4], [0., 1., 1., 0.]
        for ir in range(1, sc.nrec + 1):
            e.set_filter(ir, fx, fy)
            p.set_misfit_filter(ir, fx, fy)
    trials = synthetic.bilat_strike_sweep(5, step=2.0)
    m, n, g = oracle_misfits(e, 1, trials)
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert np.allclose(pn[0], n[0], rtol=SPEC_RTOL, atol=0)
    assert np.allclose(pm, m, rtol=SPEC_RTOL, atol=1e-7 * np.abs(n[0]).max()), np.max(np.abs(pm - m) / np.abs(m))
    assert np.allclose(pg, g, rtol=SPEC_RTOL)


@pytest.mark.parametrize("method", ["ampspec_l2norm", "l2norm"])
def test_amplitude_spectra_output(method):
    """output_seismogram_spectra (receiver.f90:666-708, probe_get_amp_spectrum comparator.f90:333-354): amplitude spectra of
    synthetic and reference probes, plain and filtered, sized as the comparator sizes the pair (a fresh reference engine
    after one misfit evaluation), under a spectral and under a time-domain method."""
    sc = Scenario(nrec=4)
    e, p = build(sc)
    mid = {"ampspec_l2norm": 3, "l2norm": 1}[method]
    e.set_misfit_method(mid)
    p.set_misfit_method(method)
    fx, fy = [0.01, 0.03, 0.25, 0.4], [0., 1., 1., 0.]
    for ir in (1, 3):                                   # receivers 2 and 4 stay without filter
        e.set_filter(ir, fx, fy)
        p.set_misfit_filter(ir, fx, fy)
    trials = synthetic.bilat_strike_sweep(2, step=3.0)
    p.set_source_params("bilateral", trials)
    p.eval()
    before = p.get_misfits()[0].copy()
    e.set_source_params(1, trials[1])
    e.get_misfits()
    for ir in (1, 2, 3):
        for k in (1, 3):
            for synth in (True, False):
                for filt in (False, True):
                    odf, want = e.amp_spectrum(ir, k, synth, filt)
                    df, got = p.get_amp_spectrum(ir, k, "synthetics" if synth else "references", filt, isrc=1)
                    assert abs(df - odf) <= 1e-7 * odf and got.shape == want.shape
                    assert np.allclose(got, want, rtol=SPEC_RTOL, atol=2e-6 * want.max()), (ir, k, synth, filt)
                    if filt and ir in (1, 3):
                        assert got[0] == 0.0 and got[-1] == 0.0 and not np.array_equal(got, p.get_amp_spectrum(ir, k, "synthetics" if synth else "references", False, isrc=1)[1])
    p.eval()                                            # the engine's own state (method, filters) is as before
    assert np.array_equal(p.get_misfits()[0], before)


@pytest.mark.parametrize("method", ["l2norm", "l1norm"])
def test_time_domain_norms_with_frequency_filter(method):
    """comparator.f90:810-813,1233-1263: r2c -> cosine PLF filter -> c2r / ntrans -> zero outside the taper."""
    sc = Scenario()
    e, p = build(sc, method)
    fx, fy = [0.01, 0.03, 0.25, 0.4], [0., 1., 1., 0.]
    for ir in range(1, sc.nrec + 1):
        if ir != 2:                       # receiver 2 stays unfiltered: both paths in one batch
            e.set_filter(ir, fx, fy)
            p.set_misfit_filter(ir, fx, fy)
    trials = synthetic.bilat_strike_sweep(4, step=2.5)
    m, n, g = oracle_misfits(e, 1, trials)
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert np.array_equal(pn[0][3:6], n[0][3:6])              # unfiltered receiver: host-side, exact
    assert misfit_close(pm[:, 3:6], m[:, 3:6], pn[:, 3:6])
    assert np.allclose(pn[0], n[0], rtol=SPEC_RTOL)
    assert np.allclose(pm, m, rtol=SPEC_RTOL, atol=1e-7 * np.abs(n[0]).max()), np.max(np.abs(pm - m) / np.abs(m))
    assert np.allclose(pg, g, rtol=SPEC_RTOL)
    e.set_source_params(1, trials[1])
    e.get_misfits()
    lo_o, so = e.synthetic(1, 3, 3)
    lo_p, sp = p.get_synthetics(1, 1, 3, 3)
    a = max(lo_o, lo_p)
    b = min(lo_o + len(so), lo_p + len(sp))
    assert b - a > 200
    assert np.max(np.abs(so[a - lo_o:b - lo_o] - sp[a - lo_p:b - lo_p])) <= 2e-5 * np.max(np.abs(so))


@pytest.mark.parametrize("method,with_filter", [("ampspec_l2norm", True), ("ampspec_l1norm", False), ("l2norm", True), ("l1norm", True),
                                                ("scalar_product", True), ("peak", True)])
def test_amplitude_spectrum_norms_in_one_kernel(method, with_filter, monkeypatch):
    """ampspec_* norms: spec_fft_norm_kernel transforms every (slot, source) row in LDS and reduces it to the misfit.  Its two
    row sources -- the plain synthetics (fold, moment and taper applied on the way in; the default) and the tapered rows
    misfit_kernel writes when the processed synthetics are kept -- give the same bits; the library path
    (KIWI_HIP_FUSED_FFT=0: hipFFT r2c + spec_norm_kernel) agrees to transform round-off, and all agree with the oracle.
    Sources with a rise time (the fold runs inside the load), two transform lengths in the batch.
    The time-domain norms on frequency-filtered traces go through the same transform forward AND back
    (spec_fft_filter_norm_kernel; with kept synthetics the library pair r2c / c2r runs): same checks to round-off."""
    sc = Scenario()
    mid = {"ampspec_l2norm": 3, "ampspec_l1norm": 4, "l2norm": 1, "l1norm": 2, "scalar_product": 5, "peak": 6}[method]
    spectral = mid in (3, 4)
    # an l1 / peak value of a filtered trace carries the transforms' round-off linearly over the window: no fixed figure, the
    # bound of tests/common.py (fft_roundoff_bound: log2 N, window length, what the filter rejects)
    derived = not (spectral or mid in (1, 5))
    tol = SPEC_RTOL
    dt = sc.gf["dt"]
    bound_dd = None               # (device against device: twice the round-off bound of the slots, from the oracle's scales below)
    trials = np.array([[0.3 * i, 0., 0., 9500. + 300 * i] + synthetic.mt_from_sdr(40. * i, 50. + 5 * i, -60. + 30 * i) + [1.0 + 0.7 * i]
                       for i in range(6)], np.float32)
    trials[4, 10] = 170.0                                   # a long source time function: the next transform length
    res = {}
    for mode in ("fused", "library"):
        monkeypatch.setenv("KIWI_HIP_FUSED_FFT", "1" if mode == "fused" else "0")
        e, p = build(sc)
        e.set_misfit_method(mid)
        p.set_misfit_method(method)
        if with_filter:
            fx, fy = [0.01, 0.03, 0.25, 0.4], [0., 1., 1., 0.]
            for ir in range(1, sc.nrec + 1):
                if ir != 2:
                    e.set_filter(ir, fx, fy)
                    p.set_misfit_filter(ir, fx, fy)
        p.set_source_params("moment_tensor", trials)
        p.eval()
        direct = [x.copy() for x in p.get_misfits()]
        p.set_keep_synthetics(2)                            # processed synthetics kept: misfit_kernel writes the rows
        p.eval()
        kept = p.get_misfits()
        p.set_keep_synthetics(0)
        for a, b in zip(direct, kept):
            if spectral or mode == "library":
                assert a.tobytes() == b.tobytes(), mode
            elif not derived:                               # kept synthetics: the library transforms run instead
                assert np.allclose(a, b, rtol=0, atol=tol * np.abs(direct[1]).max()), mode
            elif a.ndim == 2:                               # (misfits and norm factors per slot)
                if bound_dd is None:
                    # the oracle engine `e` holds the TRUE source's probes: scales of the right order for every trial of this test
                    e.get_misfits()
                    sc0 = slot_scales(e, sc.comps, dt)
                    bound_dd = 2 * FFT_ROUNDOFF_C * fft_roundoff_bound(method, dt, 2 * sc0[0], sc0[1], sc0[2], 4.0 * sc0[2], c=1.0)
                assert np.all(np.abs(a - b) <= MISFIT_RTOL * np.abs(direct[1]).max() + bound_dd[None, :]), mode
        res[mode] = direct
        if mode == "fused":
            for i, t in enumerate(trials):                  # every source against a fresh oracle engine
                fe = sc.oracle(); sc.apply_setup(fe, True); fe.set_misfit_method(mid)
                if with_filter:
                    for ir in range(1, sc.nrec + 1):
                        if ir != 2:
                            fe.set_filter(ir, fx, fy)
                om, on, og = oracle_misfits(fe, 6, t[None, :])
                if derived:
                    ok, ratio = spectral_close(method, dt, direct[0][i], om[0], on[0], slot_scales(fe, sc.comps, dt), direct[1][i])
                    assert ok, (i, ratio)
                else:
                    assert np.allclose(direct[1][i], on[0], rtol=tol, atol=0), i
                    assert np.allclose(direct[0][i], om[0], rtol=tol, atol=tol * np.abs(on[0]).max()), i
                fe.close()
            # the source the references were made from (by the oracle: equal to the device's synthetics to an ulp or two)
            p.set_source_params(sc.true_type, sc.true_params[None, :])
            p.eval()
            tm, tn, _ = p.get_misfits()
            if mid not in (5, 6):                           # (scalar product and peak are not difference measures)
                assert np.all(tm <= (1e-4 if derived else 1e-5) * tn)
        p.close(); e.close()
    fm, fn, fg = res["fused"]
    lm, ln, lg = res["library"]
    if spectral:
        assert len({x.tobytes() for x in fn}) > 1            # two transform lengths
    assert np.allclose(fn, ln, rtol=max(2e-6, tol / 10), atol=0) and np.allclose(fm, lm, rtol=0, atol=max(4e-6, tol / 5) * np.abs(fn).max())


@pytest.mark.parametrize("method,with_filter", [("ampspec_l2norm", False), ("ampspec_l1norm", True), ("l2norm", True)])
def test_spectral_results_do_not_depend_on_the_batch(method, with_filter, monkeypatch):
    """The transform length of a probe pair follows the trial source's OWN strips (what a fresh reference engine gives it,
    comparator.f90:222-271,464-486), not the batch it is evaluated in: a source evaluated alone, inside a batch, in a
    different batch order or through small internal chunks gives the same misfits, norm factors and global misfit bit
    for bit -- what a multi-GPU shard of the trial list needs to reproduce the one-GPU numbers."""
    sc = Scenario()
    e, p = build(sc)
    mid = {"ampspec_l2norm": 3, "ampspec_l1norm": 4, "l2norm": 1}[method]
    e.set_misfit_method(mid)
    p.set_misfit_method(method)
    if with_filter:
        fx, fy = [0.01, 0.03, 0.25, 0.4], [0., 1., 1., 0.]
        for ir in range(1, sc.nrec + 1):
            if ir != 2:
                e.set_filter(ir, fx, fy)
                p.set_misfit_filter(ir, fx, fy)
    # moment-tensor point sources; two of them with a source time function of 150 s / 170 s: their strips are more than
    # twice as long as the others', so their probe pairs need the next transform length
    trials = np.array([[0.3 * i, 0., 0., 9500. + 300 * i] + synthetic.mt_from_sdr(40. * i, 50. + 5 * i, -60. + 30 * i) + [1.0]
                       for i in range(6)], np.float32)
    trials[1, 10] = 150.0
    trials[4, 10] = 170.0
    stype, sid = "moment_tensor", 6
    p.set_source_params(stype, trials)
    p.eval()
    bm, bn, bg = [x.copy() for x in p.get_misfits()]
    assert len({x.tobytes() for x in bn}) > 1            # norm factors follow the transform length: two lengths in this batch
    for i in range(len(trials)):                          # alone
        p.set_source_params(stype, trials[i:i + 1])
        p.eval()
        am, an, ag = p.get_misfits()
        assert am[0].tobytes() == bm[i].tobytes() and an[0].tobytes() == bn[i].tobytes() and ag[0] == bg[i], i
    perm = [4, 0, 5, 1, 3, 2]                             # another batch order
    p.set_source_params(stype, trials[perm])
    p.eval()
    qm, qn, qg = p.get_misfits()
    assert np.array_equal(qm, bm[perm]) and np.array_equal(qn, bn[perm]) and np.array_equal(qg, bg[perm])
    p.eval(0, 2); p.eval(2, 4)                            # evaluated in pieces
    rm, rn, rg = p.get_misfits()
    assert np.array_equal(rm, qm) and np.array_equal(rn, qn) and np.array_equal(rg, qg)
    # and every source agrees with a FRESH oracle engine (the reference's spans remember earlier sources)
    for i in (0, 1, 4):
        ef = sc.oracle()
        sc.apply_setup(ef, True)
        ef.set_misfit_method(mid)
        if with_filter:
            for ir in range(1, sc.nrec + 1):
                if ir != 2:
                    ef.set_filter(ir, fx, fy)
        ef.set_source_params(sid, trials[i])
        om, on, og = ef.get_misfits()
        assert np.allclose(bn[i], on, rtol=SPEC_RTOL, atol=0), i
        assert np.allclose(bm[i], om, rtol=SPEC_RTOL, atol=2e-5 * np.abs(on).max()), (i, np.max(np.abs(bm[i] - om) / on))
        assert abs(bg[i] - og) <= 2e-5 * og
        ef.close()


@pytest.mark.parametrize("L,edt,stagger", [(700, 0.5, False), (1500, 1.0, False), (1500, 0.5, True)])
def test_cell_kernels_are_bit_identical(monkeypatch, L, edt, stagger):
    """Sources whose centroids are all different points (an eikonal rupture) through the four accumulate paths: the cell
    kernel with a tile per wave (accumulate_cellw_kernel, the default), the one with a shared tile (KIWI_HIP_CELL_WAVE=0), the
    grouped kernel with groups of one centroid (KIWI_HIP_CELL=0) and the direct kernel -- the same synthetics bit for bit.
    Several tiles per window, receivers the cell kernels do not take (one component family only), a slow rupture (shift
    ranges beyond what a per-wave halo holds: cellgroup_kernel cuts the runs) and a fast one.  `stagger`: the components of a
    node start at different samples (the cell kernels then take forty row descriptors per run instead of four)."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eikonal_vectors.npz"))
    comps = ["ned", "ned", "d", "ne", "aru", "ned"]
    sc = Scenario(nz=6, L=L, nrec=6, comps_list=comps, effective_dt=edt)
    if stagger:
        sc.gf["first"] = sc.gf["first"] + np.arange(sc.gf["first"].shape[2], dtype=sc.gf["first"].dtype)[None, None, :] * 3
    sc.oracle()
    if stagger:
        first = sc.odb.dense_tables()[0]
        assert np.any(np.ptp(first, axis=2) > 0)
    trials = []
    for i, vfac in enumerate([0.8, 0.25, 0.8]):          # relative rupture velocity: 0.25 stretches the arrival times of a cell's centroids
        common = [0.2 * i, 300.0 * i, -200.0 * i, 10500.0 + 300 * i]
        trials.append(common + [7e18, 80.0 + 5 * i, 70.0, -170.0 + 10 * i] + [100.0, -50.0, 2500.0 + 400 * i] + [500.0 - 300 * i, 200.0]
                      + [vfac, 0.0])
    trials = np.array(trials, np.float32)
    cp = np.array([[0, 0, 6500.0], [0, 0, 15500.0], [0, -2000.0, 0]], np.float32)
    cn = np.array([[0, 0, -1.0], [0, 0, 1.0], [0.2, -1.0, 0]], np.float32)
    res, ranges = {}, None
    for mode in ("cellw", "cell", "grouped", "direct"):
        for k in ("KIWI_HIP_ACCUM", "KIWI_HIP_CELL", "KIWI_HIP_CELL_WAVE"):
            monkeypatch.delenv(k, raising=False)
        if mode == "direct":
            monkeypatch.setenv("KIWI_HIP_ACCUM", "direct")
        elif mode == "grouped":
            monkeypatch.setenv("KIWI_HIP_CELL", "0")
        else:
            monkeypatch.setenv("KIWI_HIP_CELL", "1")
            monkeypatch.setenv("KIWI_HIP_CELL_WAVE", "1" if mode == "cellw" else "0")
        p = sc.product()
        p.set_source_crust(G["rupture_profile"], G["origin_profile"])
        p.set_source_constraints(cp, cn)
        p.set_keep_synthetics(1)
        p.set_source_params("eikonal", trials)
        p.eval()
        res[mode] = [p.get_synthetics(s, ir, k, 1)[1] for s in range(len(trials)) for ir in range(1, 7)
                     for k in range(1, len(comps[ir - 1]) + 1)]
        if mode in ("cellw", "cell"):
            g = [p.get_geometry(s, 1) for s in range(len(trials))]
            # group hints of the cell pass: length and shift range at the records that start a run
            rng_ = [int(((r["pad"] >> 8) & 0xff).max() + ((r["pad"] >> 16) & 0xff).max()) for r in g]
            lens = [int((r["pad"] & 0xff).max()) for r in g]
            if mode == "cellw":
                ranges = rng_
                assert max(lens) > 3                       # runs of several centroids per cell
                assert max(rng_) <= 2 * 23                 # (each side of the head's shift within the per-wave limit)
            else:
                assert max(rng_) >= max(ranges)            # the shared tile allows the longer runs
        p.close()
    assert any(np.any(a != 0) for a in res["direct"]) and len(res["direct"]) == 3 * 15
    for mode in ("cellw", "cell", "grouped"):
        for i, (a, b) in enumerate(zip(res[mode], res["direct"])):
            assert same_bits(a, b), (mode, i)


@pytest.mark.parametrize("stype", ["eikonal", "mt_eikonal"])
def test_eikonal_sources_with_risetime_fold(stype):
    """Variable-rupture-speed sources: discretised by the product's host code from the crust
    profile + constraints, rise time applied by the fold in the misfit kernel."""
    import os
    from oracle import ko
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eikonal_vectors.npz"))
    prof = G["rupture_profile"]
    sc = Scenario(nz=6)
    e, p = build(sc)
    st = 4 if stype == "eikonal" else 5
    rng = np.random.default_rng(5 + st)
    trials = []
    for i in range(4):
        common = [0.2 * i, 300.0 * i, -200.0 * i, 10500.0 + 300 * i]
        bord = [100.0, -50.0, 2500.0 + 400 * i]
        nukl = [500.0 - 300 * i, 200.0]
        if st == 5:
            trials.append(common + [1.0, 80.0 + 5 * i, 70.0] + bord + nukl + [0.8] + list(rng.standard_normal(6) * 7e17) + [0.8 * i])
        else:
            trials.append(common + [7e18, 80.0 + 5 * i, 70.0, -170.0 + 10 * i] + bord + nukl + [0.8, 0.8 * i])
    trials = np.array(trials, np.float32)
    cp = np.array([[0, 0, 6500.0], [0, 0, 15500.0], [0, -2000.0, 0]], np.float32)
    cn = np.array([[0, 0, -1.0], [0, 0, 1.0], [0.2, -1.0, 0]], np.float32)
    p.set_source_crust(prof, G["origin_profile"])
    thick = ko.crust_thickness(ko.crust_profile(*np.split(G["origin_profile"], [8, 16, 24])))
    assert p.get_source_crustal_thickness() == thick
    p.set_source_crustal_thickness_limit(9000.0)
    assert p.get_source_crustal_thickness() == 9000.0
    p.set_source_constraints(cp, cn)
    p.set_source_params(stype, trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    oprof = ko.crust_profile(*np.split(prof, [8, 16, 24]))
    ncent = []
    for i, t in enumerate(trials):
        c, mo, ri, _ = ko.discretize_eikonal(st, t, sc.effective_dt, oprof, cp, cn)
        ncent.append(len(c))
        e.set_centroids(c, mo, ri)
        m, n, g = e.get_misfits()
        assert misfit_close(pm[i], m, pn[i]), i
        assert np.array_equal(pn[i], n)
        assert abs(pg[i] - g) <= MISFIT_RTOL * g
    assert min(ncent) > 30 and len(set(ncent)) > 1
    # an impossible source is reported with the reference's message, not fatal
    bad = trials[:1].copy()
    bad[0, 3] = 500.0
    with pytest.raises(KiwiHipError, match="Empty rupture area"):
        p.set_source_params(stype, bad)
    p.set_source_params(stype, trials[:1])
    p.eval()
    assert misfit_close(p.get_misfits()[0][0], pm[0], pn[0])
    # ... and inside a batch it is skipped, not fatal (seismosizer.py:703-720): a grid with two "Empty rupture area"
    # points and one nucleation point outside of the rupture region
    grid = np.concatenate([trials[:2], bad, trials[2:3], bad, trials[3:]], 0)
    grid[4, 3] = 300.0
    out = trials[1:2].copy()
    out[0, (10 if st == 5 else 11)] = 9000.0            # nukl-shift-x far outside the 2.9 km bounding circle
    grid = np.concatenate([grid, out], 0)
    p.set_source_params(stype, grid)
    status = p.get_source_status()
    assert list(status) == [0, 0, 5, 0, 5, 0, 6]
    assert p.source_status_message(5) == "Empty rupture area"
    assert p.source_status_message(6) == "position of nucleation point is outside of rupture region"
    mis, nor, failings = p.make_misfits_for_sources()
    assert failings == [2, 4, 6]
    gm, gn, gg = p.get_misfits()
    good = [0, 1, 3, 5]
    assert np.array_equal(gm[good], pm) and np.array_equal(gn[good], pn) and np.array_equal(gg[good], pg)
    assert np.all(gm[failings] == 0) and np.all(gn[failings] == 0) and np.all(gg[failings] == 0)
    assert np.all(mis[failings] == 0) and np.all(nor[failings] == 0) and np.all(mis[good][:, 0] != 0)
    from kiwi_amd.engine import make_global_misfits
    gl, _ = make_global_misfits(mis, nor)
    assert np.all(np.isnan(gl[failings])) and np.all(np.isfinite(gl[good]))
    # the same through the sweep API when every trial fails: all failings, no exception
    mis, nor, failings = p.make_misfits_for_sources(stype, np.concatenate([bad, bad], 0))
    assert failings == [0, 1] and np.all(mis == 0) and np.all(nor == 0)
    # one call for the whole trial list, discretiser and device overlapped (kiwi_hip_misfits_for_params): whatever the piece
    # size -- pieces that hold failings, a piece of nothing but failings (sources 7 and 8 at piece = 2 ... ) -- the results are
    # those of the calls above, bit for bit
    big = np.concatenate([grid, bad, bad, trials[:1]], 0)
    want_status = [0, 0, 5, 0, 5, 0, 6, 5, 5, 0]
    # (lists of 18 below: the list's last piece is worked on as a ramp of an eighth, an eighth, a quarter and half of it, kiwi_hip.hip)
    for piece in (1, 2, 3, 4, 5, 6, 7, 10, 0):
        fm, fn, fg, fs = p.misfits_for_params(stype, big, piece)
        assert list(fs) == want_status, piece
        ok = [0, 1, 3, 5, 9]
        assert np.array_equal(fm[ok], pm[[0, 1, 2, 3, 0]]) and np.array_equal(fn[ok], pn[[0, 1, 2, 3, 0]]), piece
        assert np.array_equal(fg[ok], pg[[0, 1, 2, 3, 0]]), piece
        nok = [i for i in range(10) if i not in ok]
        assert np.all(fm[nok] == 0) and np.all(fn[nok] == 0) and np.all(fg[nok] == 0), piece
    # a head piece of nothing but failings: the engine is left with the first piece that uploaded anything -- with pieces of 2 the
    # ramp's half piece [4, 5) ... whatever the cut, source 0 of the engine is the first good trial of that piece
    from kiwi_amd.engine import _pieces
    headbad = np.concatenate([bad, bad, trials, bad, trials[:2]] + [bad] * 8 + [trials[3:]], 0)
    for piece in (2, 8, 9, 16):
        hm, hn, hg, hs = p.misfits_for_params(stype, headbad, piece)
        assert list(hs) == [5, 5, 0, 0, 0, 0, 5, 0, 0] + [5] * 8 + [0], piece
        good2 = [2, 3, 4, 5, 7, 8, 17]
        assert np.array_equal(hm[good2], pm[[0, 1, 2, 3, 0, 1, 3]]) and np.array_equal(hg[good2], pg[[0, 1, 2, 3, 0, 1, 3]]), piece
        assert len(_pieces(len(headbad), piece, st)) == {2: 9, 8: 3, 9: 5, 16: 2}[piece]
        first, cnt = next((f, c) for f, c in _pieces(len(headbad), piece, st) if np.any(hs[f:f + c] == 0))
        assert p.nsrc == cnt, (piece, p.nsrc, cnt)
        em, en, eg = p.get_misfits()
        assert np.array_equal(em, hm[first:first + cnt]) and np.array_equal(eg, hg[first:first + cnt]), piece
    mis2, nor2, failings2 = p.make_misfits_for_sources(stype, big, piece=3)
    mis3, nor3, failings3 = p.make_misfits_for_sources(stype, big)
    assert failings2 == [2, 4, 6, 7, 8] == failings3
    assert np.array_equal(mis2, mis3) and np.array_equal(nor2, nor3)
    assert np.all(mis2[failings2] == 0) and np.all(mis2[[0, 1, 3, 5, 9]][:, 0] != 0)


@pytest.mark.parametrize("method,filt", [("l2norm", False), ("ampspec_l1norm", True), ("l1norm", True)])
def test_one_call_for_a_trial_list_equals_the_piecewise_calls(method, filt):
    """kiwi_hip_misfits_for_params (make_misfits_for_sources in one call, seismosizer.py:682-722; the host discretiser of
    the next piece runs while the device evaluates the present one) returns what set_source_params + get_misfits return
    for the whole list, bit for bit and for every piece size -- time-domain, spectral and filtered comparators (the
    spectral norm factors are per source: they must not depend on the piece a source falls in either)."""
    sc = Scenario(nz=6)
    e, p = build(sc)
    p.set_misfit_method(method)
    if filt:
        for ir in range(1, sc.nrec + 1):
            p.set_misfit_filter(ir, [0.01, 0.03, 0.2, 0.4], [0., 1., 1., 0.])
    trials = synthetic.bilat_strike_sweep(23, step=3.0)
    trials[:, 3] += 150.0 * np.arange(23)
    trials[::5, 13] = 6.0                                   # some long rise times: other data spans, other transform lengths
    p.set_source_params("bilateral", trials)
    p.eval()
    m0, n0, g0 = p.get_misfits()
    assert np.all(m0 != 0)
    for piece in (1, 4, 8, 23, 64, 0):
        m, n, g, st = p.misfits_for_params("bilateral", trials, piece)
        assert not st.any()
        assert np.array_equal(m, m0) and np.array_equal(n, n0) and np.array_equal(g, g0), piece
        assert p.nsrc == min(piece or 1024, 23)
        gm = p.get_misfits()[0]                             # the context is left with the head of the list
        assert np.array_equal(gm, m0[:p.nsrc])
    with pytest.raises(KiwiHipError, match="wrong number"):
        p.misfits_for_params("moment_tensor", trials, 4)


@pytest.mark.parametrize("stype", ["bilateral", "mt_eikonal", "moment_tensor"])
def test_moment_and_risetime_sweeps_rescale_instead_of_resynthesising(monkeypatch, stype):
    """minimizer_engine.f90:516-521 (source_bilat.f90:206, source_mt_eikonal.f90:234-239): when only the moment (or, for the
    eikonal types, the rise time) changes the reference re-scales the seismograms without synthesising them again.  Here
    trial sources of one batch whose centroid tables are identical are synthesised once; their misfits are bit-identical to
    evaluating every source on its own (KIWI_HIP_DEDUPE=0) and match the oracle, and the accumulate kernel ran for the
    distinct tables only (its time shows it)."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eikonal_vectors.npz"))
    sc = Scenario(nz=6)
    e, p = build(sc)
    if stype == "bilateral":
        base = synthetic.bilat_strike_sweep(3, step=4.0)
        trials = np.repeat(base, 4, axis=0)
        trials[:, 4] *= np.tile([1.0, 0.5, 2.0, 1.7], 3)               # moment
        sid = 1
    elif stype == "moment_tensor":
        # point sources at one location (runs of geometry-identical sources share blended tiles) with exact repeats, as the
        # degenerate strike / dip / rake combinations of a double-couple grid produce them
        mts = [synthetic.mt_from_sdr(30. * i, 60., -90. + 20 * i) for i in range(5)]
        trials = np.array([[0., 0., 0., 10000.] + mts[k] + [1.0] for k in (0, 1, 0, 2, 1, 3, 3, 4, 0, 2, 4, 1)], np.float32)
        sid = 6
    else:
        t0 = np.array([0.2, 300., -200., 10500., 1.0, 85., 70., 100., -50., 2900., 200., 200., 0.8] +
                      synthetic.mt_from_sdr(40., 60., -70., 7e17) + [0.0], np.float32)
        t1 = t0.copy(); t1[5] += 8.0
        trials = np.repeat(np.stack([t0, t1]), 4, axis=0)
        trials[:, 19] = np.tile([0.0, 1.5, 0.8, 1.5], 2)                # rise time
        trials[:, 4] = np.tile([1.0, 1.0, 0.6, 2.0], 2)                 # moment factor
        sid = 5
        cp = np.array([[0, 0, 6500.0], [0, 0, 15500.0]], np.float32)
        cn = np.array([[0, 0, -1.0], [0, 0, 1.0]], np.float32)
        for q in (p,):
            q.set_source_crust(G["rupture_profile"], G["origin_profile"])
            q.set_source_constraints(cp, cn)
    trials = trials[np.array([0, 1, 4, 2, 5, 3] + list(range(6, len(trials))))]      # identical tables need not be neighbours
    monkeypatch.setenv("KIWI_HIP_DEDUPE", "2")             # also for point sources (by default only sources of >= 8 centroids)
    p.close()
    p = sc.product()
    sc.apply_setup(p, False)
    if sid == 5:
        p.set_source_crust(G["rupture_profile"], G["origin_profile"])
        p.set_source_constraints(cp, cn)
    p.set_source_params(stype, trials)
    p.eval()
    a = [x.copy() for x in p.get_misfits()]
    ms_shared, _ = p.kernel_ms()
    monkeypatch.setenv("KIWI_HIP_DEDUPE", "0")
    q = sc.product()
    sc.apply_setup(q, False)
    if sid == 5:
        q.set_source_crust(G["rupture_profile"], G["origin_profile"])
        q.set_source_constraints(cp, cn)
    q.set_source_params(stype, trials)
    q.eval()
    b = q.get_misfits()
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()
    if sid != 6:
        assert len({x.tobytes() for x in a[0]}) == len(trials)         # every trial has its own misfits
    else:
        assert len({x.tobytes() for x in a[0]}) == 5
    if sid in (1, 6):
        m, n, g = oracle_misfits(e, sid, trials)
        assert misfit_close(a[0], m, a[1]) and misfit_close(a[2], g, glob=True)
    p.eval(0, 2)                                                        # a chunk that holds the first of a family only
    p.eval(2, len(trials) - 2)                                          # ... the rest refers back across the chunk border: synthesised
    c = p.get_misfits()
    for x, y in zip(a, c):
        assert x.tobytes() == y.tobytes()


def _knock_out(sc, holes):
    """Mark single traces (ix, iz, ig; 0-based) of the scenario's database as not stored (nsamp = 0: the chunk index holds no
    reference for them, gfdb.f90:1003)."""
    for ix, iz, ig in holes:
        sc.gf["nsamp"][ix, iz, ig] = 0


@pytest.mark.parametrize("bilinear", [True, False])
@pytest.mark.parametrize("accum", ["grouped", "direct"])
def test_cycle_at_the_first_missing_trace(monkeypatch, bilinear, accum):
    """seismogram.f90:171-250: `if (.not. associated(tracep)) cycle` leaves a centroid at the first trace that is not
    stored -- what was added before stays (non-rotating branch, vertical block), horizontals collected for the rotation
    are dropped, and a gap among the horizontals also skips the vertical block.  Database with single (ix, iz, ig)
    traces unset at nodes the sources use; point source at the origin (no rotation) and extended ones (rotation)."""
    if accum == "direct":
        monkeypatch.setenv("KIWI_HIP_ACCUM", "direct")
    comps = ["ned", "d", "ne", "ar", "ned", "u", "ned", "ned"]
    sc = Scenario(nrec=8, comps_list=comps, nx=10, nz=5)
    holes = []
    for iz in range(5):
        holes += [(1, iz, 2), (2, iz, 8), (3, iz, 4), (4, iz, 6), (5, iz, 9), (6, iz, 0), (7, iz, 5)]
    holes += [(0, 1, 3), (0, 2, 7)]
    e = sc.oracle()
    sc.make_references(e)             # references from the complete database
    sc.apply_setup(e, True)
    _knock_out(sc, holes)
    e = sc.oracle()
    sc.apply_setup(e, True)
    p = sc.product()
    sc.apply_setup(p, False)
    nmissing = int((sc.gf["nsamp"] == 0).sum())
    assert nmissing == len(holes)
    # (a) moment tensor point sources at the origin: lambda == 0, partial sums stay
    mt = np.array([[0., 0., 0., 10000.] + synthetic.mt_from_sdr(30. * i, 60., -90. + 20 * i) + [1.0] for i in range(4)], np.float32)
    m, n, g = oracle_misfits(e, 6, mt)
    p.set_source_params("moment_tensor", mt)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert np.array_equal(pn[0], n[0])
    assert misfit_close(pm, m, pn), np.max(np.abs(pm - m) / np.abs(m))
    # the partial rule is really exercised: some centroid has bit 3 set, and dropping it changes the answer
    flags = np.concatenate([p.get_geometry(0, ir)["flags"] for ir in range(1, 9)])
    rows = np.concatenate([p.get_geometry(0, ir)["row"][:, 0] for ir in range(1, 9)])
    assert np.any(flags & 8) and np.any(rows < 0) and np.any((rows >= 0) & ((flags & 8) == 0))
    # (b) extended sources: rotation branch for every sub-fault off the origin
    trials = synthetic.bilat_strike_sweep(5, step=7.0)
    trials[:, 3] = [9000., 10000., 11000., 8000., 12000.]
    m, n, g = oracle_misfits(e, 1, trials)
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert misfit_close(pm, m, pn), np.max(np.abs(pm - m) / np.abs(m))
    assert misfit_close(pg, g, glob=True)
    flags = np.concatenate([p.get_geometry(0, ir)["flags"] for ir in range(1, 9)])
    assert np.any((flags & 8) != 0) and np.any((flags & 10) == 2)
    # sample for sample on one source
    e.set_source_params(1, trials[2])
    e.get_misfits()
    nempty = 0
    for ir in range(1, 9):
        for k in range(1, len(comps[ir - 1]) + 1):
            lo_o, so = e.synthetic(ir, k, 1)
            lo_p, sp = p.get_synthetics(2, ir, k, 1)
            if len(so) <= 1:                       # every centroid left before anything was added to this strip
                assert np.all(sp == 0)
                nempty += 1
                continue
            a, b = max(lo_o, lo_p), min(lo_o + len(so), lo_p + len(sp))
            assert b - a > 150
            assert np.max(np.abs(so[a - lo_o:b - lo_o] - sp[a - lo_p:b - lo_p])) <= SYN_RTOL * max(np.max(np.abs(so)), 1e-30)
    assert 0 < nempty < 10


@pytest.mark.parametrize("method", ["ampspec_l2norm", "l2norm_untapered"])
def test_radial_and_transverse_strips_keep_their_own_spans(method):
    """The two horizontal sums are separate strips (seismogram.f90:205-231): a point source at the origin (no rotation) whose
    transverse traces are missing extends the radial strip only, so the right / left component's probe stays empty while the
    away and the north / east components (made equal before the final rotation, :268-283) carry data.  Matters wherever the
    comparator follows the strips' data spans: transform lengths of the spectral norms, un-tapered norms (found by the
    randomised sweep with missing traces)."""
    comps = ["ar", "nl", "ne", "lc", "rd", "n"]
    sc = Scenario(nrec=6, comps_list=comps, L=96, nx=10, nz=5)
    e = sc.oracle()
    sc.make_references(e)
    if method == "l2norm_untapered":
        sc.tapers = {}
    _knock_out(sc, [(ix, iz, 3) for ix in range(10) for iz in range(5)])      # every first transverse trace: 4 and 5 never added
    mt = np.array([[0., 0., 0., 10000.] + synthetic.mt_from_sdr(30. * i, 60., -90. + 20 * i) + [1.0] for i in range(3)], np.float32)
    sc.oracle()                              # (the product takes its tables from the scenario's packed database: rebuild it)
    p = sc.product()
    sc.apply_setup(p, False)
    mid = 3 if method == "ampspec_l2norm" else 1
    p.set_misfit_method(method.split("_untapered")[0])
    p.set_source_params("moment_tensor", mt)
    p.eval()
    pm, pn, pg = p.get_misfits()
    for i in range(3):                       # fresh engine per source: the reference's spans remember earlier sources
        ef = sc.oracle()
        sc.apply_setup(ef, True)
        ef.set_misfit_method(mid)
        ef.set_source_params(6, mt[i])
        om, on, og = ef.get_misfits()
        tol = SPEC_RTOL if mid == 3 else MISFIT_RTOL
        assert np.allclose(pn[i], on, rtol=tol, atol=0), (i, pn[i], on)
        assert np.allclose(pm[i], om, rtol=tol, atol=tol * np.abs(on).max()), (i, np.max(np.abs(pm[i] - om) / on))
        ef.close()
    # the transverse probes see no synthetic at all: their misfit is the norm of the reference
    assert np.allclose(pm[:, 1], pn[:, 1], rtol=SPEC_RTOL) and not np.allclose(pm[:, 0], pn[:, 0], rtol=1e-3)


def test_grouped_and_direct_accumulate_are_bit_identical(monkeypatch):
    """The LDS-staged kernel only moves where the blended traces are read from: its synthetics equal the direct kernel's
    bit for bit (also with traces missing from the database and the static variant's repeated end values)."""
    sc = Scenario(nrec=6, comps_list=["ned", "d", "ne", "ar", "ned", "u"], variant="static")
    _knock_out(sc, [(2, iz, 8) for iz in range(5)] + [(4, iz, 6) for iz in range(5)] + [(3, 2, 1)])
    res = {}
    for mode in ("grouped", "direct"):
        if mode == "direct":
            monkeypatch.setenv("KIWI_HIP_ACCUM", "direct")
        e = sc.oracle()
        p = sc.product()
        trials = np.vstack([synthetic.bilat_strike_sweep(3, step=5.0)])
        p.set_source_params("bilateral", trials)
        p.set_keep_synthetics(1)
        p.eval()
        res[mode] = [p.get_synthetics(s, ir, k, 1)[1] for s in range(3) for ir in range(1, 7) for k in range(1, len(sc.comps[ir - 1]) + 1)]
        p.close()
    assert len(res["grouped"]) == len(res["direct"]) > 30
    for a, b in zip(res["grouped"], res["direct"]):
        assert same_bits(a, b)
    assert any(np.any(a != 0) for a in res["grouped"])


@pytest.mark.parametrize("method", ["floating_l2norm", "floating_l1norm"])
def test_floating_norms(method):
    """A15 (receiver.f90:439-510): argmin over integer shifts of the reference, per receiver."""
    sc = Scenario(nrec=5, comps_list=["ned", "ne", "d", "ned", "n"])
    e, p = build(sc)
    mid = 7 if method == "floating_l2norm" else 8
    e.set_misfit_method(mid)
    p.set_misfit_method(method)
    dt = sc.gf["dt"]
    ranges = [(-4, 3), (-2, 2), (0, 5), (-6, 0), (1, 1)]
    for ir, (lo, hi) in enumerate(ranges):
        e.set_floating_shiftrange(ir + 1, lo, hi)
        p.set_floating_shiftrange(ir + 1, lo * dt + 0.1 * dt, hi * dt - 0.1 * dt)       # nint() of seconds / dt
    trials = synthetic.bilat_strike_sweep(6, step=1.5)
    trials[:, 0] = [-1.6, -0.7, 0.0, 0.4, 1.1, 2.3]              # origin times: the best shift follows them
    m, n, g = oracle_misfits(e, 1, trials)
    shifts = []
    for t in trials:
        e.set_source_params(1, t)
        e.get_misfits()
        shifts.append([e.floating_shift(ir + 1) * dt for ir in range(5)])
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert np.array_equal(pn[0], n[0])
    assert misfit_close(pm, m, n)
    assert misfit_close(pg, g, glob=True)
    ps = p.get_floating_shifts()
    assert np.array_equal(ps, np.array(shifts, np.float32))
    assert len(set(ps[:, 0])) > 2                                 # the winning shift does move with the origin time
    # a single shift of zero is the plain norm
    p.set_floating_shiftrange(0, 0.0, 0.0)
    p.eval()
    fm, fn, fg = p.get_misfits()
    p.set_misfit_method(method[len("floating_"):])
    p.eval()
    qm, qn, qg = p.get_misfits()
    assert np.array_equal(fm, qm) and np.array_equal(fn, qn) and np.array_equal(fg, qg)
    with pytest.raises(KiwiHipError):
        p.set_floating_shiftrange(99, 0., 1.)


def test_synthesis_before_any_reference_is_set():
    """output_seismograms on a fresh engine (minimizer.f90:1296-1380, how a synthetic reference is made): the window of
    every receiver is the natural span of its synthetic strips (seismogram.f90:102-130)."""
    sc = Scenario(comps_list=["ned", "d", "ne", "ned", "e", "ned"])
    e = sc.oracle()
    p = sc.product()
    for rise in (0.0, 2.0):
        trial = synthetic.bilat_strike_sweep(2, step=3.0)
        from oracle import ko
        tabs = [ko.discretize(1, t, sc.effective_dt) for t in trial]
        p.set_sources([t[0] for t in tabs], [t[1] for t in tabs], [rise, rise])
        p.set_keep_synthetics(1)
        p.eval()
        for i in range(2):
            e.set_centroids(tabs[i][0], tabs[i][1], rise)
            e.calculate_seismograms()
            e.scale_seismograms()
            for ir, comps in enumerate(sc.comps):
                for k in range(len(comps)):
                    lo_o, so = e.synthetic(ir + 1, k + 1, 1)
                    lo_p, sp = p.get_synthetics(i, ir + 1, k + 1, 1)
                    a, b = max(lo_o, lo_p), min(lo_o + len(so), lo_p + len(sp))
                    assert b - a >= len(so) - 2                   # the device window covers the oracle's strip
                    assert np.max(np.abs(so[a - lo_o:b - lo_o] - sp[a - lo_p:b - lo_p])) <= SYN_RTOL * np.max(np.abs(so))
    # once references and tapers are in, misfits work on the same engine
    sc.make_references(e)
    sc.apply_setup(p, False)
    sc.apply_setup(e, True)
    p.set_keep_synthetics(0)
    trials = synthetic.bilat_strike_sweep(3, step=2.0)
    m, n, g = oracle_misfits(e, 1, trials)
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert misfit_close(pm, m, pn) and misfit_close(pg, g, glob=True)


def test_shift_and_autoshift_of_reference_seismograms():
    """shift_ref_seismogram / autoshift_ref_seismogram (minimizer_engine.f90:354-419, receiver.f90:800-832)."""
    sc = Scenario(nrec=4, comps_list=["ned", "ne", "d", "ned"])
    e, p = build(sc)
    dt = sc.gf["dt"]
    trial = synthetic.bilat_strike_sweep(1, step=2.0)
    trial[0, 0] = 1.2                                    # the synthetics arrive later than the references
    # plain shift of receiver 2 by -3 samples
    e.shift_ref_seismogram(2, -3)
    p.shift_ref_seismogram(2, -3 * dt + 0.1 * dt)
    m, n, g = oracle_misfits(e, 1, trial)
    p.set_source_params("bilateral", trial)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert misfit_close(pm, m, pn) and np.array_equal(pn[0], n[0])
    # the cross-correlations autoshift is built on (output_cross_correlations, receiver.f90:597-616)
    for ir in (1, 3):
        want_cc = e.cross_correlations(ir, -5, 5)
        first, cc = p.get_cross_correlations(ir, -5 * dt, 5 * dt)
        assert first == -5 and cc.shape == want_cc.shape == (len(sc.comps[ir - 1]), 11)
        assert np.allclose(cc, want_cc, rtol=1e-5, atol=1e-6 * np.max(np.abs(want_cc)))
    # autoshift all receivers within [-5, +5] samples against the current source
    want = [e.autoshift_ref_seismogram(ir + 1, -5, 5) * dt for ir in range(4)]
    got = p.autoshift_ref_seismogram(0, -5 * dt, 5 * dt, isrc=0)
    assert np.array_equal(got, np.array(want, np.float32))
    assert len(set(want)) > 1 and max(np.abs(want)) > 0
    m2, n2, g2 = oracle_misfits(e, 1, trial)
    p.eval()
    pm2, pn2, pg2 = p.get_misfits()
    assert misfit_close(pm2, m2, pn2) and np.array_equal(pn2[0], n2[0]) and misfit_close(pg2, g2, glob=True)
    # one receiver only
    one = e.autoshift_ref_seismogram(3, -2, 2) * dt
    assert p.autoshift_ref_seismogram(3, -2 * dt, 2 * dt)[0] == np.float32(one)


def test_sources_sharing_geometry_reuse_blended_tiles(monkeypatch):
    """Runs of consecutive trial sources with identical centroid points and times (a moment-tensor grid at a fixed
    location) are applied from ONE build of the blended tiles: results must be the bits of the unshared path."""
    sc = Scenario(nrec=5, comps_list=["ned", "ne", "d", "ned", "e"], true_type=6,
                  true_params=synthetic.mt_sdr_grid(step=30)[17])
    grid = synthetic.mt_sdr_grid(step=30)                      # 12 x 4 x 12 sources, all at one point
    other = grid[5:9].copy()
    other[:, 3] += 700.0                                       # a second location in the middle: breaks the run
    tr = np.vstack([grid[:40], other, grid[40:90], grid[90:91] * np.float32(1.0)])
    tr[60, 10] = 3.0                                           # a different rise time: different centroid times
    e, p = build(sc)
    p.set_source_params("moment_tensor", tr)
    p.set_keep_synthetics(1)
    p.eval()
    pm, pn, pg = p.get_misfits()
    syn = [p.get_synthetics(i, 1, 1, 1)[1].copy() for i in (0, 39, 41, 60, 94)]
    p.close()
    monkeypatch.setenv("KIWI_HIP_RUNS", "0")
    q = sc.product()
    sc.apply_setup(q, False)
    q.set_source_params("moment_tensor", tr)
    q.set_keep_synthetics(1)
    q.eval()
    qm, qn, qg = q.get_misfits()
    assert np.array_equal(pm, qm) and np.array_equal(pg, qg)
    for a, i in zip(syn, (0, 39, 41, 60, 94)):
        assert np.array_equal(a, q.get_synthetics(i, 1, 1, 1)[1])
    m, n, g = oracle_misfits(e, 6, tr[[0, 17, 41, 60, 94]])
    assert misfit_close(pm[[0, 41, 60, 94]], m[[0, 2, 3, 4]], pn[[0, 41, 60, 94]]) and np.all(pm[17] <= 1e-6 * pn[17])


@pytest.mark.parametrize("method", ["l2norm", "l1norm", "scalar_product", "peak"])
def test_fused_comparator_equals_separate_misfit_kernel(monkeypatch, method):
    """Few-centroid sources without rise-time fold are compared inside the accumulate kernel (no synthetics written);
    KIWI_HIP_FUSE=0 keeps the two-kernel path.  Same samples, fp64 partial sums in a different (fixed) order."""
    sc = Scenario(nrec=5, comps_list=["ned", "ne", "d", "ned", "e"], true_type=6,
                  true_params=synthetic.mt_sdr_grid(step=30)[17])
    tr = synthetic.mt_sdr_grid(step=30)[:60]
    tr[30:, 3] += 900.0
    e, p = build(sc, method)
    p.set_synthetics_factor(1.5 if method == "l1norm" else 1.0)
    e.set_synthetics_factor(1.5 if method == "l1norm" else 1.0)
    p.set_source_params("moment_tensor", tr)
    p.eval()
    pm, pn, pg = p.get_misfits()
    monkeypatch.setenv("KIWI_HIP_FUSE", "0")
    q = sc.product()
    sc.apply_setup(q, False)
    q.set_misfit_method(method)
    q.set_synthetics_factor(1.5 if method == "l1norm" else 1.0)
    q.set_source_params("moment_tensor", tr)
    q.eval()
    qm, qn, qg = q.get_misfits()
    assert np.all(np.abs(pm - qm) <= 1e-6 * np.maximum(np.abs(qm), (1.0 if arith() == "fused" else 1e-6) * qn)) and np.array_equal(pn, qn)
    m, n, g = oracle_misfits(e, 6, tr[[0, 17, 31, 59]])
    sel = pm[[0, 17, 31, 59]]
    assert np.all(np.abs(sel - m) <= MISFIT_RTOL * np.maximum(np.abs(m), (1.0 if arith() == "fused" else 1e-6) * n))


@pytest.mark.parametrize("stype", ["moment_tensor", "bilateral"])
def test_internal_chunking_cuts_through_runs(monkeypatch, stype):
    """A workspace bound of 1 MiB forces several launches per eval (and cuts runs of geometry-identical sources):
    same bits as one launch."""
    if stype == "moment_tensor":
        sc = Scenario(nrec=5, true_type=6, true_params=synthetic.mt_sdr_grid(step=30)[17])
        tr = synthetic.mt_sdr_grid(step=30)[:300]
    else:
        sc = Scenario(nrec=5)
        tr = synthetic.bilat_strike_sweep(40, step=0.5)
    e, p = build(sc)
    p.set_source_params(stype, tr)
    p.eval()
    a = p.get_misfits()
    monkeypatch.setenv("KIWI_HIP_CHUNK_MB", "1")
    q = sc.product()
    sc.apply_setup(q, False)
    q.set_source_params(stype, tr)
    q.eval()
    b = q.get_misfits()
    ms, launches = q.kernel_ms()
    assert launches[1] >= 3
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def _forward_step_factory(evaluate, sourcetype, start, mask, mins=None, maxs=None):
    """lm_forward_step (minimizer_engine.f90:806-872) restated for the checker: clamp with penalty, psm_set_subparams through
    the normalised copy of ALL parameters (source_all.f90:377-425), residuals = misfits * (1 + penalty).
    evaluate(params) -> (misfits, global)."""
    from kiwi_amd import lm
    F = np.float32
    norm = np.array(lm.PARAMS_NORM[sourcetype], F)
    idx = np.flatnonzero(mask)
    state = {"cur": np.array(start, F), "steps": 0, "global": None}

    def step(x):
        penalty = F(0)
        if mins is not None:
            for i in range(len(idx)):
                nrm = norm[idx[i]]
                if x[i] * nrm < mins[i]:
                    penalty = penalty + abs(x[i] * nrm - mins[i]) / abs(maxs[i] - mins[i])
                    x[i] = mins[i] / nrm
                if x[i] * nrm > maxs[i]:
                    penalty = penalty + abs(x[i] * nrm - maxs[i]) / abs(maxs[i] - mins[i])
                    x[i] = maxs[i] / nrm
        copy = (state["cur"] / norm).astype(F)
        copy[idx] = x
        state["cur"] = (copy * norm).astype(F)
        m, g = evaluate(state["cur"])
        state["steps"] += 1
        state["global"] = g
        return (np.asarray(m, F) * (F(1) + penalty)).astype(F)

    return step, state, (np.array(start, F)[idx] / norm[idx]).astype(F)


@pytest.mark.parametrize("limits", [False, True])
def test_minimize_lm_is_the_reference_minpack_run_with_batched_jacobians(limits):
    """minimize_lm (minimizer_engine.f90:728-874): kiwi_hip_minimize_lm evaluates each Jacobian as one batch; the plain
    optimiser kiwi_hip_lmdif -- pinned bit for bit to the reference's own sminpack lmdif by tests/golden/lm_vectors.npz
    (tests/test_lm_minpack.py) -- driving the SAME engine ONE forward step at a time through lm_forward_step restated in the
    test must take exactly the same path (info, forward steps, final source, misfit: bit for bit), and driving the CPU
    oracle a close one.  The reference's sminpack itself is not run here (oracle/_ref stays in the build container):
    its outputs on the MINPACK problems are the committed fixture kiwi_hip_lmdif is pinned to."""
    from kiwi_amd import lm, lib as klib
    from oracle import ko
    import lm_problems as P
    sc = Scenario(nrec=6)
    e, p = build(sc)
    start = sc.true_params.copy()
    start[5] += 4.0          # strike
    start[6] -= 3.0          # dip
    start[7] += 6.0          # slip-rake
    start[3] += 600.0        # depth
    names = ["depth", "strike", "dip", "slip-rake"]
    mask = np.array([n in names for n in lm.SOURCE_PARAMS["bilateral"]])
    mins = maxs = None
    if limits:               # strike may not come closer than one degree to the truth: the optimiser leans on the limit
        mins = np.array([5000.0, sc.true_params[5] + 1.0, 60.0, 100.0], np.float32)
        maxs = np.array([15000.0, 200.0, 95.0, 200.0], np.float32)
    p.set_source_params("bilateral", start[None, :])
    p.eval()
    g0 = p.get_misfits()[2][0]
    res = lm.minimize_lm(p, "bilateral", start, mask, mins, maxs)
    assert res.info in (1, 2, 3, 4) and res.misfit < (0.5 if limits else 0.05) * g0
    if limits:
        assert res.best[5] >= mins[1] - 1e-3 and abs(res.best[5] - mins[1]) < 0.2
    else:
        assert abs(res.best[5] - sc.true_params[5]) < 0.5 and abs(res.best[6] - sc.true_params[6]) < 0.5
        assert abs(res.best[3] - sc.true_params[3]) < 150.0
    # the engine is left with the last forward step (what get_source_subparams reports in the reference)
    m, n, g = p.get_misfits()
    assert g[0] == np.float32(res.misfit)

    def run_reference_minpack(evaluate):
        step, state, x0 = _forward_step_factory(evaluate, "bilateral", start, mask, mins, maxs)
        st = dict(P.SETTINGS["minimize_lm"])
        # one point per call of `step`, in lmdif's order
        x, fvec, info, nfev = P.run_product(klib.load(), klib, "kiwi", len(m[0]), len(x0), x0, step, st)
        return (4 if info == 8 else info), state["steps"], state["cur"], state["global"]

    def eval_product(params):
        p.set_source_params("bilateral", params[None, :])
        p.eval()
        mm, _, gg = p.get_misfits()
        return mm[0], gg[0]

    info, steps, last, glob = run_reference_minpack(eval_product)
    assert (info, steps) == (res.info, res.iterations)
    assert np.array_equal(last.view(np.uint32), res.params.view(np.uint32)) and np.float32(glob) == np.float32(res.misfit)

    def eval_oracle(params):
        mm, _, gg = oracle_misfits(e, 1, params[None, :])
        return mm[0], gg[0]

    info_o, steps_o, last_o, glob_o = run_reference_minpack(eval_oracle)
    assert info_o in (1, 2, 3, 4)
    scale = np.array(lm.PARAMS_NORM["bilateral"], np.float32)
    assert np.max(np.abs(last_o - res.params) / scale) < 2e-3 and abs(glob_o - res.misfit) < 1e-3 * g0 + 1e-5


@pytest.mark.parametrize("mode", ["tapered", "untapered", "synthesis_only"])
def test_peak_amplitudes_and_arias_intensities(mode):
    """get_peak_amplitudes / get_arias_intensities (receiver.f90:512-594, comparator.f90:519-625): per enabled receiver the
    peak velocity / acceleration vector norm and the Arias intensity of the (tapered) synthetics.  The values are built on
    sample differences, so the synthetics' tolerance shows amplified: 1e-4 relative."""
    comps = ["ned", "d", "ar", "nd", "une", "rau"]
    sc = Scenario(nrec=6, comps_list=comps)
    e = sc.oracle()
    p = sc.product()
    if mode != "synthesis_only":
        sc.make_references(e)
        if mode == "untapered":
            sc.tapers = {}
        sc.apply_setup(e, True)
        sc.apply_setup(p, False)
    for rise, factor in ((0.0, 1.0), (2.0, 0.7)):
        e.set_synthetics_factor(factor)
        p.set_synthetics_factor(factor)
        trials = synthetic.bilat_strike_sweep(3, step=4.0)
        trials[:, 13] = rise
        p.set_source_params("bilateral", trials)
        for i in (2, 0):
            if mode != "tapered":              # strip spans remember earlier sources in the reference: fresh engine each time
                e.close()
                e = sc.oracle()
                sc.apply_setup(e, True)
                e.set_synthetics_factor(factor)
            e.set_source_params(1, trials[i])
            for want, got in ((e.peak_amplitudes(1), p.get_peak_amplitudes(1, i)), (e.peak_amplitudes(2), p.get_peak_amplitudes(2, i)),
                              (e.arias_intensities(), p.get_arias_intensities(i))):
                assert len(got) == len(want) == 6 and np.all(want > 0)
                assert np.allclose(got, want, rtol=1e-4, atol=0), (mode, rise, i, got, want)
    if mode == "tapered":                     # misfits are unaffected by the diagnostic evaluations in between
        m, n, g = oracle_misfits(e, 1, trials)
        p.eval()
        assert misfit_close(p.get_misfits()[2], g, glob=True)
        p.switch_receiver(2, False)
        assert len(p.get_arias_intensities(1)) == 5
        with pytest.raises(KiwiHipError, match="differentiate argument must be 1"):
            p.get_peak_amplitudes(0)
        p.set_misfit_filter(1, [0.01, 0.02, 0.1, 0.2], [0, 1, 1, 0])
        with pytest.raises(KiwiHipError, match="not available with a misfit filter"):
            p.get_arias_intensities(0)


def test_descriptor_rows_of_group_starts_only(monkeypatch):
    """geometry_kernel writes the full descriptor row only where accumulate_grouped_kernel starts a group and just the
    coefficient line elsewhere.  With the table poisoned before every evaluation (KIWI_HIP_POISON) a wrong prediction of
    the group starts would show as a wild address; the sources here have same-point runs that split into several groups:
    80 time steps at one point (shift spread beyond the LDS halo, more than 64 centroids) and a bilateral source."""
    monkeypatch.setenv("KIWI_HIP_POISON", "1")
    sc = Scenario(nrec=5, L=700)
    e, p = build(sc)
    mt = np.array([[0.3, 200., -300., 9000., 1e18, -5e17, 2e17, 3e17, -1e17, 4e17, 40.0],      # rise time 40 s: 81 time steps
                   [0.0, 0., 0., 10000., 1e18, 1e18, -2e18, 0., 0., 0., 33.3]], np.float32)
    from oracle import ko
    assert len(ko.discretize(6, mt[0], sc.effective_dt)[0]) > 70
    m, n, g = oracle_misfits(e, 6, mt)
    p.set_source_params("moment_tensor", mt)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert misfit_close(pm, m, pn) and misfit_close(pg, g, glob=True)
    trials = synthetic.bilat_strike_sweep(4, step=5.0)
    trials[:, 13] = 6.0                                  # longer rise time: more time steps per sub-fault
    m, n, g = oracle_misfits(e, 1, trials)
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert misfit_close(pm, m, pn) and misfit_close(pg, g, glob=True)


def test_point_lp_source():
    """point_lp (source type 3, source_point_lp.f90): band-limited point source, ~40 time steps at one point."""
    sc = Scenario(nrec=4)
    e, p = build(sc)
    trials = np.array([[0.0, 0., 0., 10000., 7e18, 1., 0., -1., 1., 1., 1., 20., 10.],
                       [0.7, 300., -500., 9000., 3e18, 0.3, -0.8, 0.5, 0.1, -0.2, 0.7, 7.5, 3.0]], np.float32)
    m, n, g = oracle_misfits(e, 3, trials)
    p.set_source_params("point_lp", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert misfit_close(pm, m, pn) and misfit_close(pg, g, glob=True) and np.array_equal(pn[0], n[0])


@pytest.mark.parametrize("method", ["l2norm", "l1norm", "floating_l1norm"])
def test_untapered_comparator_fresh_evaluation_semantics(method):
    """Without misfit tapers (the reference's own benchmark/kiwibench.py runs floating_l1norm that way) a norm runs over
    the union of the reference's data span and the data span of the synthetic strip (comparator.f90:798-800).  In the
    reference that strip never shrinks, so the span depends on the sources evaluated before; the device gives every
    source the span of a FRESH evaluation -- compared here with a fresh oracle engine per trial source."""
    sc = Scenario(nrec=4, comps_list=["ned", "ne", "d", "ned"])
    e0 = sc.oracle()
    sc.make_references(e0)
    mid = {"l2norm": 1, "l1norm": 2, "floating_l1norm": 8}[method]
    dt = sc.gf["dt"]
    trials = synthetic.bilat_strike_sweep(4, step=2.0)
    trials[:, 0] = [-1.1, 0.0, 0.6, 2.1]                    # origin times move the strips against the references
    trials[2, 13] = 0.0                                     # and one without rise time
    want_m, want_n, want_g, want_s = [], [], [], []
    for t in trials:
        e = sc.oracle()
        for (ir, k), (lo, d) in sc.refs.items():
            e.set_reference(ir, k, lo, d)
        e.set_misfit_method(mid)
        if mid == 8:
            for ir in range(4):
                e.set_floating_shiftrange(ir + 1, -2, 2)
        e.set_source_params(1, t)
        m, n, g = e.get_misfits()
        want_m.append(m); want_n.append(n); want_g.append(g)
        want_s.append([e.floating_shift(ir + 1) * dt for ir in range(4)])
        e.close()
    p = sc.product()
    for (ir, k), (lo, d) in sc.refs.items():
        p.set_ref_seismogram(ir, k, lo, d)
    p.set_misfit_method(method)
    if mid == 8:
        p.set_floating_shiftrange(0, -2 * dt, 2 * dt)
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert misfit_close(pm, np.array(want_m), pn) and misfit_close(pg, np.array(want_g, np.float32), glob=True)
    assert np.array_equal(pn, np.array(want_n))
    if mid == 8:
        assert np.array_equal(p.get_floating_shifts(), np.array(want_s, np.float32))


@pytest.mark.parametrize("fused", [False, True])
def test_odd_window_lengths_and_offsets(fused):
    """Taper spans of arbitrary length / start (not multiples of the 4-sample lane chunks), different per receiver."""
    sc = Scenario(nrec=5, comps_list=["ned", "ne", "d", "ned", "e"], true_type=6 if fused else 1,
                  true_params=synthetic.mt_sdr_grid(step=30)[17] if fused else None)
    e, p = build(sc)
    dt = sc.gf["dt"]
    for ir in range(5):
        lo, d = sc.refs[(ir + 1, 1)]
        t0, t1 = (lo + 3 + ir) * dt + 0.13, (lo + len(d) - 11 - 2 * ir) * dt - 0.21
        x, y = [t0, t0 + 4.7, t1 - 6.1, t1], [0., 1., 1., 0.]
        e.set_taper(ir + 1, x, y)
        p.set_misfit_taper(ir + 1, x, y)
    if fused:
        trials = synthetic.mt_sdr_grid(step=30)[40:52]
        m, n, g = oracle_misfits(e, 6, trials)
        p.set_source_params("moment_tensor", trials)
    else:
        trials = synthetic.bilat_strike_sweep(5, step=1.5)
        m, n, g = oracle_misfits(e, 1, trials)
        p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert np.array_equal(pn[0], n[0]) and misfit_close(pm, m, pn) and misfit_close(pg, g, glob=True)
    lens = {p.get_synthetics(0, ir + 1, 1, 2)[1].size for ir in range(5)}
    assert any(v % 4 for v in lens)


def test_all_component_letters():
    """Every component letter of receiver.f90:294-351: w s u l c are the negated e n d r a."""
    comps = ["wsu", "lc", "ard", "ne", "cw", "uln"]
    sc = Scenario(comps_list=comps)
    e, p = build(sc)
    trials = synthetic.bilat_strike_sweep(3, step=2.0)
    m, n, g = oracle_misfits(e, 1, trials)
    p.set_source_params("bilateral", trials)
    p.set_keep_synthetics(1)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert pm.shape[1] == sum(len(c) for c in comps)
    assert np.array_equal(pn[0], n[0]) and misfit_close(pm, m, pn) and misfit_close(pg, g, glob=True)
    # w = -e, s = -n, u = -d at one receiver pair sharing the geometry? (receivers differ) -> check against the oracle's traces
    e.set_source_params(1, trials[0])
    e.get_misfits()
    for ir, cs in enumerate(comps):
        for k in range(len(cs)):
            lo_o, so = e.synthetic(ir + 1, k + 1, 1)
            lo_p, sp = p.get_synthetics(0, ir + 1, k + 1, 1)
            a, b = max(lo_o, lo_p), min(lo_o + len(so), lo_p + len(sp))
            assert np.max(np.abs(so[a - lo_o:b - lo_o] - sp[a - lo_p:b - lo_p])) <= SYN_RTOL * np.max(np.abs(so))


def test_example_inversion_recovers_the_source():
    """examples/invert_bilateral.py: grid search + bootstrap + LM on noisy synthetic data through the public API."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("invert_bilateral", os.path.join(os.path.dirname(os.path.dirname(__file__)),
                                                                                    "examples", "invert_bilateral.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    true, grid, res = mod.main(nrec=12, L=512, verbose=False)
    assert abs(grid.best_source[5] - true[5]) <= 3 and abs(grid.best_source[6] - true[6]) <= 3
    assert abs(res.best[5] - true[5]) < 1.5 and abs(res.best[6] - true[6]) < 1.5 and abs(res.best[7] - true[7]) < 3
    assert res.misfit <= grid.get_best_misfit() + 1e-4


def test_many_small_evaluations_recycle_timing_events():
    """Thousands of evaluations without anybody reading kernel_ms(): the event list is recycled, results stay right."""
    sc = Scenario(nrec=2, L=96)
    e, p = build(sc)
    trials = synthetic.bilat_strike_sweep(2, step=3.0)
    m, n, g = oracle_misfits(e, 1, trials)
    p.set_source_params("bilateral", trials)
    for _ in range(2500):
        p.eval()
    pm, pn, pg = p.get_misfits()
    assert misfit_close(pm, m, pn) and misfit_close(pg, g, glob=True)
    ms, launches = p.kernel_ms()
    assert 0 < launches[1] < 2500


def test_reference_probes_plain_tapered_filtered():
    """output_seismograms ... references plain|tapered|filtered (receiver.f90:618-680) against the oracle's probes."""
    sc = Scenario(nrec=3, comps_list=["ned", "d", "ne"])
    e, p = build(sc)
    fx, fy = [0.02, 0.04, 0.15, 0.3], [0., 1., 1., 0.]
    for ir in range(3):
        e.set_filter(ir + 1, fx, fy)
        p.set_misfit_filter(ir + 1, fx, fy)
    trial = synthetic.bilat_strike_sweep(1, step=2.0)
    e.set_source_params(1, trial[0])
    e.get_misfits()
    p.set_source_params("bilateral", trial)
    p.eval()
    for ir, comps in enumerate(sc.comps):
        for k in range(len(comps)):
            lo0, d0 = sc.refs[(ir + 1, k + 1)]
            lo, d = p.get_reference(ir + 1, k + 1, 1)
            assert lo == lo0 and np.array_equal(d, d0)
            for which, tol in ((2, 0.0), (3, 2e-5)):
                lo_p, dp = p.get_reference(ir + 1, k + 1, which)
                lo_o, do = e.reference(ir + 1, k + 1, which)
                a, b = max(lo_o, lo_p), min(lo_o + len(do), lo_p + len(dp))
                assert b - a >= len(dp) - 1
                assert np.max(np.abs(do[a - lo_o:b - lo_o] - dp[a - lo_p:b - lo_p])) <= tol * np.max(np.abs(do)) + 0.0
    with pytest.raises(KiwiHipError):
        p.get_reference(1, 9, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("L", [700, 1500])
def test_two_sources_per_workgroup_is_bit_identical(monkeypatch, L):
    """accumulate_duo_kernel (two consecutive trial sources of equal structure per workgroup: node rows loaded once,
    blended twice) against the one-source kernels, synthetics bit for bit: neighbours in the same cells (fine strike
    steps), neighbours in different cells (coarse steps: the two tile sets are built one after the other), a time sweep
    (integer shifts differ between the two: other tile origin), an odd number of sources, a source of another structure
    in between (no mate), receivers the kernel does not take (one component block only) and traces missing from the
    database (those pairs go to the grouped kernel)."""
    sc = Scenario(nrec=6, L=L, comps_list=["ned", "ned", "d", "ne", "ned", "ar"], variant="probe")
    _knock_out(sc, [(4, iz, 6) for iz in range(5)])
    fine = synthetic.bilat_strike_sweep(6, step=0.05)
    coarse = synthetic.bilat_strike_sweep(4, step=9.0)
    coarse[:, 1] += 2500.0 * np.arange(4)                      # north-shift: other cells of the database
    times = synthetic.bilat_strike_sweep(4, step=0.0)
    times[:, 0] += np.array([0.0, 0.3, 0.55, 1.3], np.float32)
    other = synthetic.bilat_strike_sweep(1, step=0.0)
    other[:, 9] *= 0.5                                          # shorter rupture: another number of sub-faults
    trials = np.vstack([fine, coarse, other, times, fine[:3]]).astype(np.float32)
    assert len(trials) % 2 == 0 and len(trials) == 18
    # two aligned groups of FOUR sources of a time sweep (origin times up to seven samples apart: their groups share one tile
    # origin, that of the largest shift), then an odd count: the last source has no mate
    times8 = synthetic.bilat_strike_sweep(8, step=0.0)
    times8[:, 0] += np.array([0.0, 0.3, 0.55, 1.3, 2.1, 2.15, 2.6, 3.4], np.float32)
    trials = np.vstack([trials, fine[1:3], times8, fine[:1]]).astype(np.float32)
    assert len(trials) == 29
    res = {}
    sc.oracle()                                                 # (packs the database the product is handed)
    for mode in ("quad", "duo", "single", "direct"):
        monkeypatch.delenv("KIWI_HIP_ACCUM", raising=False)
        monkeypatch.setenv("KIWI_HIP_DUO", {"quad": "4", "duo": "2"}.get(mode, "0"))
        if mode == "direct":
            monkeypatch.setenv("KIWI_HIP_ACCUM", "direct")
        p = sc.product()
        p.set_source_params("bilateral", trials)
        p.set_keep_synthetics(1)
        p.eval()
        res[mode] = [p.get_synthetics(s, ir, k, 1)[1] for s in range(len(trials)) for ir in range(1, 7)
                     for k in range(1, len(sc.comps[ir - 1]) + 1)]
        p.close()
    assert len(res["quad"]) == len(res["duo"]) == len(res["single"]) == len(res["direct"]) > 300
    for q, a, b, c in zip(res["quad"], res["duo"], res["single"], res["direct"]):
        assert same_bits(q, a) and same_bits(a, b) and same_bits(b, c)
    assert sum(1 for a in res["duo"] if np.any(a != 0)) > 150
    # and through the fused comparator: misfits against the oracle
    monkeypatch.setenv("KIWI_HIP_DUO", "4")
    monkeypatch.delenv("KIWI_HIP_ACCUM", raising=False)
    e, p = build(sc)
    tr = np.vstack([trials[:15], times8, fine[3:5]])            # (no repeated source: those would share synthetics instead)
    m, n, g = oracle_misfits(e, 1, tr)
    p.set_source_params("bilateral", tr)
    p.eval()
    pm, pn, pg = p.get_misfits()
    assert misfit_close(pm, m, pn) and misfit_close(pg, g, glob=True)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["ampspec_l2norm", "l2norm"])
def test_spectral_results_do_not_depend_on_which_transform_the_neighbours_take(method):
    """In-LDS or library transform is decided per (source, slot) pair by the pair's own length.  A batch with ONE pair the in-LDS
    kernels do not take (8 samples: a three-sample reference at a receiver the second source does not reach) used to send every
    pair of the batch -- and the reference variants made in that call -- through the library transforms, so a source's spectral
    misfits and norm factors differed by an ulp between a batch and an evaluation on its own (found by the randomised sweep).
    Batch against one source at a time, bit for bit; spectral norm and time-domain norm on filtered traces."""
    sc = Scenario(nrec=2)
    e = sc.oracle()
    sc.make_references(e)
    e.close()
    dt = sc.gf["dt"]
    p = sc.product()
    for (ir, k), (lo, d) in sc.refs.items():
        if ir == 2:
            lo, d = 5, np.array([0.5, -1.0, 0.25], np.float32) * float(np.max(np.abs(d)))
        p.set_ref_seismogram(ir, k, lo, d)
    p.set_misfit_taper(1, *sc.tapers[1])
    p.set_misfit_taper(2, [4.5 * dt, 5.0 * dt, 7.0 * dt, 7.5 * dt], [0., 1., 1., 0.])
    p.set_misfit_method(method)
    if method == "l2norm":
        for ir in (1, 2):
            p.set_misfit_filter(ir, [0.01, 0.03, 0.25, 0.4], [0., 1., 1., 0.])
    near = synthetic.bilat_strike_sweep(3, step=2.0)
    far = synthetic.bilat_strike_sweep(2, step=2.0)
    far[:, 1] += 400e3                                          # north-shift: no receiver inside the database's range
    trials = np.vstack([near[:1], far[:1], near[1:], far[1:]]).astype(np.float32)
    p.set_source_params("bilateral", trials)
    p.eval()
    bm, bn, bg = [x.copy() for x in p.get_misfits()]
    assert np.all(bm[[1, 4]] >= 0) and np.all(np.isfinite(bm)) and np.any(bm[0] != bm[2])
    for i in range(len(trials)):
        p.set_source_params("bilateral", trials[i:i + 1])
        p.eval()
        m, n, g = p.get_misfits()
        assert m.tobytes() == bm[i:i + 1].tobytes() and n.tobytes() == bn[i:i + 1].tobytes() and g.tobytes() == bg[i:i + 1].tobytes(), i
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tapers", ["mixed", "none"])
@pytest.mark.parametrize("what", ["ampspec_l2norm", "ampspec_l1norm", "filter_l2norm", "filter_l1norm"])
def test_spectral_norms_without_a_taper_fresh_evaluation_semantics(what, tapers):
    """Amplitude-spectrum norms and misfit filters on receivers WITHOUT a misfit taper (comparator.f90:861-886 over the whole padded
    probes: zeros before the data span, the end value repeated behind it, :259-265,320-324; common span of a pair
    allowed_span(union of the data spans, max of twice the data lengths), :464-486,1092-1109).  The reference's probe spans follow every
    source evaluated before; the device gives every (source, slot) pair the span of a FRESH engine, as for the un-tapered time-domain
    norms -- compared with a fresh oracle engine per trial source: misfits and norm factors (the reference's spectrum / filtered trace
    belongs to the pair there, so the norm factor does too).  `mixed`: receivers 1, 2 keep their tapers (variant tables), 3 and 4 have
    none (per-pair reference arrays) in the same batch; origin times move the strips against the references (another span, another
    number of repeated end values, two transform lengths), with and without rise time."""
    sc = Scenario(nrec=4, comps_list=["ned", "ne", "d", "ned"])
    e0 = sc.oracle()
    sc.make_references(e0)
    dt = sc.gf["dt"]
    method = what[7:] if what.startswith("filter_") else what
    mid = {"ampspec_l2norm": 3, "ampspec_l1norm": 4, "l2norm": 1, "l1norm": 2}[method]
    fx, fy = [0.01, 0.03, 0.25, 0.4], [0., 1., 1., 0.]
    filtered = (1, 3, 4) if what.startswith("filter_") else ()           # receiver 2: no filter (compared on the plain arrays)
    tapered = (1, 2) if tapers == "mixed" else ()
    trials = synthetic.bilat_strike_sweep(5, step=2.0)
    trials[:, 0] = [-1.1, 0.0, 0.6, 2.1, 400.0]              # (the last one far behind the references: a longer union, the next transform length)
    trials[2, 13] = 0.0
    trials[4, 13] = 9.0                                      # a long rise time: a longer strip
    p = sc.product()
    for (ir, k), (lo, d) in sc.refs.items():
        p.set_ref_seismogram(ir, k, lo, d)
    for ir in tapered:
        p.set_misfit_taper(ir, *sc.tapers[ir])
    for ir in filtered:
        p.set_misfit_filter(ir, fx, fy)
    p.set_misfit_method(method)
    p.set_source_params("bilateral", trials)
    p.eval()
    pm, pn, pg = p.get_misfits()
    ntr = set()
    for i, t in enumerate(trials):
        e = sc.oracle()
        for (ir, k), (lo, d) in sc.refs.items():
            e.set_reference(ir, k, lo, d)
        for ir in tapered:
            e.set_taper(ir, *sc.tapers[ir])
        for ir in filtered:
            e.set_filter(ir, fx, fy)
        e.set_misfit_method(mid)
        e.set_source_params(1, t)
        m, n, g = e.get_misfits()
        scales = slot_scales(e, sc.comps, dt)
        ntr.update(int(x) for x in scales[0])
        ok, ratio = spectral_close(method, dt, pm[i], m, n, scales, pn[i])
        assert ok, (i, ratio, pm[i], m, pn[i], n)
        assert abs(pg[i] - g) <= 2e-5 * abs(g), (i, pg[i], g)
        e.close()
    assert len(ntr) >= 2 and np.all(pn > 0)
    # one source at a time gives the same bits as the batch (nothing of a pair depends on its neighbours)
    p.set_source_params("bilateral", trials[3:4])
    p.eval()
    qm, qn, qg = p.get_misfits()
    assert qm.tobytes() == pm[3:4].tobytes() and qn.tobytes() == pn[3:4].tobytes()
    # output_seismogram_spectra of an un-tapered receiver: the spectra of the PAIR (current source, slot), reference side included
    e = sc.oracle()
    for (ir, k), (lo, d) in sc.refs.items():
        e.set_reference(ir, k, lo, d)
    for ir in tapered:
        e.set_taper(ir, *sc.tapers[ir])
    for ir in filtered:
        e.set_filter(ir, fx, fy)
    e.set_misfit_method(mid)
    e.set_source_params(1, trials[3])
    e.get_misfits()
    for ir, k in ((4, 1), (3, 1), (1, 2)):
        for synth in (True, False):
            odf, want = e.amp_spectrum(ir, k, synth, False)
            df, got = p.get_amp_spectrum(ir, k, "synthetics" if synth else "references", False, isrc=0)
            assert abs(df - odf) <= 1e-7 * odf and got.shape == want.shape, (ir, k, synth)
            assert np.allclose(got, want, rtol=SPEC_RTOL, atol=2e-6 * want.max()), (ir, k, synth)
    e.close()
    if tapers == "none":
        with pytest.raises(KiwiHipError, match="name a source"):
            p.get_amp_spectrum(4, 1, "references", False, isrc=-1)
    if filtered:
        # get_reference(filtered) of an un-tapered receiver: the filtered reference of the pair (current source, slot) inside the window
        e = sc.oracle()
        for (ir, k), (lo, d) in sc.refs.items():
            e.set_reference(ir, k, lo, d)
        for ir in tapered:
            e.set_taper(ir, *sc.tapers[ir])
        for ir in filtered:
            e.set_filter(ir, fx, fy)
        e.set_misfit_method(mid)
        e.set_source_params(1, trials[3])
        e.get_misfits()
        for ir, k in ((4, 1), (4, 3), (3, 1)):
            lo_p, dp = p.get_reference(ir, k, 3)
            lo_o, do = e.reference(ir, k, 3)
            a, b = max(lo_o, lo_p), min(lo_o + len(do), lo_p + len(dp))
            assert b - a >= 264                                 # (at least the reference's data span)
            assert np.max(np.abs(do[a - lo_o:b - lo_o] - dp[a - lo_p:b - lo_p])) <= 2e-5 * np.max(np.abs(do))
        e.close()
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("method,with_filter", [("ampspec_l2norm", False), ("l2norm", True)])
def test_spectral_results_of_folded_sources_do_not_depend_on_the_batch(method, with_filter):
    """strip_fold grows a strip by the taps of the source's OWN rise time (receiver.f90:868-897), so the transform
    length of a probe pair of an `mt_eikonal` source follows that rise time, not the longest one of the batch it is evaluated
    in (fft_size_kernel): mixed rise times -- one of them long enough to change the transform length -- alone == in a batch
    == in another order == through the one-call in pieces, bit for bit; and each agrees with a fresh oracle engine."""
    import os
    from oracle import ko
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eikonal_vectors.npz"))
    prof = G["rupture_profile"]
    sc = Scenario(nz=6, L=200)
    e, p = build(sc)
    mid = {"ampspec_l2norm": 3, "l2norm": 1}[method]
    p.set_misfit_method(method)
    fx, fy = [0.01, 0.03, 0.25, 0.4], [0., 1., 1., 0.]
    if with_filter:
        for ir in range(1, sc.nrec + 1):
            p.set_misfit_filter(ir, fx, fy)
    rng = np.random.default_rng(11)
    rises = [0.0, 1.5, 30.0, 0.6, 60.0]
    trials = []
    for i, rise in enumerate(rises):
        common = [0.2 * i, 300.0 * i, -200.0 * i, 10500.0 + 300 * i]
        bord = [100.0, -50.0, 2500.0 + 400 * i]
        nukl = [500.0 - 300 * i, 200.0]
        trials.append(common + [1.0, 80.0 + 5 * i, 70.0] + bord + nukl + [0.8] + list(rng.standard_normal(6) * 7e17) + [rise])
    trials = np.array(trials, np.float32)
    cp = np.array([[0, 0, 6500.0], [0, 0, 15500.0]], np.float32)
    cn = np.array([[0, 0, -1.0], [0, 0, 1.0]], np.float32)
    p.set_source_crust(prof, G["origin_profile"])
    p.set_source_constraints(cp, cn)
    p.set_source_params("mt_eikonal", trials)
    p.eval()
    bm, bn, bg = [x.copy() for x in p.get_misfits()]
    assert len({x.tobytes() for x in bn}) > 1            # more than one transform length in this batch
    for i in range(len(trials)):                          # alone
        p.set_source_params("mt_eikonal", trials[i:i + 1])
        p.eval()
        am, an, ag = p.get_misfits()
        assert am[0].tobytes() == bm[i].tobytes() and an[0].tobytes() == bn[i].tobytes() and ag[0] == bg[i], i
    perm = [4, 0, 2, 1, 3]
    p.set_source_params("mt_eikonal", trials[perm])
    p.eval()
    qm, qn, qg = p.get_misfits()
    assert np.array_equal(qm, bm[perm]) and np.array_equal(qn, bn[perm]) and np.array_equal(qg, bg[perm])
    rm, rn, rg, rs = p.misfits_for_params("mt_eikonal", trials, 2)
    assert np.array_equal(rm, bm) and np.array_equal(rn, bn) and np.array_equal(rg, bg) and not rs.any()
    oprof = ko.crust_profile(*np.split(prof, [8, 16, 24]))
    for i in (1, 2, 4):
        ef = sc.oracle()
        sc.apply_setup(ef, True)
        ef.set_misfit_method(mid)
        if with_filter:
            for ir in range(1, sc.nrec + 1):
                ef.set_filter(ir, fx, fy)
        c, mo, ri, _ = ko.discretize_eikonal(5, trials[i], sc.effective_dt, oprof, cp, cn)
        ef.set_centroids(c, mo, ri)
        om, on, og = ef.get_misfits()
        # no fixed figure: 1e-6 of max(norm factor, misfit) + what fp32 transforms of the pair's length explain (tests/common.py)
        ok, ratio = spectral_close(method, sc.gf["dt"], bm[i], om, on, slot_scales(ef, sc.comps, sc.gf["dt"]), bn[i])
        assert ok, (i, ratio, np.max(np.abs(bm[i] - om) / on), np.max(np.abs(bn[i] - on) / on))
        ef.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bilinear", [True, False])
@pytest.mark.parametrize("mode", ["quad", "single", "cellw", "cell"])
def test_compact_descriptors_equal_descriptor_rows(monkeypatch, bilinear, mode):
    """Databases whose components of a node all start at the same sample and whose rows all end in zero (what trace_pack leaves of
    traces that die out inside the time range): geometry_kernel hands the accumulate kernels FOUR numbers per record instead of a
    512-byte descriptor row per group start, the kernels rebuild the row (desc_expand).  Same synthetics bit for bit as with the
    rows (KIWI_HIP_COMPACT=0), for every kernel family, nearest-neighbour and bilinear interpolation; the row table is poisoned, so a
    kernel that still read it would load from wild addresses."""
    monkeypatch.setenv("KIWI_HIP_POISON", "1")
    monkeypatch.setenv("KIWI_HIP_DUO", "4" if mode == "quad" else "0")
    monkeypatch.setenv("KIWI_HIP_CELL", "1" if mode.startswith("cell") else "0")
    monkeypatch.setenv("KIWI_HIP_CELL_WAVE", "1" if mode == "cellw" else "0")
    sc = Scenario(nrec=5, L=700, bilinear=bilinear, comps_list=["ned", "ned", "ar", "ned", "ned"], variant="probe")
    sc.oracle()                                                 # (packs the database the product is handed)
    first, nsamp, data = sc.odb.dense_tables()
    assert np.all(first == first[:, :, :1]) and np.all(nsamp > 0)          # one start per node, nothing missing
    fine = synthetic.bilat_strike_sweep(8, step=0.05)
    far = synthetic.bilat_strike_sweep(4, step=7.0)
    far[:, 1] += 2500.0 * np.arange(4)
    trials = np.vstack([fine, far, fine[:3]]).astype(np.float32)
    res = {}
    for compact in ("1", "0"):
        monkeypatch.setenv("KIWI_HIP_COMPACT", compact)
        p = sc.product()
        p.set_source_params("bilateral", trials)
        p.set_keep_synthetics(1)
        p.eval()
        res[compact] = [p.get_synthetics(s, ir, k, 1)[1] for s in range(len(trials)) for ir in range(1, 6)
                        for k in range(1, len(sc.comps[ir - 1]) + 1)]
        p.close()
    assert len(res["1"]) == len(res["0"]) > 150 and sum(1 for a in res["1"] if np.any(a != 0)) > 100
    for a, b in zip(res["1"], res["0"]):
        assert same_bits(a, b)
