#!/usr/bin/env python3
"""Randomised parity sweep (manual: `python tests/fuzz_gpu_parity.py [seconds] [seed]` on a GPU box): random databases,
receiver sets, interpolation modes, source types, tapers and norms, each compared with the CPU oracle at the tolerances
of tests/test_gpu_parity.py / tests/common.py.  A fixed slice of it runs under pytest (tests/test_gpu_fuzz_slice.py); the
cases the long runs have found are pinned as regular tests."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kiwi_amd import synthetic  # noqa: E402
from tests.common import Scenario, oracle_misfits, slot_scales, spectral_close, arith  # noqa: E402

class _Recorded:
    """an oracle engine that remembers how it was configured, so that a FRESH engine can be configured the same way for every trial
    source (the device gives every source the spans of a fresh evaluation: DESIGN.md 6)"""
    def __init__(self, eng):
        self.eng, self.calls = eng, []

    def __getattr__(self, k):
        a = getattr(self.eng, k)
        if k.startswith("set_") and k not in ("set_source_params", "set_centroids"):
            def f(*args, **kw):
                self.calls.append((k, args, kw))
                return a(*args, **kw)
            return f
        return a

    def fresh(self, sc):
        e = sc.oracle()
        for k, args, kw in self.calls:
            getattr(e, k)(*args, **kw)
        return e


FAMILIES = ["ac", "rl", "du", "ns", "ew"]            # a component and its negated twin exclude each other (receiver.f90:255-270)


def one_case(rng, verbose):
    ng = int(rng.choice([8, 10]))
    L = int(rng.choice([96, 200, 256, 700, 1500, 2300]))
    nrec = int(rng.integers(1, 7))
    comps = ["".join(str(rng.choice(list(FAMILIES[k]))) for k in rng.choice(5, size=int(rng.integers(1, 4)), replace=False))
             for _ in range(nrec)]
    bil = bool(rng.integers(0, 2))
    variant = str(rng.choice(["probe", "static"]))
    edt = float(rng.choice([0.5, 1.0]))
    depths = rng.choice([0., 0., 300., 900.], nrec)
    sc = Scenario(nx=int(rng.integers(8, 14)), nz=int(rng.integers(4, 7)), ng=ng, L=L, nrec=nrec, variant=variant, bilinear=bil,
                  effective_dt=edt, comps_list=comps, depths=depths, taper_ramp=float(rng.uniform(2, 12)),
                  dmin=float(rng.uniform(101e3, 108e3)), dspan=float(rng.uniform(10e3, 30e3)))
    e = sc.oracle()
    sc.make_references(e)
    # tapers: random sub-windows of the references
    dt = sc.gf["dt"]
    for ir in range(nrec):
        lo, d = sc.refs[(ir + 1, 1)]
        a = (lo + rng.integers(0, max(1, len(d) // 4))) * dt + rng.uniform(0, dt)
        b = (lo + len(d) - 1 - rng.integers(0, max(1, len(d) // 4))) * dt - rng.uniform(0, dt)
        r1, r2 = rng.uniform(0.5, 6), rng.uniform(0.5, 6)
        if b - a > r1 + r2 + 2 * dt:
            sc.tapers[ir + 1] = ([a, a + r1, b - r2, b], [0., 1., 1., 0.])
    # a fifth of the cases: some (or all) receivers WITHOUT a misfit taper -- the norms then run over the union of the data spans
    # (time domain) or over the padded probes of the pair (spectral norms, filters); spans of a fresh evaluation: one source
    notaper = rng.random() < float(os.environ.get("KIWI_FUZZ_NOTAPER", "0.2"))
    if arith() == "fused":
        # (exact contract only: where a rise time folds an un-tapered strip, its span is cut at the trailing run of EQUAL values
        # (strip_dataspan) -- a decision on bit patterns, which the fused contract's last-bit differences can move by a sample:
        # seen once in 8227 cases, 8e-5 of a spectral L1 norm; INTEGRATION.md "Arithmetic contract")
        notaper = False
    if notaper:
        for ir in rng.choice(nrec, size=int(rng.integers(1, nrec + 1)), replace=False):
            sc.tapers.pop(int(ir) + 1, None)
    # a quarter of the cases: traces missing from the database at nodes the sources use (single components or whole nodes;
    # the references stay those of the complete database): the reference leaves a centroid at the first trace it does not
    # find (seismogram.f90:171-250)
    holes = rng.random() < float(os.environ.get("KIWI_FUZZ_HOLES", "0.25"))
    if holes:
        e.close()
        nxg, nzg = sc.gf["nsamp"].shape[:2]
        for _ in range(int(rng.integers(1, 9))):
            ix, iz, ig = int(rng.integers(0, min(nxg, 10))), int(rng.integers(1, min(nzg, 4))), int(rng.integers(0, ng))
            if rng.random() < 0.25:
                sc.gf["nsamp"][ix, iz, :] = 0
            else:
                sc.gf["nsamp"][ix, iz, ig] = 0
        e = sc.oracle()
    sc.apply_setup(e, True)
    p = sc.product()
    sc.apply_setup(p, False)
    method = str(rng.choice(["l2norm", "l1norm", "scalar_product", "peak", "floating_l2norm", "floating_l1norm",
                             "ampspec_l2norm", "ampspec_l1norm"]))
    mid = {"l2norm": 1, "l1norm": 2, "ampspec_l2norm": 3, "ampspec_l1norm": 4, "scalar_product": 5, "peak": 6,
           "floating_l2norm": 7, "floating_l1norm": 8}[method]
    spectral = mid in (3, 4)
    filtered = (spectral or mid in (1, 2)) and rng.random() < 0.4
    if spectral or filtered or notaper:
        # transform lengths follow the spans the probes have grown to, and the engine that made the references has
        # already seen the "true" source: compare with a FRESH oracle engine, as the device evaluates
        e.close()
        e = _Recorded(sc.oracle())
        sc.apply_setup(e, True)
        e.set_misfit_method(mid)
    per_source = spectral or filtered or notaper       # every trial source against an oracle engine of its own
    if filtered:                                               # cosine frequency filter (comparator.f90:1186-1263)
        f0 = rng.uniform(0.01, 0.05)
        fx, fy = [f0, 2 * f0, 6 * f0, 9 * f0], [0., 1., 1., 0.]
        for ir in range(nrec):
            e.set_filter(ir + 1, fx, fy)
            p.set_misfit_filter(ir + 1, fx, fy)
    e.set_misfit_method(mid)
    p.set_misfit_method(method)
    if mid >= 7:
        for ir in range(nrec):
            lo, hi = sorted(int(v) for v in rng.integers(-5, 6, 2))
            e.set_floating_shiftrange(ir + 1, lo, hi)
            p.set_floating_shiftrange(ir + 1, lo * dt, hi * dt)
    if rng.random() < 0.3:
        xus, zus = int(rng.integers(1, 3)), int(rng.integers(1, 3))
        e.set_interpolation(bil, xus, zus)
        p.set_spacial_undersampling(xus, zus)
    f = float(rng.choice([1.0, 1.0, 0.7]))
    e.set_synthetics_factor(f)
    p.set_synthetics_factor(f)
    stype = int(rng.choice([1, 2, 3, 4, 5, 6]))
    n = int(rng.integers(1, 9))
    if per_source:
        n = min(n, 4)    # transform lengths and un-tapered spans follow the probes' history in the reference: a fresh engine per source
    base = np.array(synthetic.TRUE_BILAT, np.float32)
    if stype == 1:
        tr = np.tile(base, (n, 1))
        tr[:, 5] += rng.uniform(-20, 20, n); tr[:, 6] -= rng.uniform(0, 20, n); tr[:, 3] += rng.uniform(-1500, 1500, n)
        tr[:, 0] += rng.uniform(-2, 2, n); tr[:, 13] = rng.choice([0., 0.3, 2.], n)
        if rng.random() < 0.3:
            tr[:, 9:12] = 0.0                                  # point source
    elif stype == 2:
        tr = np.tile(np.array([0, 0, 0, 10000, 5e19, 80, 70, 100, 3000, 3000, 1.5], np.float32), (n, 1))
        tr[:, 5] += rng.uniform(-20, 20, n); tr[:, 8] = rng.uniform(500, 4000, n)
    elif stype == 3:
        tr = np.tile(np.array([0, 0, 0, 10000, 7e18, 1, 0, -1, 1, 1, 1, 20, 10], np.float32), (n, 1))
        tr[:, 5:11] += rng.standard_normal((n, 6)).astype(np.float32) * 0.3; tr[:, 11] = rng.uniform(3, 25, n)
    else:
        same = rng.random() < 0.6                              # runs of geometry-identical sources
        tr = np.tile(np.array([0, 0, 0, 10000, 1, 0, 0, 0, 0, 0, 1.0], np.float32), (n, 1))
        tr[:, 4:10] = rng.standard_normal((n, 6)) * 1e18
        if not same:
            tr[:, 3] += rng.uniform(-1500, 1500, n); tr[:, 10] = rng.choice([0.4, 1.0, 2.6], n)
    name = {1: "bilateral", 2: "circular", 3: "point_lp", 4: "eikonal", 5: "mt_eikonal", 6: "moment_tensor"}[stype]
    if stype in (4, 5):
        from oracle import ko
        nz = sc.gf["data"].shape[1]
        cp = np.array([[0, 0, 6500.], [0, 0, 6000. + (nz - 1) * 2000. - 500.]], np.float32)
        cn = np.array([[0, 0, -1.], [0, 0, 1.]], np.float32)
        n = min(n, 3)
        tr = []
        for i in range(n):
            common = [rng.uniform(-1, 1), rng.uniform(-500, 500), rng.uniform(-500, 500), rng.uniform(8000, 11000)]
            bord = [rng.uniform(-300, 300), rng.uniform(-300, 300), rng.uniform(1500, 4000)]
            nukl = [rng.uniform(-400, 400), rng.uniform(-300, 300)]
            if stype == 5:
                tr.append(common + [1.0, rng.uniform(0, 360), rng.uniform(40, 90)] + bord + nukl + [rng.uniform(0.7, 1.0)]
                          + list(rng.standard_normal(6) * 1e18) + [float(rng.choice([0., 1.2]))])
            else:
                tr.append(common + [5e18, rng.uniform(0, 360), rng.uniform(40, 90), rng.uniform(-180, 180)] + bord + nukl
                          + [rng.uniform(0.7, 1.0), float(rng.choice([0., 1.2]))])
        tr = np.array(tr, np.float32)
        prof = synthetic.SYNTH_CRUST
        oprof = ko.crust_profile(prof[0:8], prof[8:16], prof[16:24], prof[24:31])
        p.set_source_crust(prof, prof)
        p.set_source_constraints(cp, cn)
        ms, ns, gs, scales_by_src = [], [], [], []
        for t in tr:
            c, mo, ri, _ = ko.discretize_eikonal(stype, t, edt, oprof, cp, cn)
            fe = e.fresh(sc) if per_source else e
            fe.set_centroids(c, mo, ri)
            a, b, cglob = fe.get_misfits()
            ms.append(a); ns.append(b); gs.append(cglob)
            if per_source:
                scales_by_src.append(slot_scales(fe, comps, dt))
                fe.close()
        m, nn, g = np.array(ms), np.array(ns), np.array(gs, np.float32)
    else:
        # a fifth of the multi-source cases: some trials repeated with another moment (bilateral / circular / point_lp:
        # parameter 5) -- identical centroid tables, synthesised once and re-scaled (KIWI_HIP_DEDUPE=2 includes point sources)
        if n > 1 and stype in (1, 2, 3) and rng.random() < 0.2:
            k = int(rng.integers(1, n))
            extra = tr[rng.integers(0, n, k)].copy()
            extra[:, 4] *= rng.choice([0.5, 2.0, 3.7], k).astype(np.float32)
            tr = np.concatenate([tr, extra])[rng.permutation(n + k)]
            n = len(tr)
        if per_source:
            ms, ns, gs, scales_by_src = [], [], [], []
            for t in tr:
                fe = e.fresh(sc)
                a, b, cglob = oracle_misfits(fe, stype, t[None, :])
                ms.append(a[0]); ns.append(b[0]); gs.append(cglob[0])
                scales_by_src.append(slot_scales(fe, comps, dt))
                fe.close()
            m, nn, g = np.array(ms), np.array(ns), np.array(gs, np.float32)
        else:
            m, nn, g = oracle_misfits(e, stype, tr)
    p.set_source_params(name, tr)
    p.eval()
    pm, pn, pg = p.get_misfits()
    if spectral or filtered:
        # fp32 transforms against the oracle's fp64 DFT: MISFIT_RTOL of max(norm factor, misfit) plus the round-off an fp32
        # transform of that length explains (tests/common.py fft_roundoff_bound: grows with log2 N, with the window length for an
        # L1 sum and with what the frequency filter rejects); slots without a frequency filter under a time-domain norm are
        # compared on the plain tapered arrays and get no round-off term (they pass through no transform)
        scale = np.maximum(np.abs(m), np.maximum(nn, 1e-30))
        ok, ratio = True, 0.0
        for i in range(len(tr)):
            scales = scales_by_src[i]
            ok_i, ratio_i = spectral_close(method, dt, pm[i], m[i], nn[i], scales, pn[i])
            ok, ratio = ok and ok_i, max(ratio, ratio_i)
        bad = np.zeros_like(pm, bool)
        if os.environ.get("KIWI_FUZZ_STATS"):
            print("FFTSTAT %s filtered=%d L=%d ntrans=%d ratio=%.3f" % (method, int(filtered), L, int(scales[0].max()), ratio))
    else:
        # (fused arithmetic contract: relative to max(misfit, norm factor), tests/common.py misfit_close)
        scale = np.maximum(np.abs(m), (1.0 if arith() == "fused" else 1e-6) * np.maximum(nn, 1e-30))
        tol = 1e-6 if mid not in (5,) else 2e-6
        if arith() == "fused" and mid == 6:
            # `peak` is ONE sample of the difference trace: no sum averages its round-off, the bound is the synthetics' own
            # (2e-6 of the trace maximum, tests/common.py SYN_RTOL; seen: 1.1e-6 of the norm factor, case 5303 / 6701)
            tol = 2e-6
        bad = np.abs(pm - m) > tol * scale
        ok = np.array_equal(pn, nn) and not bad.any()
    # the whole list through the overlapped one-call in random pieces: the same bits as the calls above
    if ok and n > 1 and rng.random() < 0.5:
        piece = int(rng.integers(1, n + 1))
        qm, qn, qg, qs = p.misfits_for_params(name, tr, piece)
        if not (np.array_equal(qm, pm) and np.array_equal(qn, pn) and np.array_equal(qg, pg) and not qs.any()):
            ok = False
            print("BAD misfits_for_params in pieces of %d differs from set_source_params + get_misfits" % piece)
            if os.environ.get("KIWI_HIP_DEBUG"):
                print("  misfits differ at", np.argwhere(qm != pm)[:6].tolist(), "norms at", np.argwhere(qn != pn)[:6].tolist(), "globals at", np.argwhere(qg != pg)[:6].tolist(), "status", qs)
                print("  batch", pm[qm != pm][:4], "pieces", qm[qm != pm][:4], "norm batch", pn[qn != pn][:4], "pieces", qn[qn != pn][:4])
    if not ok and os.environ.get("KIWI_HIP_DEBUG"):
        import ctypes as C
        from oracle import ko as _ko
        L_ = _ko.lib()
        L_.ko_engine_probe_spans.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        for ir in range(nrec):
            for k in range(len(comps[ir])):
                a, b = (C.c_int * 4)(), (C.c_int * 4)()
                L_.ko_engine_probe_spans(e.h, ir + 1, k + 1, 0, a)
                L_.ko_engine_probe_spans(e.h, ir + 1, k + 1, 1, b)
                print("oracle rec %d comp %d: ref span %s data %s | syn span %s data %s -> ntrans %d" %
                      (ir + 1, k + 1, list(a[:2]), list(a[2:]), list(b[:2]), list(b[2:]), a[1] - a[0] + 1))
        print("gpu", pm[0], "oracle", m[0], "norm gpu", pn[0], "oracle", nn[0], "rel norm diff", np.abs(pn[0] - nn[0]) / nn[0])
        for ir in range(nrec):
            for k in range(len(comps[ir])):
                lo_o, do = e.synthetic(ir + 1, k + 1, 1)
                lo_p, dp = p.get_synthetics(0, ir + 1, k + 1, 1)
                nz = np.nonzero(dp)[0]
                a0, b0 = max(lo_o, lo_p), min(lo_o + len(do), lo_p + len(dp))
                dd = np.abs(do[a0 - lo_o:b0 - lo_o] - dp[a0 - lo_p:b0 - lo_p]) if b0 > a0 else np.zeros(1)
                print("synthetic rec %d comp %d: oracle [%d, %d] (first %g last %g) | device window [%d, %d] nonzero [%s] | max diff on overlap %.3g, device behind oracle's end: %s"
                      % (ir + 1, k + 1, lo_o, lo_o + len(do) - 1, do[0] if len(do) else 0, do[-1] if len(do) else 0, lo_p, lo_p + len(dp) - 1,
                         ("%d, %d" % (lo_p + nz[0], lo_p + nz[-1])) if len(nz) else "none", float(dd.max()),
                         dp[lo_o + len(do) - lo_p:lo_o + len(do) - lo_p + 3] if lo_o + len(do) - lo_p < len(dp) else []))
    if verbose or not ok:
        print("%s ng=%d L=%d nrec=%d comps=%s bil=%d %s edt=%.1f %s x%d method=%s%s factor=%.1f -> worst %.2e (median misfit / norm %.3f)"
              % ("ok " if ok else "BAD", ng, L, nrec, comps, bil, variant, edt, name, n, method, ("+filter" if filtered else "") + ("+notaper" if notaper else ""), f,
                 float(np.max(np.abs(pm - m) / scale)), float(np.median(np.abs(m) / np.maximum(nn, 1e-30)))))
    p.close()
    e.close()
    sc.odb.close()
    return ok


def main():
    """usage: fuzz_gpu_parity.py [seconds] [seed]  |  case <seed> <index> (replays one case)  |  range <seed> <first> <end>"""
    if len(sys.argv) > 1 and sys.argv[1] == "case":
        seed, idx = int(sys.argv[2]), int(sys.argv[3])
        ok = one_case(np.random.default_rng([seed, idx]), verbose=True)
        sys.exit(0 if ok else 1)
    if len(sys.argv) > 1 and sys.argv[1] == "range":          # range <seed> <first> <last+1>: those cases, quiet unless bad
        seed, n0, n1 = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        bad = [n for n in range(n0, n1) if not one_case(np.random.default_rng([seed, n]), verbose=False)]
        print("fuzz range: %d cases, bad: %s, seed %d" % (n1 - n0, bad, seed))
        sys.exit(1 if bad else 0)
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261002
    t0 = time.time()
    n = nbad = 0
    while time.time() - t0 < seconds:
        ok = one_case(np.random.default_rng([seed, n]), verbose=(n < 3))       # every case replayable: `case <seed> <n>`
        if not ok:
            print("   ^ replay with: python tests/fuzz_gpu_parity.py case %d %d" % (seed, n))
            nbad += 1
        n += 1
    print("fuzz: %d cases, %d bad, seed %d" % (n, nbad, seed))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
