"""Shared scenario builders: the same deterministic setup fed to the CPU oracle (oracle/ko.py)
and to the product (kiwi_amd.Engine)."""
import numpy as np

import os

from kiwi_amd import synthetic
from oracle import ko

# Optional toolchains of the build image.  Tests that depend on them do not skip silently: tests/test_product_cpu.py
# asserts that this image (and the GPU box, which runs the same image) has them, so a branch that did not run shows
# up as a failure, not as a quietly shorter test.  KIWI_TEST_ALLOW_MISSING=1 lifts that on a bare machine.
HAVE_HDF5 = os.path.exists("/opt/conda/include/hdf5.h")
HAVE_FLANG = os.path.exists("/opt/rocm/bin/amdflang")


# ---- the ONE statement of the spectral / filtered tolerances.  The oracle transforms in fp64 (an exact DFT
# rounded once; FFTW's own fp32 rounding is "parity unpinned"), the device in fp32 (in-LDS radix-4 or hipFFT).  What an
# fp32 transform of N points does to its output is known: an error vector of 2-norm <= c eps log2(N) ||input||_2 (eps = 2^-24,
# c a small constant; Higham, Accuracy and Stability of Numerical Algorithms, section 24.1) -- relative to the transform's INPUT,
# the tapered trace, whatever a frequency filter leaves of it afterwards.  Pushed through the four norms (a, b: reference and
# synthetic of a slot, n2(x) = sqrt(dt sum x^2) of the TAPERED, UNFILTERED trace, W the window length in samples):
#   l2norm on filtered traces   |dm| <= sqrt(dt) ||da - db||_2                 <= c eps log2N  (n2(a) + n2(b))
#   l1norm on filtered traces   |dm| <= dt sum |da_i - db_i| <= dt sqrt(W) ||.||_2 <= c eps log2N  sqrt(W dt) (n2(a) + n2(b))
#   ampspec_l2norm              |dm| <= sqrt(df) ||dA - dB||_2, ||X||_2 <= sqrt(N) ||x||_2  <= c eps log2N  (n2(a) + n2(b)) / dt
#   ampspec_l1norm              |dm| <= df sqrt(N / 2 + 1) ||dA - dB||_2              <= c eps log2N  (n2(a) + n2(b)) / dt^1.5
# so the bound GROWS with the window length for an L1 sum and with what the filter rejects (n2 is the unfiltered trace): the
# cases that used to set fixed figures (5e-4 ... 1e-3 of a filtered-L1 norm factor at 2300 samples) are inside it
# with the same c as a 200-sample window.  Everything that is NOT transform round-off -- window placement, taper, fold, the sums --
# has to agree to MISFIT_RTOL of max(norm factor, misfit) on top: a one-sample window error on a quiet trace is orders of
# magnitude above both terms.  No fixed tolerance is left for these norms.


# c: the derivation's worst case has every rounding error of a transform pulling the same way (c of the order of 1 ... 5); what
# is seen is a random walk -- largest |error| / bound(c = 1): 0.17 for the oracle's textbook fp32 FFT against its fp64 one
# (tests/test_oracle_fft32.py, ampspec_l2norm at 256 points), 0.006 for the device beyond its 1e-6 (3000 randomised cases, in-LDS
# radix-4 and hipFFT alike, round 4)
FFT_ROUNDOFF_C = 0.25


def fft_roundoff_bound(method, dt, ntrans, wlen, n2_ref, n2_syn, c=None):
    """absolute bound on |device misfit - oracle misfit| of one slot that fp32 transforms of `ntrans` points explain"""
    c = FFT_ROUNDOFF_C if c is None else c
    base = c * 2.0 ** -24 * np.log2(np.maximum(ntrans, 2)) * (np.asarray(n2_ref, np.float64) + np.asarray(n2_syn, np.float64))
    if method == "l2norm":
        return base
    if method == "l1norm":
        return base * np.sqrt(np.asarray(wlen, np.float64) * dt)
    if method == "ampspec_l2norm":
        return base / dt
    if method == "ampspec_l1norm":
        return base / dt ** 1.5
    if method == "peak":                      # max |da_i|, |db_i| <= ||.||_2
        return base / np.sqrt(dt)
    raise ValueError(method)


def slot_scales(e, comps, dt):
    """per misfit slot of an oracle engine that has just evaluated a source: (ntrans, window length, n2(reference), n2(synthetic))
    of the tapered, unfiltered probes -- what fft_roundoff_bound is made of.  Enabled receivers, receiver-major."""
    nt, wl, na, nb = [], [], [], []
    for ir, cs in enumerate(comps):
        for k in range(len(cs)):
            _, a = e.reference(ir + 1, k + 1, 2)
            _, b = e.synthetic(ir + 1, k + 1, 2)
            _, amps = e.amp_spectrum(ir + 1, k + 1, True, False)
            nt.append(max(2 * (len(amps) - 1), 2)); wl.append(max(len(a), len(b), 1))
            na.append(np.sqrt(dt * np.sum(np.asarray(a, np.float64) ** 2))); nb.append(np.sqrt(dt * np.sum(np.asarray(b, np.float64) ** 2)))
    return np.array(nt), np.array(wl), np.array(na), np.array(nb)


def spectral_close(method, dt, pm, m, n, scales, pn=None):
    """device misfits pm (and norm factors pn) of ONE source against the oracle's m, n under a spectral norm or a frequency
    filter: MISFIT_RTOL of max(norm factor, misfit) plus what fp32 transforms explain (fft_roundoff_bound).  Returns
    (ok, worst ratio of the excess over the round-off bound at c = 1) -- the ratio is what FFT_ROUNDOFF_C was calibrated on."""
    nt, wl, na, nb = scales
    pm, m, n = (np.asarray(x, np.float64) for x in (pm, m, n))
    scale = np.maximum(np.abs(n), np.abs(m))
    b1 = fft_roundoff_bound(method, dt, nt, wl, na, nb, c=1.0)
    excess = np.abs(pm - m) - MISFIT_RTOL * scale
    ratio = float(np.max(excess / np.maximum(b1, 1e-300)))
    ok = bool(np.all(excess <= FFT_ROUNDOFF_C * b1))
    if pn is not None:                       # norm factors of the references: the same transform on the reference alone
        pn = np.asarray(pn, np.float64)
        bn = fft_roundoff_bound(method, dt, nt, wl, na, 0.0 * na, c=1.0)
        exn = np.abs(pn - n) - MISFIT_RTOL * np.abs(n)
        ratio = max(ratio, float(np.max(exn / np.maximum(bn, 1e-300))))
        ok = ok and bool(np.all(exn <= FFT_ROUNDOFF_C * bn))
    return ok, ratio


# ---- the two arithmetic contracts of the accumulate kernels (include/kiwi_hip.h; tests/conftest.py runs every GPU test under both)
MISFIT_RTOL = 1e-6        # BASELINE.json north_star: misfits within 1e-6 relative
SYN_RTOL = 2e-6           # synthetics: |device - oracle| <= SYN_RTOL * max|oracle| per trace


def arith():
    return os.environ.get("KIWI_HIP_ARITH", "exact")


def misfit_close(a, b, norm=None, glob=False):
    """Device misfits a against oracle misfits b.  exact: |a - b| <= 1e-6 |b| per value.  fused: a misfit is the norm of a
    DIFFERENCE of traces, its round-off scales with the traces (the norm factor), not with itself -- a trial next to the true
    source has a misfit far below its norm factor --: |a - b| <= 1e-6 max(|b|, norm factor) per slot, `norm` = the slots' norm
    factors (required: there is no other floor).  glob=True: a, b are GLOBAL misfits g = |m| / |n| (minimizer_engine.f90:936-942),
    normalised by construction -- the same rule with norm factor 1: from |dm_i| <= 1e-6 max(m_i, n_i) follows
    |dg| <= 1e-6 sqrt(g^2 + 1)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = np.maximum(np.abs(b), 1e-30)
    if arith() == "fused":
        if glob:
            scale = np.sqrt(b * b + 1.0)
        else:
            assert norm is not None, "fused contract: per-slot misfits are compared on the scale of their norm factors"
            scale = np.maximum(scale, np.abs(np.asarray(norm, np.float64)))
    return bool(np.all(np.abs(a - b) <= MISFIT_RTOL * scale))


def same_bits(a, b):
    """Two evaluations of the same trial by different kernels (or batch shapes).  exact: the same bits.  fused: each kernel
    instantiation contracts its multiply-add pairs on its own (a variant of the same template is not bound to fuse the same
    pairs): within SYN_RTOL of the larger array's maximum."""
    a = np.asarray(a)
    b = np.asarray(b)
    if arith() == "exact":
        return a.shape == b.shape and a.tobytes() == b.tobytes()
    if a.shape != b.shape:
        return False
    top = max(float(np.max(np.abs(a), initial=0.0)), float(np.max(np.abs(b), initial=0.0)))
    return bool(np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= SYN_RTOL * top))


class Scenario:
    """Small synthetic inversion setup (scaled down)."""

    def __init__(self, nx=10, nz=5, ng=10, L=256, nrec=6, comps="ned", variant="probe", bilinear=True,
                 dmin=104e3, dspan=24e3, effective_dt=0.5, true_params=None, true_type=1, depths=None,
                 taper_ramp=10.0, comps_list=None):
        self.gf = synthetic.make_gfdb(nx=nx, nz=nz, ng=ng, L=L, variant=variant)
        self.lat, self.lon, self.depth, self.comps, self.dist = synthetic.make_receivers(
            nrec, dmin=dmin, dspan=dspan, comps=comps)
        if comps_list is not None:
            self.comps = list(comps_list)
        if depths is not None:
            self.depth = np.asarray(depths, np.float32)
        self.nrec = nrec
        self.bilinear = bilinear
        self.effective_dt = effective_dt
        self.true_type = true_type
        self.true_params = np.array(synthetic.TRUE_BILAT if true_params is None else true_params, np.float32)
        self.taper_ramp = taper_ramp
        self.refs = {}
        self.tapers = {}

    # ---------------------------------------------------------------- oracle side
    def oracle(self, nthreads=1):
        g = self.gf
        nx, nz, ng, L = g["data"].shape
        db = ko.Gfdb(nx, nz, ng, g["dt"], g["dx"], g["dz"], g["firstx"], g["firstz"])
        for ix in range(nx):
            for iz in range(nz):
                for ig in range(ng):
                    n = int(g["nsamp"][ix, iz, ig])
                    if n > 0:
                        db.set_trace(ix + 1, iz + 1, ig + 1, int(g["first"][ix, iz, ig]), g["data"][ix, iz, ig, :n])
        e = ko.Engine(db)
        e.set_receivers(self.lat, self.lon, self.depth, self.comps)
        e.set_source_location(40.0, 30.0, 0.0)
        e.set_effective_dt(self.effective_dt)
        e.set_interpolation(self.bilinear)
        e.set_nthreads(nthreads)
        self.odb = db
        return e

    def make_references(self, e):
        """Reference traces = oracle synthetics of the 'true' source; tapers over the whole trace."""
        e.set_source_params(self.true_type, self.true_params)
        e.calculate_seismograms()
        e.scale_seismograms()
        dt = self.gf["dt"]
        for ir in range(self.nrec):
            for k in range(len(self.comps[ir])):
                lo, d = e.synthetic(ir + 1, k + 1, 1)
                self.refs[(ir + 1, k + 1)] = (lo, d)
            if len(self.comps[ir]):
                lo, d = self.refs[(ir + 1, 1)]
                self.tapers[ir + 1] = synthetic.full_taper(lo, len(d), dt, self.taper_ramp)

    def apply_setup(self, eng, is_oracle):
        for (ir, k), (lo, d) in self.refs.items():
            if is_oracle:
                eng.set_reference(ir, k, lo, d)
            else:
                eng.set_ref_seismogram(ir, k, lo, d)
        for ir, (x, y) in self.tapers.items():
            if is_oracle:
                eng.set_taper(ir, x, y)
            else:
                eng.set_misfit_taper(ir, x, y)

    # ---------------------------------------------------------------- product side
    def product(self, device=0):
        from kiwi_amd import Engine
        g = self.gf
        first, nsamp, data = self.odb.dense_tables()      # spans exactly as the packed DB stores them
        p = Engine(device)
        p.set_database(g["dt"], g["dx"], g["dz"], g["firstx"], g["firstz"], data, first, nsamp)
        p.set_receivers(self.lat, self.lon, self.depth, self.comps)
        p.set_source_location(40.0, 30.0, 0.0)
        p.set_effective_dt(self.effective_dt)
        p.set_local_interpolation("bilinear" if self.bilinear else "nearest")
        return p


def oracle_misfits(e, sourcetype, params):
    """Loop of set_source_params + get_misfits (seismosizer.py:703-718) on the oracle."""
    ms, ns, gs = [], [], []
    for p in np.atleast_2d(params):
        e.set_source_params(sourcetype, p)
        m, n, g = e.get_misfits()
        ms.append(m)
        ns.append(n)
        gs.append(g)
    return np.array(ms), np.array(ns), np.array(gs, np.float32)
