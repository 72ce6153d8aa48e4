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


# ---- the ONE statement of the spectral / filtered tolerances: per misfit slot, the difference between
# device and oracle relative to max(norm factor, |misfit|) of the slot -- an amplitude-spectrum misfit is a difference of
# nearly equal spectra, its round-off scales with the spectra, not with itself.  The oracle transforms in fp64 (FFTW's own
# rounding is "parity unpinned"), the device in fp32 (in-LDS radix-4 or hipFFT); an L1 sum adds the transforms'
# round-off linearly over the window.  Used by the full-size tests, the randomised sweep and its pytest slice alike.
SPECTRAL_TOL = {("ampspec_l2norm", False): 2e-5, ("ampspec_l2norm", True): 2e-5,
                ("ampspec_l1norm", False): 5e-5, ("ampspec_l1norm", True): 5e-5,
                ("l2norm", True): 3e-5,           # time-domain L2 on frequency-filtered traces (forward, filter, back)
                ("l1norm", True): 1e-3}           # ... L1 there: grows with the crest factor of the trace
SPECTRAL_NORM_TOL = 5e-5                          # norm factors of FILTERED references, on the scale of >= 1/20 of the case's largest


def spectral_tol(method, filtered):
    return SPECTRAL_TOL[(method, bool(filtered))]


# ---- the two arithmetic contracts of the accumulate kernels (include/kiwi_hip.h; tests/conftest.py runs every GPU test under both)
MISFIT_RTOL = 1e-6        # BASELINE.json north_star: misfits within 1e-6 relative
SYN_RTOL = 2e-6           # synthetics: |device - oracle| <= SYN_RTOL * max|oracle| per trace


def arith():
    return os.environ.get("KIWI_HIP_ARITH", "exact")


def misfit_close(a, b, norm=None):
    """Device misfits a against oracle misfits b.  exact: |a - b| <= 1e-6 |b| per value.  fused: a misfit is the norm of a
    DIFFERENCE of traces, its round-off scales with the traces (the norm factor), not with itself -- a trial next to the true
    source has a misfit far below its norm factor --: |a - b| <= 1e-6 max(|b|, norm factor); without norm factors at hand
    (global misfits, which are normalised already) the floor is a twentieth of the batch's largest value."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = np.maximum(np.abs(b), 1e-30)
    if arith() == "fused":
        floor = np.abs(np.asarray(norm, np.float64)) if norm is not None else 0.05 * np.max(np.abs(b), initial=0.0)
        scale = np.maximum(scale, floor)
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
