"""ctypes loader for the CPU oracle (oracle/libko.so) and the reference build (oracle/_ref).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by kiwi_amd (the product).
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_double_p = C.POINTER(C.c_double)


def build(ref=True):
    """Compile libko.so (gcc) and, when /root/reference exists, _ref/libkiwi_ref.so (amdflang)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "libko.so"])
    if ref and os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


class Strip(C.Structure):
    _fields_ = [("d", c_float_p), ("lo", C.c_int), ("n", C.c_int)]


class Trace(C.Structure):
    _fields_ = [("nstrips", C.c_int), ("span", C.c_int * 2), ("strips", C.POINTER(Strip))]


class Plf(C.Structure):
    _fields_ = [("n", C.c_int), ("x", C.c_float * 64), ("y", C.c_float * 64)]


class Centroid(C.Structure):
    _fields_ = [("north", C.c_float), ("east", C.c_float), ("depth", C.c_float), ("time", C.c_float),
                ("m", C.c_float * 6)]


class Psm(C.Structure):
    _fields_ = [("sourcetype", C.c_int), ("nparams", C.c_int), ("params", C.c_float * 24),
                ("moment", C.c_float), ("risetime", C.c_float),
                ("rotmat_rup", (C.c_float * 3) * 3), ("rotmat_slip", (C.c_float * 3) * 3),
                ("grid_size", C.c_int * 3), ("inited", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libko.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.ko_d2r_d.restype = C.c_double
        L.ko_d2r_d.argtypes = [C.c_double]
        L.ko_d2r_r.restype = C.c_float
        L.ko_d2r_r.argtypes = [C.c_float]
        L.ko_distance_accurate50m.restype = C.c_double
        L.ko_plf_integrate.restype = C.c_float
        L.ko_gfdb_create.restype = C.c_void_p
        L.ko_gfdb_create.argtypes = [C.c_int] * 3 + [C.c_float] * 5
        L.ko_gfdb_destroy.argtypes = [C.c_void_p]
        L.ko_gfdb_set_trace_dense.argtypes = [C.c_void_p] + [C.c_int] * 5 + [c_float_p]
        L.ko_gfdb_trace_span.argtypes = [C.c_void_p] + [C.c_int] * 3 + [c_int_p]
        L.ko_gfdb_trace_unpack.argtypes = [C.c_void_p] + [C.c_int] * 3 + [c_float_p]
        L.ko_gfdb_get_indices.argtypes = [C.c_void_p, C.c_float, C.c_float, c_int_p, c_int_p]
        L.ko_gfdb_get_indices_bilin.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int,
                                                c_int_p, c_int_p, c_float_p, c_float_p]
        L.ko_engine_create.restype = C.c_void_p
        L.ko_engine_create.argtypes = [C.c_void_p]
        L.ko_engine_destroy.argtypes = [C.c_void_p]
        L.ko_engine_set_receivers.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_float_p,
                                              C.POINTER(C.c_char_p)]
        L.ko_engine_switch_receiver.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ko_engine_set_source_location.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_double]
        L.ko_engine_set_effective_dt.argtypes = [C.c_void_p, C.c_float]
        L.ko_engine_set_interpolation.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ko_engine_set_source_params.argtypes = [C.c_void_p, C.c_int, c_float_p]
        L.ko_engine_set_centroids.argtypes = [C.c_void_p, C.c_int, c_float_p, C.c_float, C.c_float]
        L.ko_engine_set_reference.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p]
        L.ko_engine_set_taper.argtypes = [C.c_void_p, C.c_int, C.c_int, c_float_p, c_float_p]
        L.ko_engine_set_filter.argtypes = [C.c_void_p, C.c_int, C.c_int, c_float_p, c_float_p]
        L.ko_engine_set_misfit_method.argtypes = [C.c_void_p, C.c_int]
        L.ko_engine_set_synthetics_factor.argtypes = [C.c_void_p, C.c_float]
        L.ko_engine_set_floating_shiftrange.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ko_engine_shift_ref_seismogram.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ko_engine_autoshift_ref_seismogram.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ko_engine_autoshift_ref_seismogram.restype = C.c_int
        L.ko_engine_get_floating_shift.argtypes = [C.c_void_p, C.c_int]
        L.ko_engine_get_floating_shift.restype = C.c_int
        L.ko_engine_set_nthreads.argtypes = [C.c_void_p, C.c_int]
        L.ko_engine_amp_spectrum.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, C.c_int]
        L.ko_engine_amp_spectrum.restype = C.c_int
        L.ko_engine_cross_correlations.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_float_p]
        L.ko_engine_cross_correlations.restype = C.c_int
        L.ko_engine_shake.argtypes = [C.c_void_p, C.c_int, c_float_p]
        L.ko_engine_shake.restype = C.c_int
        for f in ("calculate_seismograms", "scale_seismograms", "calculate_misfits"):
            getattr(L, "ko_engine_" + f).argtypes = [C.c_void_p]
        L.ko_engine_get_misfits.argtypes = [C.c_void_p, c_float_p, c_float_p, C.c_int]
        L.ko_engine_get_global_misfit.restype = C.c_float
        L.ko_engine_get_global_misfit.argtypes = [C.c_void_p]
        L.ko_engine_get_displacement.argtypes = [C.c_void_p, C.c_int, C.c_int, c_int_p, c_float_p, C.c_int]
        L.ko_engine_get_synthetic.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_int_p, c_float_p, C.c_int]
        L.ko_engine_centroid_geometry.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ko_engine_receiver_geometry.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p]
        L.ko_probes_norm.restype = C.c_float
        L.ko_probe_norm.restype = C.c_float
        _lib = L
    return _lib


_ref = None


def ref_available():
    """oracle/_ref/libkiwi_ref.so exists (nothing is loaded: collection of a test module must not map it)"""
    return os.path.exists(os.path.join(HERE, "_ref", "libkiwi_ref.so"))


class LazyRef:
    """Stands for ref() in a test module: the library is loaded by the first test that calls into it, never at import --
    a `pytest -m gpu` process collects these modules and deselects their tests, and should not map the checker."""

    def __getattr__(self, name):
        return getattr(ref(), name)


def ref():
    """The reference's own Fortran modules (oracle/_ref/libkiwi_ref.so) or None if not built."""
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libkiwi_ref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_d2r_d.restype = C.c_double
        R.ref_d2r_d.argtypes = [C.c_double]
        R.ref_d2r_r.restype = C.c_float
        R.ref_d2r_r.argtypes = [C.c_float]
        _ref = R
    return _ref


# ---------------------------------------------------------------- small functional wrappers

def strip_from(lo, data):
    data = np.ascontiguousarray(data, np.float32)
    s = Strip()
    lib().ko_strip_init(C.byref(s), C.c_int(lo), C.c_int(lo + len(data) - 1), _fp(data))
    return s


def strip_to_np(s):
    if s.n == 0:
        return s.lo, np.zeros(0, np.float32)
    return s.lo, np.ctypeslib.as_array(s.d, (s.n,)).copy()


def multiply_add(tlo, tdata, strip, factor=1.0, mode=0, ishift=0, rshift=0.0):
    """trace_pack(tdata) then trace_multiply_add onto strip (None = unallocated). Returns (lo, data)."""
    L = lib()
    ts = strip_from(tlo, tdata)
    t = Trace()
    L.ko_trace_pack(C.byref(ts), C.byref(t))
    s = Strip() if strip is None else strip_from(*strip)
    L.ko_trace_multiply_add(C.byref(t), C.byref(s), C.c_float(factor), C.c_int(mode), C.c_int(ishift),
                            C.c_float(rshift))
    out = strip_to_np(s)
    L.ko_trace_destroy(C.byref(t))
    L.ko_strip_destroy(C.byref(ts))
    L.ko_strip_destroy(C.byref(s))
    return out


def trace_pack_spans(lo, data):
    L = lib()
    ts = strip_from(lo, data)
    t = Trace()
    L.ko_trace_pack(C.byref(ts), C.byref(t))
    spans = [(t.strips[i].lo, t.strips[i].lo + t.strips[i].n - 1) for i in range(t.nstrips)]
    tspan = (t.span[0], t.span[1])
    L.ko_trace_destroy(C.byref(t))
    L.ko_strip_destroy(C.byref(ts))
    return spans, tspan


def strip_dataspan(lo, data):
    s = strip_from(lo, data)
    out = (C.c_int * 2)()
    lib().ko_strip_dataspan(C.byref(s), out)
    lib().ko_strip_destroy(C.byref(s))
    return out[0], out[1]


def strip_fold(lo, data, shifts, amps):
    s = strip_from(lo, data)
    shifts = np.ascontiguousarray(shifts, np.float32)
    amps = np.ascontiguousarray(amps, np.float32)
    lib().ko_strip_fold(C.byref(s), C.c_int(len(shifts)), _fp(shifts), _fp(amps))
    out = strip_to_np(s)
    lib().ko_strip_destroy(C.byref(s))
    return out


def make_plf(x, y):
    p = Plf()
    p.n = len(x)
    for i, (a, b) in enumerate(zip(x, y)):
        p.x[i] = a
        p.y[i] = b
    return p


def discretize(sourcetype, params, effective_dt):
    """psm_set + psm_to_tdsm.  Returns (centroids[n,10] float32, moment, risetime, grid_size)."""
    L = lib()
    psm = Psm()
    params = np.ascontiguousarray(params, np.float32)
    if L.ko_psm_set(C.byref(psm), C.c_int(sourcetype), _fp(params)) < 0:
        raise ValueError("bad source type")
    out = C.POINTER(Centroid)()
    n = L.ko_psm_to_tdsm(C.byref(psm), C.c_float(effective_dt), C.byref(out))
    if n < 0:
        raise ValueError("discretisation failed")
    arr = np.ctypeslib.as_array(C.cast(out, c_float_p), (n, 10)).copy() if n else np.zeros((0, 10), np.float32)
    C.CDLL(None).free(out)
    return arr, float(psm.moment), float(psm.risetime), tuple(psm.grid_size)


def set_fft_precision(bits):
    """64 (default): the comparator's transforms are an exact DFT rounded once; 32: a textbook fp32 radix-2 FFT (second checker of the
    spectral tolerances).  Process-wide; engines evaluate their probes with the precision in force at that moment."""
    L = lib()
    L.ko_set_fft_precision.argtypes = [C.c_int]
    L.ko_set_fft_precision(int(bits))


class Gfdb:
    def __init__(self, nx, nz, ng, dt, dx, dz, firstx, firstz):
        self.nx, self.nz, self.ng = nx, nz, ng
        self.dt, self.dx, self.dz, self.firstx, self.firstz = dt, dx, dz, firstx, firstz
        self.h = lib().ko_gfdb_create(nx, nz, ng, dt, dx, dz, firstx, firstz)

    def set_trace(self, ix, iz, ig, lo, data):
        """1-based ix, iz, ig; dense samples at indices lo.. are trace_pack'ed."""
        data = np.ascontiguousarray(data, np.float32)
        lib().ko_gfdb_set_trace_dense(self.h, ix, iz, ig, lo, lo + len(data) - 1, _fp(data))

    def span(self, ix, iz, ig):
        s = (C.c_int * 2)()
        if not lib().ko_gfdb_trace_span(self.h, ix, iz, ig, s):
            return None
        return s[0], s[1]

    def unpack(self, ix, iz, ig):
        sp = self.span(ix, iz, ig)
        out = np.zeros(sp[1] - sp[0] + 1, np.float32)
        lib().ko_gfdb_trace_unpack(self.h, ix, iz, ig, _fp(out))
        return sp[0], out

    def dense_tables(self):
        """(first[nx,nz,ng], nsamp[nx,nz,ng], data[nx,nz,ng,Lmax]) of the PACKED traces -- what a
        GFDB reader hands the product (kiwi_hip_set_gfdb)."""
        first = np.zeros((self.nx, self.nz, self.ng), np.int32)
        ns = np.zeros((self.nx, self.nz, self.ng), np.int32)
        rows = {}
        for ix in range(self.nx):
            for iz in range(self.nz):
                for ig in range(self.ng):
                    sp = self.span(ix + 1, iz + 1, ig + 1)
                    if sp is None:
                        continue
                    lo, d = self.unpack(ix + 1, iz + 1, ig + 1)
                    first[ix, iz, ig] = lo
                    ns[ix, iz, ig] = len(d)
                    rows[(ix, iz, ig)] = d
        lmax = max(int(ns.max()), 1)
        data = np.zeros((self.nx, self.nz, self.ng, lmax), np.float32)
        for (ix, iz, ig), d in rows.items():
            data[ix, iz, ig, :len(d)] = d
        return first, ns, data

    def close(self):
        if self.h:
            lib().ko_gfdb_destroy(self.h)
            self.h = None


class Engine:
    """Mirror of the minimizer_engine state used by the hot path (oracle side)."""

    def __init__(self, db):
        self.db = db
        self.h = lib().ko_engine_create(db.h)
        self.ncomp = []

    def set_receivers(self, lat_deg, lon_deg, depth, comps):
        n = len(lat_deg)
        lat = np.ascontiguousarray(lat_deg, np.float64)
        lon = np.ascontiguousarray(lon_deg, np.float64)
        dep = np.ascontiguousarray(depth if depth is not None else np.zeros(n), np.float32)
        arr = (C.c_char_p * n)(*[c.encode() for c in comps])
        rc = lib().ko_engine_set_receivers(self.h, n, lat.ctypes.data_as(c_double_p), lon.ctypes.data_as(c_double_p),
                                           _fp(dep), arr)
        if rc != 0:
            raise ValueError("receiver_init failed")
        self.ncomp = [len(c) for c in comps]
        self.nrec = n

    def switch_receiver(self, irec1, state):
        lib().ko_engine_switch_receiver(self.h, irec1, int(state))

    def set_source_location(self, lat_deg, lon_deg, ref_time=0.0):
        lib().ko_engine_set_source_location(self.h, lat_deg, lon_deg, ref_time)

    def set_effective_dt(self, dt):
        lib().ko_engine_set_effective_dt(self.h, dt)

    def set_interpolation(self, bilinear, xus=1, zus=1):
        lib().ko_engine_set_interpolation(self.h, int(bilinear), xus, zus)

    def set_source_params(self, sourcetype, params):
        p = np.ascontiguousarray(params, np.float32)
        if lib().ko_engine_set_source_params(self.h, sourcetype, _fp(p)) != 0:
            raise ValueError("set_source_params failed")

    def set_centroids(self, cent, moment=1.0, risetime=0.0):
        c = np.ascontiguousarray(cent, np.float32)
        lib().ko_engine_set_centroids(self.h, len(c), _fp(c), moment, risetime)

    def set_reference(self, irec1, icomp1, first, data):
        d = np.ascontiguousarray(data, np.float32)
        lib().ko_engine_set_reference(self.h, irec1, icomp1, first, len(d), _fp(d))

    def set_taper(self, irec1, x, y):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        lib().ko_engine_set_taper(self.h, irec1, len(x), _fp(x), _fp(y))

    def set_filter(self, irec1, x, y):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        lib().ko_engine_set_filter(self.h, irec1, len(x), _fp(x), _fp(y))

    def set_misfit_method(self, method):
        lib().ko_engine_set_misfit_method(self.h, method)

    def set_synthetics_factor(self, f):
        lib().ko_engine_set_synthetics_factor(self.h, f)

    def set_floating_shiftrange(self, irec1, lo, hi):
        """shift range in SAMPLES (set_floating_shiftrange converts seconds with nint(shift/dt), minimizer_engine.f90:432)"""
        lib().ko_engine_set_floating_shiftrange(self.h, irec1, lo, hi)

    def shift_ref_seismogram(self, irec1, ishift):
        lib().ko_engine_shift_ref_seismogram(self.h, irec1, ishift)

    def autoshift_ref_seismogram(self, irec1, lo, hi):
        """needs current synthetics (get_misfits first); range and result in samples"""
        return lib().ko_engine_autoshift_ref_seismogram(self.h, irec1, lo, hi)

    def floating_shift(self, irec1):
        return lib().ko_engine_get_floating_shift(self.h, irec1)

    def set_nthreads(self, n):
        lib().ko_engine_set_nthreads(self.h, n)

    def amp_spectrum(self, irec1, icomp1, synthetic=True, filtered=False, maxn=1 << 20):
        """probe_get_amp_spectrum of a synthetic or reference probe as it stands: (df, amplitudes)."""
        out = np.zeros(maxn, np.float32)
        df = C.c_float()
        n = lib().ko_engine_amp_spectrum(self.h, irec1, icomp1, int(synthetic), int(filtered), C.byref(df), _fp(out), maxn)
        return float(df.value), out[:n].copy()

    def cross_correlations(self, irec1, lo, hi):
        """receiver_calculate_cross_correlations for integer shifts lo..hi: cc[ncomp, nshift] (update_misfits first)."""
        self.get_misfits()
        ns = hi - lo + 1
        out = np.zeros(5 * ns, np.float32)
        nc = lib().ko_engine_cross_correlations(self.h, irec1, lo, hi, _fp(out))
        return out[:nc * ns].reshape(nc, ns)

    def peak_amplitudes(self, differentiate):
        """get_peak_amplitudes: peak velocity (1) / acceleration (2) vector norm per enabled receiver (update_syn_probes first)."""
        self.calculate_seismograms()
        self.scale_seismograms()
        out = np.zeros(len(self.ncomp), np.float32)
        n = lib().ko_engine_shake(self.h, differentiate, _fp(out))
        return out[:n]

    def arias_intensities(self):
        """get_arias_intensities per enabled receiver."""
        return self.peak_amplitudes(0)

    def calculate_seismograms(self):
        lib().ko_engine_calculate_seismograms(self.h)

    def scale_seismograms(self):
        lib().ko_engine_scale_seismograms(self.h)

    def get_misfits(self):
        """update_misfits + get_misfits: (m[nmis], n[nmis]) for enabled receivers, and the global misfit."""
        nmax = sum(self.ncomp)
        m = np.zeros(nmax, np.float32)
        n = np.zeros(nmax, np.float32)
        k = lib().ko_engine_get_misfits(self.h, _fp(m), _fp(n), nmax)
        return m[:k], n[:k], float(lib().ko_engine_get_global_misfit(self.h))

    def displacement(self, irec1, icomp1, maxn=1 << 20):
        out = np.zeros(maxn, np.float32)
        lo = C.c_int()
        n = lib().ko_engine_get_displacement(self.h, irec1, icomp1, C.byref(lo), _fp(out), maxn)
        return lo.value, out[:n].copy()

    def synthetic(self, irec1, icomp1, which=1, maxn=1 << 20):
        out = np.zeros(maxn, np.float32)
        lo = C.c_int()
        n = lib().ko_engine_get_synthetic(self.h, irec1, icomp1, which, C.byref(lo), _fp(out), maxn)
        return lo.value, out[:n].copy()

    def reference(self, irec1, icomp1, which=1, maxn=1 << 20):
        out = np.zeros(maxn, np.float32)
        lo = C.c_int()
        L = lib()
        L.ko_engine_get_reference.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), c_float_p, C.c_int]
        n = L.ko_engine_get_reference(self.h, irec1, icomp1, which, C.byref(lo), _fp(out), maxn)
        return lo.value, out[:n].copy()

    def centroid_geometry(self, irec1, ncent, dtype):
        """GeoRec-compatible records of the current centroid table at receiver irec1."""
        out = np.zeros(ncent, dtype)
        for i in range(ncent):
            lib().ko_engine_centroid_geometry(self.h, irec1, i, out[i:i + 1].ctypes.data_as(C.c_void_p))
        return out

    def receiver_geometry(self, irec1):
        a, b, d = C.c_double(), C.c_double(), C.c_double()
        lib().ko_engine_receiver_geometry(self.h, irec1, C.byref(a), C.byref(b), C.byref(d))
        return a.value, b.value, d.value

    def close(self):
        if self.h:
            lib().ko_engine_destroy(self.h)
            self.h = None


# ---------------------------------------------------------------- eikonal sources
class CrustProfile(C.Structure):
    _fields_ = [("vp", C.c_float * 8), ("vs", C.c_float * 8), ("rho", C.c_float * 8), ("thickness", C.c_float * 7)]


def crust_profile(vp, vs, rho, thickness):
    p = CrustProfile()
    for i in range(8):
        p.vp[i], p.vs[i], p.rho[i] = vp[i], vs[i], rho[i]
    for i in range(7):
        p.thickness[i] = thickness[i]
    return p


def crust_thickness(profile):
    """crust2x2_get_profile_averages (crust2x2.f90:146-168): total crustal thickness."""
    v = [C.c_float() for _ in range(4)]
    lib().ko_crust_profile_averages(C.byref(profile), *[C.byref(x) for x in v])
    return v[3].value


def discretize_eikonal(sourcetype, params, effective_dt, prof_speed, con_points, con_normals):
    """source_eikonal (4) / source_mt_eikonal (5): (centroids[n,10], moment, risetime, grid) or raises."""
    L = lib()
    p = np.ascontiguousarray(params, np.float32)
    cp = np.ascontiguousarray(con_points, np.float32)
    cn = np.ascontiguousarray(con_normals, np.float32)
    out = C.POINTER(Centroid)()
    mo, ri = C.c_float(), C.c_float()
    gs = (C.c_int * 2)()
    n = L.ko_psm_to_tdsm_eikonal(C.c_int(sourcetype), _fp(p), C.c_float(effective_dt), C.byref(prof_speed),
                                 C.c_int(len(cp)), _fp(cp), _fp(cn), C.byref(out), C.byref(mo), C.byref(ri), gs)
    if n < 0:
        raise ValueError("Empty rupture area" if n == -1 else "position of nucleation point is outside of rupture region")
    arr = np.ctypeslib.as_array(C.cast(out, c_float_p), (n, 10)).copy() if n else np.zeros((0, 10), np.float32)
    C.CDLL(None).free(out)
    return arr, mo.value, ri.value, (gs[0], gs[1])
