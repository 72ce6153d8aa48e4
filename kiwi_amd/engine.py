"""Python host side of the C-ABI: one ``Engine`` == one ``minimizer`` process of the reference.

Method names follow the reference's stdin commands / minimizer_engine procedures
(minimizer.f90:1729-1811) so that the calling code reads like python/tunguska/seismosizer.py;
``make_misfits_for_sources`` and ``make_global_misfits`` restate the two Python functions that sit
directly on the path (seismosizer.py:682-722, 843-922)."""
import ctypes as C

import numpy as np

from . import lib as _lib
from .lib import KiwiHipError, c_float_p, c_int_p, c_double_p

SOURCE_TYPES = {"bilateral": 1, "circular": 2, "point_lp": 3, "eikonal": 4, "mt_eikonal": 5,
                "moment_tensor": 6}          # source_all.f90:88-98
NORMS = {"l2norm": 1, "l1norm": 2, "ampspec_l2norm": 3, "ampspec_l1norm": 4, "scalar_product": 5,
         "peak": 6, "floating_l2norm": 7, "floating_l1norm": 8}      # comparator.f90:137-146

GEOREC = np.dtype([("row", np.int32, 4), ("w", np.float32, 4), ("ishift", np.int32), ("wfrac", np.float32),
                   ("f", np.float32, 6), ("cl", np.float32), ("sl", np.float32), ("flags", np.int32),
                   ("pad", np.int32)])


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def discretize(sourcetype, params, effective_dt):
    """psm_set + psm_to_tdsm on the host (kiwi_hip_discretize): (centroids[n,10], moment, risetime)."""
    L = _lib.load()
    st = SOURCE_TYPES.get(sourcetype, sourcetype)
    p = np.ascontiguousarray(params, np.float32)
    n, mo, ri = C.c_int(), C.c_float(), C.c_float()
    rc = L.kiwi_hip_discretize(st, _fp(p), len(p), effective_dt, None, 0, C.byref(n), C.byref(mo), C.byref(ri))
    if rc != 0:
        raise KiwiHipError("discretize failed (rc=%d): unsupported source type or wrong parameter count" % rc)
    cent = np.zeros((n.value, 10), np.float32)
    rc = L.kiwi_hip_discretize(st, _fp(p), len(p), effective_dt, _fp(cent), n.value, C.byref(n), C.byref(mo),
                               C.byref(ri))
    if rc != 0:
        raise KiwiHipError("discretize failed (rc=%d)" % rc)
    return cent, mo.value, ri.value


def principal_axes(sourcetype, params):
    """`get_principal_axes`: ((P azimuth, P polar angle), (T azimuth, T polar angle)) in degrees; bilateral sources only."""
    L = _lib.load()
    p = np.ascontiguousarray(params, np.float32)
    pax, tax = np.zeros(2, np.float32), np.zeros(2, np.float32)
    if L.kiwi_hip_principal_axes(SOURCE_TYPES[sourcetype], _fp(p), _fp(pax), _fp(tax)) != 0:
        raise KiwiHipError("principal axes are defined for bilateral sources only")
    return pax, tax


def pack_crust_profile(vp, vs, rho, thickness):
    """t_crust2x2_1d_profile (crust2x2.f90:45-50) as the 31 floats the C-ABI takes."""
    out = np.concatenate([np.asarray(vp, np.float32), np.asarray(vs, np.float32), np.asarray(rho, np.float32),
                          np.asarray(thickness, np.float32)])
    if out.shape != (31,):
        raise KiwiHipError("a crust profile is vp[8], vs[8], rho[8], thickness[7]")
    return np.ascontiguousarray(out)


_EIKONAL_ERRORS = {5: "Empty rupture area", 6: "position of nucleation point is outside of rupture region"}


def discretize_eikonal(sourcetype, params, effective_dt, rupture_profile, con_points, con_normals):
    """psm_set + psm_to_tdsm of `eikonal` / `mt_eikonal` (kiwi_hip_discretize_eikonal)."""
    L = _lib.load()
    st = SOURCE_TYPES.get(sourcetype, sourcetype)
    p = np.ascontiguousarray(params, np.float32)
    prof = np.ascontiguousarray(rupture_profile, np.float32)
    pts = np.ascontiguousarray(con_points, np.float32).reshape(-1, 3)
    nrm = np.ascontiguousarray(con_normals, np.float32).reshape(-1, 3)
    n, mo, ri = C.c_int(), C.c_float(), C.c_float()

    def call(cent, maxcent):
        rc = L.kiwi_hip_discretize_eikonal(st, _fp(p), len(p), effective_dt, _fp(prof), len(pts), _fp(pts), _fp(nrm),
                                           cent, maxcent, C.byref(n), C.byref(mo), C.byref(ri))
        if rc != 0:
            raise KiwiHipError("discretize failed (rc=%d): %s" % (rc, _EIKONAL_ERRORS.get(rc, "bad arguments")))

    call(None, 0)
    cent = np.zeros((n.value, 10), np.float32)
    call(_fp(cent), n.value)
    return cent, mo.value, ri.value


def _pieces(n, piece, st):
    """The pieces [first, first + count) kiwi_hip_misfits_for_params cuts a list of n sources into (list order; it works
    from the last to the first): `piece` sources each, and for the eikonal types the last one as a ramp of half, a quarter,
    an eighth and an eighth of it -- worked on in the order eighth, eighth, quarter, half -- so that the device starts early."""
    out = [(s0, min(piece, n - s0)) for s0 in range(0, n, piece)]
    if st in (4, 5) and len(out) >= 2 and out[-1][1] >= 8:
        first, cnt = out.pop()
        e, q = cnt // 8, cnt // 4
        h = cnt - 2 * e - q
        out += [(first, h), (first + h, q), (first + h + q, e), (first + h + q + e, e)]
    return out


class Engine:
    def __init__(self, device=0, ndev=None):
        """device: the GPU of a one-device engine; ndev: instead, ONE engine over that many devices of this process
        (kiwi_hip_init_multi; 0 = all visible): setters are repeated on every device, misfits_for_params /
        make_misfits_for_sources shard their trial list over them."""
        self.L = _lib.load()
        self.h = C.c_void_p()
        rc = self.L.kiwi_hip_init(device, C.byref(self.h)) if ndev is None else self.L.kiwi_hip_init_multi(ndev, C.byref(self.h))
        if rc != 0:
            buf = C.create_string_buffer(512)
            self.L.kiwi_hip_last_error(None, buf, 512)
            self.h = None
            raise KiwiHipError("kiwi_hip_init: " + buf.value.decode())
        self.nsrc = 0

    def ndevices(self):
        n = C.c_int(0)
        self._ck(self.L.kiwi_hip_ndevices(self.h, C.byref(n)), "ndevices")
        return n.value

    def set_arithmetic(self, mode):
        """'exact' (default: every fp32 multiply and add of the superposition rounded on its own, as the reference's host does:
        bit-identical to the CPU oracle) or 'fused' (multiply + consuming add as one fused multiply-add: misfits within 1e-6 of
        the norm factor, half the vector instructions); include/kiwi_hip.h KIWI_ARITH_*."""
        m = {"exact": 0, "fused": 1, 0: 0, 1: 1}.get(mode)
        if m is None:
            raise ValueError("arithmetic: 'exact' or 'fused'")
        self._ck(self.L.kiwi_hip_set_arithmetic(self.h, m), "set_arithmetic")

    def arithmetic(self):
        m = C.c_int(0)
        self._ck(self.L.kiwi_hip_get_arithmetic(self.h, C.byref(m)), "get_arithmetic")
        return "fused" if m.value == 1 else "exact"

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc, what):
        if rc != 0:
            buf = C.create_string_buffer(1024)
            self.L.kiwi_hip_last_error(self.h, buf, 1024)
            raise KiwiHipError("%s: nok > %s" % (what, buf.value.decode()))

    def close(self):
        if self.h:
            self.L.kiwi_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ setup commands
    def set_database(self, dt, dx, dz, firstx, firstz, data, first, nsamp):
        """data[nx,nz,ng,L] float32, first/nsamp[nx,nz,ng] int32 (see kiwi_hip_set_gfdb)."""
        data = np.ascontiguousarray(data, np.float32)
        first = np.ascontiguousarray(first, np.int32)
        nsamp = np.ascontiguousarray(nsamp, np.int32)
        nx, nz, ng, L = data.shape
        assert first.shape == (nx, nz, ng) and nsamp.shape == (nx, nz, ng)
        self._ck(self.L.kiwi_hip_set_gfdb(self.h, nx, nz, ng, L, dt, dx, dz, firstx, firstz, _fp(data), _ip(first),
                                          _ip(nsamp)), "set_database")
        self.dt = dt

    def set_local_interpolation(self, kind):
        self._interp = (kind in ("bilinear", True, 1))
        self._ck(self.L.kiwi_hip_set_interp(self.h, int(self._interp), getattr(self, "_xus", 1),
                                            getattr(self, "_zus", 1)), "set_local_interpolation")

    def set_spacial_undersampling(self, xus, zus):
        self._xus, self._zus = xus, zus
        self._ck(self.L.kiwi_hip_set_interp(self.h, int(getattr(self, "_interp", False)), xus, zus),
                 "set_spacial_undersampling")

    def set_effective_dt(self, dt):
        self._ck(self.L.kiwi_hip_set_effective_dt(self.h, dt), "set_effective_dt")

    def set_source_location(self, lat_deg, lon_deg, ref_time=0.0):
        self._ck(self.L.kiwi_hip_set_source_location(self.h, lat_deg, lon_deg, ref_time), "set_source_location")

    def set_receivers(self, lat_deg, lon_deg, depth, components):
        n = len(lat_deg)
        lat = np.ascontiguousarray(lat_deg, np.float64)
        lon = np.ascontiguousarray(lon_deg, np.float64)
        dep = np.ascontiguousarray(np.zeros(n) if depth is None else depth, np.float32)
        arr = (C.c_char_p * n)(*[c.encode() for c in components])
        self._ck(self.L.kiwi_hip_set_receivers(self.h, n, lat.ctypes.data_as(c_double_p),
                                               lon.ctypes.data_as(c_double_p), _fp(dep), arr), "set_receivers")
        self.components = list(components)
        self.enabled = [len(c) > 0 for c in components]

    def switch_receiver(self, irec, state):
        self._ck(self.L.kiwi_hip_switch_receiver(self.h, irec, int(bool(state))), "switch_receiver")
        self.enabled[irec - 1] = bool(state)

    def set_ref_seismogram(self, irec, icomp, first, data):
        d = np.ascontiguousarray(data, np.float32)
        self._ck(self.L.kiwi_hip_set_reference(self.h, irec, icomp, first, len(d), _fp(d)), "set_ref_seismograms")

    def set_misfit_taper(self, irec, x, y):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        self._ck(self.L.kiwi_hip_set_taper(self.h, irec, len(x), _fp(x), _fp(y)), "set_misfit_taper")

    def set_misfit_filter(self, irec, x, y):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        self._ck(self.L.kiwi_hip_set_filter(self.h, irec, len(x), _fp(x), _fp(y)), "set_misfit_filter")

    def set_misfit_method(self, name):
        if not isinstance(name, int) and name not in NORMS:
            raise KiwiHipError("set_misfit_method: nok > unknown norm: %s" % name)   # minimizer.f90:842-873
        self._ck(self.L.kiwi_hip_set_misfit_method(self.h, NORMS.get(name, name)), "set_misfit_method")

    def shift_ref_seismogram(self, irec, shift):
        """`shift_ref_seismogram ireceiver shift` (seconds)."""
        self._ck(self.L.kiwi_hip_shift_ref_seismogram(self.h, irec, shift), "shift_ref_seismogram")

    def autoshift_ref_seismogram(self, irec, min_shift, max_shift, isrc=0):
        """`autoshift_ref_seismogram ireceiver min-shift max-shift` against uploaded source `isrc`; returns the shifts
        applied, in seconds (one per receiver for irec == 0)."""
        out = np.zeros(len(self.components) if irec == 0 else 1, np.float32)
        self._ck(self.L.kiwi_hip_autoshift_ref_seismogram(self.h, irec, min_shift, max_shift, isrc, _fp(out)),
                 "autoshift_ref_seismogram")
        return out

    def set_floating_shiftrange(self, irec, min_shift, max_shift):
        """`set_floating_shiftrange ireceiver min-shift max-shift` (seconds; ireceiver 0 = all receivers)."""
        self._ck(self.L.kiwi_hip_set_floating_shiftrange(self.h, irec, min_shift, max_shift), "set_floating_shiftrange")

    def get_floating_shifts(self, isrc0=0, nsrc=None):
        """`get_floating_shifts` for a batch: [nsrc, n_enabled_receivers] in seconds."""
        nsrc = self.nsrc - isrc0 if nsrc is None else nsrc
        nen = sum(1 for e, c in zip(self.enabled, self.components) if e and len(c))
        out = np.zeros((nsrc, nen), np.float32)
        self._ck(self.L.kiwi_hip_get_floating_shifts(self.h, isrc0, nsrc, _fp(out)), "get_floating_shifts")
        return out

    def set_synthetics_factor(self, f):
        self._ck(self.L.kiwi_hip_set_synthetics_factor(self.h, f), "set_synthetics_factor")

    # ------------------------------------------------------------------ trial sources
    def set_sources(self, tables, moments=None, risetimes=None):
        """tables: list of centroid tables [n_i,10]."""
        ofs = np.zeros(len(tables) + 1, np.int32)
        ofs[1:] = np.cumsum([len(t) for t in tables])
        cent = np.ascontiguousarray(np.concatenate(tables, 0) if len(tables) else np.zeros((0, 10)), np.float32)
        mo = np.ascontiguousarray(np.ones(len(tables)) if moments is None else moments, np.float32)
        ri = np.ascontiguousarray(np.zeros(len(tables)) if risetimes is None else risetimes, np.float32)
        self._ck(self.L.kiwi_hip_set_sources(self.h, len(tables), _ip(ofs), _fp(cent), _fp(mo), _fp(ri)),
                 "set_sources")
        self.nsrc = len(tables)

    def set_source_crust(self, rupture_profile, origin_profile):
        """The two CRUST2.0 look-ups of set_source_location (see include/kiwi_hip.h), 31 floats each."""
        a = np.ascontiguousarray(rupture_profile, np.float32)
        b = np.ascontiguousarray(origin_profile, np.float32)
        self._ck(self.L.kiwi_hip_set_source_crust(self.h, _fp(a), _fp(b)), "set_source_location")

    def set_source_crustal_thickness_limit(self, limit):
        self._ck(self.L.kiwi_hip_set_source_crustal_thickness_limit(self.h, limit),
                 "set_source_crustal_thickness_limit")

    def get_source_crustal_thickness(self):
        t = C.c_float()
        self._ck(self.L.kiwi_hip_get_source_crustal_thickness(self.h, C.byref(t)), "get_source_crustal_thickness")
        return t.value

    def set_source_constraints(self, points, normals):
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        self._ck(self.L.kiwi_hip_set_source_constraints(self.h, len(pts), _fp(pts), _fp(nrm)),
                 "set_source_constraints")

    def set_source_params(self, sourcetype, params):
        """Batch form of `set_source_params type p1..pn`: params[nsrc, nparams]."""
        p = np.ascontiguousarray(np.atleast_2d(params), np.float32)
        st = SOURCE_TYPES.get(sourcetype, sourcetype)
        if p.shape[1] != self.L.kiwi_hip_source_nparams(st):
            raise KiwiHipError("set_source_params: wrong number of source parameters")
        self.nsrc = p.shape[0]          # the batch is uploaded even when the call reports that no source could be discretised
        self._ck(self.L.kiwi_hip_set_sources_params(self.h, st, p.shape[0], _fp(p)), "set_source_params")

    def get_source_status(self, isrc0=0, nsrc=None):
        """Per uploaded source: 0 discretised, else the discretiser's error code (see source_status_message)."""
        nsrc = self.nsrc - isrc0 if nsrc is None else nsrc
        st = np.zeros(nsrc, np.int32)
        self._ck(self.L.kiwi_hip_get_source_status(self.h, isrc0, nsrc, _ip(st)), "set_source_params")
        return st

    def source_status_message(self, code):
        buf = C.create_string_buffer(256)
        self.L.kiwi_hip_source_status_message(int(code), buf, 256)
        return buf.value.decode()

    # ------------------------------------------------------------------ hot path
    def eval(self, isrc0=0, nsrc=None):
        nsrc = self.nsrc - isrc0 if nsrc is None else nsrc
        self._ck(self.L.kiwi_hip_eval(self.h, isrc0, nsrc), "get_misfits")

    def sync(self):
        self._ck(self.L.kiwi_hip_sync(self.h), "sync")

    def nmisfits(self):
        n = C.c_int()
        self._ck(self.L.kiwi_hip_nmisfits(self.h, C.byref(n)), "get_misfits")
        return n.value

    def get_misfits(self, isrc0=0, nsrc=None):
        """(misfit[nsrc,nmis], norm[nsrc,nmis], global[nsrc]) of already evaluated sources."""
        nsrc = self.nsrc - isrc0 if nsrc is None else nsrc
        nm = self.nmisfits()
        m = np.zeros((nsrc, nm), np.float32)
        n = np.zeros((nsrc, nm), np.float32)
        g = np.zeros(nsrc, np.float32)
        self._ck(self.L.kiwi_hip_get_misfits(self.h, isrc0, nsrc, _fp(m), _fp(n), _fp(g)), "get_misfits")
        return m, n, g

    def global_misfits_device(self, isrc0=0, nsrc=None):
        """The global misfits of evaluated sources where they lie: an object with `__cuda_array_interface__` (fp32 [nsrc] on this
        engine's device; torch.as_tensor(obj, device=...) wraps it without a copy), valid until the next eval / upload.  What the
        multi-GPU all-gather takes (kiwi_amd/shard.py gather_global_misfits_device)."""
        nsrc = self.nsrc - isrc0 if nsrc is None else nsrc
        ptr = C.c_void_p()
        self._ck(self.L.kiwi_hip_get_global_misfits_device(self.h, isrc0, nsrc, C.byref(ptr)), "get_global_misfits_device")

        class _DeviceArray:
            __cuda_array_interface__ = {"shape": (nsrc,), "typestr": "<f4", "data": (ptr.value or 0, False), "version": 2, "strides": None}
        return _DeviceArray()

    def set_keep_synthetics(self, which):
        self._ck(self.L.kiwi_hip_set_keep_synthetics(self.h, which), "set_keep_synthetics")

    def get_synthetics(self, isrc, irec, icomp, which=1, maxn=1 << 20):
        out = np.zeros(maxn, np.float32)
        first, n = C.c_int(), C.c_int()
        self._ck(self.L.kiwi_hip_get_synthetics(self.h, isrc, irec, icomp, which, C.byref(first), C.byref(n),
                                                _fp(out), maxn), "output_seismograms")
        return first.value, out[:n.value].copy()

    def get_amp_spectrum(self, irec, icomp, probe="synthetics", filtered=False, isrc=0, maxn=1 << 20):
        """`output_seismogram_spectra` for one receiver component: (df, amplitudes[ntrans / 2 + 1])."""
        out = np.zeros(maxn, np.float32)
        df, n = C.c_float(), C.c_int()
        self._ck(self.L.kiwi_hip_get_amp_spectrum(self.h, isrc, irec, icomp, 0 if probe == "references" else 1, int(bool(filtered)),
                                                  C.byref(df), C.byref(n), _fp(out), maxn), "output_seismogram_spectra")
        return float(df.value), out[:n.value].copy()

    def get_cross_correlations(self, irec, min_shift, max_shift, isrc=0):
        """`output_cross_correlations` for one receiver: (first shift in samples, cc[ncomp, nshift])."""
        dt = self.dt
        ns = int(round(max_shift / dt)) - int(round(min_shift / dt)) + 1
        nc = len(self.components[irec - 1])
        out = np.zeros(max(ns, 1) * max(nc, 1), np.float32)
        first, n = C.c_int(), C.c_int()
        self._ck(self.L.kiwi_hip_get_cross_correlations(self.h, isrc, irec, min_shift, max_shift, C.byref(first), C.byref(n),
                                                        _fp(out), len(out)), "output_cross_correlations")
        return first.value, out[:nc * n.value].reshape(nc, n.value) if n.value else np.zeros((0, 0), np.float32)

    def get_peak_amplitudes(self, differentiate, isrc=0):
        """`get_peak_amplitudes`: per enabled receiver the peak vector norm of the velocity (1) or acceleration (2)."""
        out = np.zeros(sum(1 for e in self.enabled if e), np.float32)
        self._ck(self.L.kiwi_hip_get_peak_amplitudes(self.h, isrc, differentiate, _fp(out)), "get_peak_amplitudes")
        return out

    def get_arias_intensities(self, isrc=0):
        """`get_arias_intensities` per enabled receiver."""
        out = np.zeros(sum(1 for e in self.enabled if e), np.float32)
        self._ck(self.L.kiwi_hip_get_arias_intensities(self.h, isrc, _fp(out)), "get_arias_intensities")
        return out

    def get_source_centroids(self, isrc=0):
        """The discretised source the engine holds for trial `isrc`: centroids[n, 10] (`output_source_model`)."""
        n = C.c_int()
        self._ck(self.L.kiwi_hip_get_source_centroids(self.h, isrc, 0, C.byref(n), None), "output_source_model")
        cent = np.zeros((n.value, 10), np.float32)
        self._ck(self.L.kiwi_hip_get_source_centroids(self.h, isrc, n.value, C.byref(n), _fp(cent)), "output_source_model")
        return cent

    def get_reference(self, irec, icomp, which=1, maxn=1 << 20):
        """(first sample index, samples) of a reference probe: 1 plain, 2 tapered, 3 filtered."""
        first, n = C.c_int(), C.c_int()
        buf = np.zeros(maxn, np.float32)
        self._ck(self.L.kiwi_hip_get_reference(self.h, irec, icomp, which, C.byref(first), C.byref(n), _fp(buf), maxn),
                 "output_seismograms")
        return first.value, buf[:min(n.value, maxn)].copy()

    def kernel_ms(self):
        ms = np.zeros(4, np.float32)
        ln = np.zeros(3, np.int32)
        self._ck(self.L.kiwi_hip_get_kernel_ms(self.h, _fp(ms), _ip(ln)), "kernel_ms")
        return ms, ln

    def get_geometry(self, isrc, irec, maxcent=100000):
        rec = np.zeros(maxcent, GEOREC)
        n = C.c_int()
        self._ck(self.L.kiwi_hip_get_geometry(self.h, isrc, irec, maxcent, C.byref(n), rec.ctypes.data_as(C.c_void_p)),
                 "get_geometry")
        return rec[:n.value].copy()

    def receiver_geometry(self, irec):
        a, b, d = C.c_double(), C.c_double(), C.c_double()
        self._ck(self.L.kiwi_hip_get_receiver_geometry(self.h, irec, C.byref(a), C.byref(b), C.byref(d)),
                 "output_distances")
        return a.value, b.value, d.value

    def device_bytes(self):
        b = C.c_longlong()
        self.L.kiwi_hip_get_device_bytes(self.h, C.byref(b))
        return b.value

    # ------------------------------------------------------------------ seismosizer.py counterparts
    def misfits_for_params(self, sourcetype, params, piece=0):
        """A whole trial list in one call (kiwi_hip_misfits_for_params): the list is evaluated in pieces of `piece` sources
        (0: 128 for the eikonal types, 2048 otherwise), the host discretiser of one piece running while the device
        evaluates another; afterwards the engine holds the head of the list (sources 0 .. piece - 1).  Returns (misfit[N,nmis], norm[N,nmis], global[N], status[N]); piece size does not change a bit."""
        p = np.ascontiguousarray(np.atleast_2d(params), np.float32)
        st = SOURCE_TYPES.get(sourcetype, sourcetype)
        if p.shape[1] != self.L.kiwi_hip_source_nparams(st):
            raise KiwiHipError("set_source_params: wrong number of source parameters")
        if piece <= 0:
            piece = 128 if st in (4, 5) else 2048
        N, nm = p.shape[0], self.nmisfits()
        m = np.zeros((N, nm), np.float32)
        n = np.zeros((N, nm), np.float32)
        g = np.zeros(N, np.float32)
        status = np.zeros(N, np.int32)
        try:
            self._ck(self.L.kiwi_hip_misfits_for_params(self.h, st, N, _fp(p), piece, _fp(m), _fp(n), _fp(g), _ip(status)),
                     "get_misfits")
        except KiwiHipError:
            self.nsrc = 0                  # (a call that fails midway leaves no batch this object may index)
            raise
        # pieces are evaluated from the end of the list: the context holds the first piece that uploaded anything; with a
        # multi-device engine that is shard 0's (the first device keeps the head of the list)
        # (kiwi_hip_misfits_for_params: a list of ONE source takes the one-device path; otherwise min(N, devices) shards, shard i
        # = [N i / k, N (i + 1) / k))
        ndev = self.ndevices()
        k = 1 if (ndev == 1 or N < 2) else min(N, ndev)
        n0 = N // k
        held = 0
        for s0, cnt in _pieces(n0, piece, st):
            if np.any(status[s0:s0 + cnt] == 0):
                held = cnt
                break
        if held:
            self.nsrc = held               # (no piece uploaded anything: the context keeps what it held)
        return m, n, g, status

    def make_misfits_for_sources(self, sourcetype=None, params=None, piece=0):
        """seismosizer.py:682-722: returns (misfits_by_src[N_s,N_r,N_k], norms_by_src[...], failings) -- float64 arrays,
        receivers in file order, components in string order, disabled receivers as zeros; `failings` lists the indices of
        the trial sources the engine rejected (`SeismosizersReturnedErrors` there, :716-717), whose rows stay zero.
        With `params` the trial list goes through misfits_for_params (discretiser and device overlapped, pieces of
        `piece` sources); without, the sources uploaded before are evaluated."""
        if params is not None:
            m, n, _, status = self.misfits_for_params(sourcetype, params, piece)
            nsrc = len(m)
        else:
            status = self.get_source_status()
            self.eval()
            m, n, _ = self.get_misfits()
            nsrc = self.nsrc
        failings = [int(i) for i in np.nonzero(status)[0]]
        nrec = len(self.components)
        nk = max([len(c) for c in self.components] + [1])
        mis = np.zeros((nsrc, nrec, nk))
        nor = np.zeros((nsrc, nrec, nk))
        j = 0
        for ir, comps in enumerate(self.components):
            if not self.enabled[ir]:
                continue
            k = len(comps)
            mis[:, ir, :k] = m[:, j:j + k]
            nor[:, ir, :k] = n[:, j:j + k]
            j += k
        return mis, nor, failings


def make_global_misfits(misfits_by_src, norms_by_src, outer_norm="l2norm", receiver_weights=None, receiver_mask=None,
                        anarchy=False, bootstrap=False, rng=None):
    """seismosizer.py:843-922: per-source global misfit and per source-receiver misfits from [N_s,N_r,N_k]
    arrays, float64.  `anarchy` divides each receiver's weight by its norm (:884-888); `bootstrap` multiplies
    the weights by a resampling count drawn over the enabled receivers (:855-869; sqrt of it for l2, :901-902).
    Differences, deliberate: the reference draws from numpy's global RandomState -- pass `rng` (a
    numpy Generator) for reproducible draws; its anarchy + per-receiver-weights combination under l2norm
    raises a broadcasting error for more than one source (:898), here it works as under l1norm."""
    m = np.asarray(misfits_by_src, np.float64)
    n = np.asarray(norms_by_src, np.float64)
    nrec = m.shape[1]
    w = np.ones(nrec) if receiver_weights is None else np.broadcast_to(np.asarray(receiver_weights, np.float64), (nrec,))
    rweights = np.tile(w, (m.shape[0], 1))
    if bootstrap:
        mask = np.ones(nrec, bool) if receiver_mask is None else np.asarray(receiver_mask, bool)
        if receiver_weights is not None:
            mask = np.logical_and(mask, w != 0)
        enabled = np.arange(nrec)[mask]
        rng = np.random.default_rng() if rng is None else rng
        draw = enabled[rng.integers(0, len(enabled), len(enabled))]
        bweights = np.bincount(draw, minlength=nrec).astype(np.float64)
    if outer_norm == "l1norm":
        m_sr, n_sr = m.sum(2), n.sum(2)
    elif outer_norm == "l2norm":
        m_sr, n_sr = np.sqrt((m ** 2).sum(2)), np.sqrt((n ** 2).sum(2))
    else:
        raise KiwiHipError("unknown norm method: %s" % outer_norm)
    if anarchy:
        rweights = np.maximum(rweights / np.where(n_sr != 0., n_sr, -1.), 0.)
    if bootstrap:
        rweights = rweights * (bweights if outer_norm == "l1norm" else np.sqrt(bweights))
    m_sr = m_sr * rweights
    n_sr = n_sr * rweights
    with np.errstate(divide="ignore", invalid="ignore"):
        if outer_norm == "l1norm":
            ms, ns = m_sr.sum(1), n_sr.sum(1)
            g = np.where(ns > 0, ms / ns, -1.)
        else:
            ms, ns = (m_sr ** 2).sum(1), (n_sr ** 2).sum(1)
            g = np.where(ns > 0, np.sqrt(ms / ns), -1.)
    return np.where(g < 0, np.nan, g), m_sr
