"""Loader for the product library libkiwi_hip.so.  There is NO fallback: if the library (or a
GPU, at init time) is missing the caller gets an exception."""
import ctypes as C
import os
import sys
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KIWI_HIP_LIB", os.path.join(HERE, "libkiwi_hip.so"))   # override: A/B builds
HEADER = os.path.join(os.path.dirname(HERE), "include", "kiwi_hip.h")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_double_p = C.POINTER(C.c_double)


class KiwiHipError(RuntimeError):
    """A C-ABI call returned non-zero; the message is the library's last error
    (the reference's '<cmd>: nok >' line, minimizer.f90:1689-1696)."""


# kiwi_hip_residual_fn: (user, k, m, n, xs[k][n], fv[k][m]) -> int
RESIDUAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))


def build(force=False):
    """Compile kiwi_amd/csrc for gfx950 into kiwi_amd/libkiwi_hip.so (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "csrc")])
    return LIB_PATH


def declared_symbols():
    """Every function the public header declares."""
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"\b(kiwi_hip_[a-z_0-9]+)\s*\(", txt)))


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KiwiHipError("%s not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C kiwi_amd/csrc`" % LIB_PATH)
    # One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 (same soname as /opt/rocm's, which this library is
    # linked against); whichever copy initialises the GPU first owns it, the other then reports "no ROCm-capable device"
    # (measured on the GPU box: this library + kiwi_hip_init, then `import torch` -> torch.cuda sees no GPU; torch's copy
    # initialised by another module, then kiwi_hip_init -> no device).  Loaded AFTER torch this library binds to torch's
    # copy and both work, so torch goes first wherever it is installed.
    # (only where torch is installed at all: the library itself needs no torch)
    import importlib.util
    if "torch" not in sys.modules and os.environ.get("KIWI_HIP_WITHOUT_TORCH", "0") != "1" and importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception:      # a torch that does not import: nothing to order
            pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    sig = {
        "kiwi_hip_init": [C.c_int, C.POINTER(vp)],
        "kiwi_hip_init_multi": [C.c_int, C.POINTER(vp)],
        "kiwi_hip_ndevices": [vp, c_int_p],
        "kiwi_hip_set_arithmetic": [vp, C.c_int],
        "kiwi_hip_get_arithmetic": [vp, c_int_p],
        "kiwi_hip_destroy": [vp],
        "kiwi_hip_last_error": [vp, C.c_char_p, C.c_int],
        "kiwi_hip_set_gfdb": [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                              C.c_float, c_float_p, c_int_p, c_int_p],
        "kiwi_hip_set_interp": [vp, C.c_int, C.c_int, C.c_int],
        "kiwi_hip_set_effective_dt": [vp, C.c_float],
        "kiwi_hip_set_source_location": [vp, C.c_float, C.c_float, C.c_double],
        "kiwi_hip_set_receivers": [vp, C.c_int, c_double_p, c_double_p, c_float_p, C.POINTER(C.c_char_p)],
        "kiwi_hip_switch_receiver": [vp, C.c_int, C.c_int],
        "kiwi_hip_set_reference": [vp, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p],
        "kiwi_hip_set_taper": [vp, C.c_int, C.c_int, c_float_p, c_float_p],
        "kiwi_hip_set_filter": [vp, C.c_int, C.c_int, c_float_p, c_float_p],
        "kiwi_hip_set_misfit_method": [vp, C.c_int],
        "kiwi_hip_set_synthetics_factor": [vp, C.c_float],
        "kiwi_hip_shift_ref_seismogram": [vp, C.c_int, C.c_float],
        "kiwi_hip_autoshift_ref_seismogram": [vp, C.c_int, C.c_float, C.c_float, C.c_int, c_float_p],
        "kiwi_hip_set_floating_shiftrange": [vp, C.c_int, C.c_float, C.c_float],
        "kiwi_hip_get_floating_shifts": [vp, C.c_int, C.c_int, c_float_p],
        "kiwi_hip_source_nparams": [C.c_int],
        "kiwi_hip_discretize": [C.c_int, c_float_p, C.c_int, C.c_float, c_float_p, C.c_int, c_int_p, c_float_p,
                                c_float_p],
        "kiwi_hip_set_source_crust": [vp, c_float_p, c_float_p],
        "kiwi_hip_set_source_crustal_thickness_limit": [vp, C.c_float],
        "kiwi_hip_get_source_crustal_thickness": [vp, c_float_p],
        "kiwi_hip_set_source_constraints": [vp, C.c_int, c_float_p, c_float_p],
        "kiwi_hip_discretize_eikonal": [C.c_int, c_float_p, C.c_int, C.c_float, c_float_p, C.c_int, c_float_p,
                                        c_float_p, c_float_p, C.c_int, c_int_p, c_float_p, c_float_p],
        "kiwi_hip_set_sources": [vp, C.c_int, c_int_p, c_float_p, c_float_p, c_float_p],
        "kiwi_hip_set_sources_params": [vp, C.c_int, C.c_int, c_float_p],
        "kiwi_hip_get_source_status": [vp, C.c_int, C.c_int, c_int_p],
        "kiwi_hip_source_status_message": [C.c_int, C.c_char_p, C.c_int],
        "kiwi_hip_misfits_for_params": [vp, C.c_int, C.c_int, c_float_p, C.c_int, c_float_p, c_float_p, c_float_p, c_int_p],
        "kiwi_hip_effective_cpus": [],
        "kiwi_hip_eikonal_cache_stats": [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_int],
        "kiwi_hip_fast_marching": [c_float_p, C.c_int, C.c_int, c_float_p, c_float_p, c_float_p, C.c_float, C.c_int, c_float_p,
                                   C.POINTER(C.c_longlong)],
        "kiwi_hip_eval": [vp, C.c_int, C.c_int],
        "kiwi_hip_sync": [vp],
        "kiwi_hip_set_keep_synthetics": [vp, C.c_int],
        "kiwi_hip_nmisfits": [vp, c_int_p],
        "kiwi_hip_get_misfits": [vp, C.c_int, C.c_int, c_float_p, c_float_p, c_float_p],
        "kiwi_hip_get_global_misfits_device": [vp, C.c_int, C.c_int, C.POINTER(C.c_void_p)],
        "kiwi_hip_get_synthetics": [vp, C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_float_p, C.c_int],
        "kiwi_hip_get_source_centroids": [vp, C.c_int, C.c_int, c_int_p, c_float_p],
        "kiwi_hip_get_reference": [vp, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_float_p, C.c_int],
        "kiwi_hip_get_kernel_ms": [vp, c_float_p, c_int_p],
        "kiwi_hip_get_geometry": [vp, C.c_int, C.c_int, C.c_int, c_int_p, vp],
        "kiwi_hip_get_receiver_geometry": [vp, C.c_int, c_double_p, c_double_p, c_double_p],
        "kiwi_hip_get_device_bytes": [vp, C.POINTER(C.c_longlong)],
        "kiwi_hip_measure_read_bandwidth": [vp, C.c_longlong, C.c_int, C.POINTER(C.c_double)],
        "kiwi_hip_build_flags": [C.c_char_p, C.c_int],
        "kiwi_hip_get_amp_spectrum": [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, c_int_p, c_float_p, C.c_int],
        "kiwi_hip_get_cross_correlations": [vp, C.c_int, C.c_int, C.c_float, C.c_float, c_int_p, c_int_p, c_float_p, C.c_int],
        "kiwi_hip_get_peak_amplitudes": [vp, C.c_int, C.c_int, c_float_p],
        "kiwi_hip_get_arias_intensities": [vp, C.c_int, c_float_p],
        "kiwi_hip_minimize_lm": [vp, C.c_int, c_float_p, c_int_p, c_float_p, c_float_p, c_int_p, c_int_p, c_float_p, c_float_p],
        "kiwi_hip_lmdif": [RESIDUAL_FN, vp, C.c_int, C.c_int, c_float_p, c_float_p, C.c_float, C.c_float, C.c_float, C.c_int,
                           C.c_float, c_float_p, C.c_int, C.c_float, c_int_p, c_int_p],
    }
    L.kiwi_hip_principal_axes.argtypes = [C.c_int, c_float_p, c_float_p, c_float_p]
    L.kiwi_hip_principal_axes.restype = C.c_int
    for name, argtypes in sig.items():
        f = getattr(L, name)
        f.argtypes = argtypes
        f.restype = C.c_int
    _lib = L
    return L
