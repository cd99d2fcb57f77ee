"""ctypes binding of libbas_hip.so (the C ABI declared in include/bas.h).

There is no CPU fallback: if the library is missing or a call fails, an exception
is raised.  Device memory, streams and process groups come from PyTorch-ROCm; the
arithmetic is entirely in the HIP library.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BAS_LIB_PATH: load another build of the same ABI instead (profiling tools use the diagnostic build this way)
LIB_PATH = os.environ.get("BAS_LIB_PATH") or os.path.join(_HERE, "csrc", "libbas_hip.so")

_c_int, _c_long, _c_size_t, _c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol of include/bas.h
SIGNATURES = {
    "bas_version": (_c_int, []),
    "bas_last_error": (ctypes.c_char_p, []),
    "bas_table_packed_floats": (_c_size_t, [_c_int, _c_int, _c_int]),
    "bas_table_pack_f32": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "bas_delay_signal_f32": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "bas_ring_interp_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                     _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "bas_traj_params_f64": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                     _c_void_p, _c_void_p, _c_void_p]),
    "bas_interp2d_workspace_bytes": (_c_size_t, [_c_int]),
    "bas_interp2d_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                  _c_int, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "bas_render_workspace_bytes": (_c_size_t, [_c_int, _c_long, _c_int, _c_int, _c_int]),
    "bas_render_kernel_name": (ctypes.c_char_p, [_c_int, _c_long, _c_int, _c_int, _c_int]),
    "bas_render_mix_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_int, _c_long, _c_int, _c_int, _c_int,
                                    _c_void_p, _c_int, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "bas_render_mix_profiled_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_int, _c_long, _c_int, _c_int,
                                             _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_size_t,
                                             _c_void_p, _c_void_p, _c_void_p]),
    "bas_interp2d_plan_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                                       _c_void_p, _c_size_t, _c_void_p]),
    "bas_render_fused_supported": (_c_int, [_c_int, _c_long, _c_int, _c_int, _c_int]),
    "bas_render_fused_workspace_bytes": (_c_size_t, [_c_int, _c_long, _c_int, _c_int, _c_int]),
    "bas_render_mix_fused_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_void_p, _c_int, _c_long, _c_int,
                                          _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p,
                                          _c_size_t, _c_void_p, _c_void_p, _c_void_p]),
    "bas_peak_normalize_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_int, _c_void_p]),
    "bas_scale_by_peak_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_void_p]),
    "bas_mix_partials_f32": (_c_int, [_c_void_p, _c_int, _c_long, _c_long, _c_void_p, _c_void_p, _c_void_p]),
}

_lib = None
ABI_VERSION = 2                                                      # BAS_ABI_VERSION of include/bas.h
DIAG_LIB_PATH = os.path.join(_HERE, "csrc", "libbas_hip_diag.so")   # -DBAS_DIAG build: reads BAS_FORCE_KERNEL (tests only)


class BasError(RuntimeError):
    def __init__(self, fn, code, text):
        super().__init__(f"{fn} failed with code {code}: {text}")
        self.code = code


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {os.path.dirname(path)}`.  There is no CPU fallback.")
    import torch                                # noqa: F401  first: PyTorch-ROCm brings its own libamdhip64, and the library
    handle = ctypes.CDLL(path)                  # must bind to THAT runtime (loaded the other way round a second HIP runtime
    for name, (res, args) in SIGNATURES.items():   # comes up beside torch's and sees no device)
        fn = getattr(handle, name)              # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if handle.bas_version() != ABI_VERSION:
        raise RuntimeError(f"{os.path.basename(path)} ABI version {handle.bas_version()} != {ABI_VERSION}")
    return handle


def lib():
    """The loaded library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


class use_library:
    """Context manager for tests and ablations: route every call of this process through another build of the
    same ABI (e.g. DIAG_LIB_PATH, the only build that honours BAS_FORCE_KERNEL) and restore the shipped one."""

    def __init__(self, path):
        self.handle = _load(path)

    def __enter__(self):
        global _lib
        self.saved, _lib = _lib, self.handle
        return self.handle

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


def call(name, *args):
    """Call an int-returning entry point; raise BasError on a non-zero code."""
    l = lib()
    rc = getattr(l, name)(*args)
    if rc != 0:
        raise BasError(name, rc, l.bas_last_error().decode(errors="replace"))


def ptr(t):
    """Device/host pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(device):
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(device=None):
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible to PyTorch-ROCm: this renderer has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
