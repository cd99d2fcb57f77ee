"""ctypes binding of libbas_hip.so (the C ABI declared in include/bas.h).

There is no CPU fallback: if the library is missing or a call fails, an exception
is raised.  Device memory, streams and process groups come from PyTorch-ROCm; the
arithmetic is entirely in the HIP library.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbas_hip.so")   # the shipped build; another build of the same ABI is loaded
                                                           # explicitly: set_library(path) / use_library(path), never by environment

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
    "bas_traj_params_branch_f64": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                            _c_void_p, _c_void_p, _c_int, _c_void_p]),
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
    "bas_interp2d_plan_angles_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p,
                                              _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p]),
    "bas_render_fused_supported": (_c_int, [_c_int, _c_long, _c_int, _c_int, _c_int]),
    "bas_render_fused_kernel_name": (ctypes.c_char_p, [_c_int, _c_long, _c_int, _c_int, _c_int]),
    "bas_render_fused_workspace_bytes": (_c_size_t, [_c_int, _c_long, _c_int, _c_int, _c_int]),
    "bas_render_mix_fused_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_void_p, _c_int, _c_long, _c_int,
                                          _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p,
                                          _c_size_t, _c_void_p]),
    "bas_render_mix_fused_profiled_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_void_p, _c_int, _c_long, _c_int,
                                          _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p,
                                          _c_size_t, _c_void_p, _c_void_p, _c_void_p]),
    "bas_render_fused_fir_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_void_p, _c_int, _c_long, _c_int,
                                          _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p,
                                          _c_size_t, _c_void_p]),
    "bas_render_fused_reduce_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_void_p, _c_int, _c_long, _c_int,
                                          _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p,
                                          _c_size_t, _c_void_p]),
    "bas_render_status": (_c_int, [_c_void_p, _c_size_t, _c_void_p]),
    "bas_peak_normalize_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_int, _c_void_p]),
    "bas_scale_by_peak_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_void_p]),
    "bas_mix_partials_f32": (_c_int, [_c_void_p, _c_int, _c_long, _c_long, _c_void_p, _c_void_p, _c_void_p]),
    "bas_mix_workspace_bytes": (_c_size_t, []),
    "bas_mix_finish_f32": (_c_int, [_c_void_p, _c_int, _c_long, _c_long, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_size_t,
                                    _c_void_p]),
    "bas_stream_epilogue_f32": (_c_int, [_c_void_p, _c_long, _c_int, _c_int, _c_long, _c_void_p, _c_void_p, _c_long,
                                         _c_int, _c_int, _c_void_p, _c_void_p, _c_long, _c_void_p, _c_void_p]),
    "bas_resample_up_f64": (_c_int, [_c_void_p, _c_int, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p]),
    "bas_delaydiffs_f64": (_c_int, [_c_void_p, _c_int, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "bas_render_stream_block_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_void_p, _c_int, _c_long, _c_int, _c_int,
                                             _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_size_t, _c_int, _c_void_p,
                                             _c_void_p, _c_long, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "bas_render_stream_block_profiled_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_void_p, _c_int, _c_long, _c_int,
                                                      _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_size_t,
                                                      _c_int, _c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p,
                                                      _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
}

_lib = None
ABI_VERSION = 5                                                      # BAS_ABI_VERSION of include/bas.h
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


def set_library(path):
    """Route every later call of this process through another build of the ABI (profiling tools: the diagnostic or
    the stamps build; `bench.py --lib`).  Returns the handle."""
    global _lib
    _lib = _load(path)
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


class on_device:
    """Make `device` the current HIP device for the calls inside: the library sizes its launches for the CURRENT
    device (CU count, LDS limits) and launches on the stream it is given, so the two must agree even when the
    caller's current device is another GPU.  No-op (no HIP call) when it already is."""

    def __init__(self, device):
        import torch
        self.ctx = None
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
            self.ctx = torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def on_device_of(argname_or_index):
    """Decorator: run the function with the device of its tensor argument (by position or keyword name; objects with a
    `.device` attribute such as the device table count) as the current device."""
    import functools

    def deco(fn):
        import inspect
        names = list(inspect.signature(fn).parameters)
        pos = names.index(argname_or_index) if isinstance(argname_or_index, str) else argname_or_index

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            obj = args[pos] if pos < len(args) else kwargs.get(names[pos])
            dev = getattr(obj, "device", None)
            if dev is None or getattr(dev, "type", "cuda") != "cuda":
                return fn(*args, **kwargs)
            with on_device(dev):
                return fn(*args, **kwargs)
        return wrapper
    return deco


WS_CONTROL_BYTES = 2048                                   # BAS_WS_CONTROL_BYTES of include/bas.h


def new_workspace(nbytes, device):
    """Scratch for the render / mix entry points (include/bas.h): uninitialised device bytes whose first 2048 - the
    library's control block (arrival counters of the kernel tails, device-side error record) - are zeroed once, here; every
    call leaves the counters zero.  One workspace serves one stream at a time."""
    import torch
    ws = torch.empty((max(int(nbytes), WS_CONTROL_BYTES),), dtype=torch.uint8, device=device)
    ws[:WS_CONTROL_BYTES].zero_()
    return ws


def check_status(ws, device):
    """Raise BasError if a kernel that used workspace `ws` recorded a device-side error (bas_render_status: synchronises
    the current stream of `device`; call where the caller synchronises anyway)."""
    with on_device(device):
        call("bas_render_status", ptr(ws), ws.numel(), current_stream(device))


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
