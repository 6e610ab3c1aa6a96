"""ctypes binding of libunflow_hip.so (include/unflow_hip.h).  Fails loudly when the HIP library is
missing: there is NO CPU fallback in the product path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UNFLOW_LIB_PATH") or os.path.join(_HERE, "csrc", "libunflow_hip.so")   # override: A/B builds

STATUS_TEXT = {
    -1: "null pointer argument",
    -2: "Invalid correlation settings",                      # correlation_op.cc:60-61
    -3: "kernel_size must be odd",                           # correlation_op.h:16-17
    -4: "Input height and width must be divisible by scale",  # downsample_op.cc:37-40
    -5: "Input shapes have to be the same",                  # correlation_op.cc:47-48
    -7: "unsupported configuration",
    -8: "HIP launch failed",
    -9: "workspace too small",
}


class UnflowError(ValueError):
    def __init__(self, status, where=""):
        self.status = status
        super().__init__("%s%s (status %d)" % (where + ": " if where else "", STATUS_TEXT.get(status, "error"), status))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libunflow_hip.so not found at %s — build it with `python -m unflow_amd.build` "
                "(or __graft_entry__.build()); there is no CPU fallback." % LIB_PATH)
        # torch must be imported first: libunflow_hip.so depends on libamdhip64 by SONAME, and it has to bind
        # to the ONE HIP runtime torch already loaded (its streams / device pointers are that runtime's).
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.unflow_status_string.restype = ctypes.c_char_p
        _lib.unflow_correlation_workspace_bytes.restype = ctypes.c_size_t
        _lib.unflow_conv_workspace_bytes.restype = ctypes.c_size_t
        _lib.unflow_conv_pl_workspace_bytes.restype = ctypes.c_size_t
        _lib.unflow_weight_planes_elems.restype = ctypes.c_size_t
        _lib.unflow_forward_warp_workspace_bytes.restype = ctypes.c_size_t
        _lib.unflow_option_names.restype = ctypes.c_char_p
        _apply_env_options(_lib)
    return _lib


def option_names():
    return [n for n in lib().unflow_option_names().decode().split("\n") if n]


def set_option(name, value):
    """unflow_set_option (csrc/options.h): kernel-selection / split-planning switches of the library.  The library never
    reads the environment; this and _apply_env_options below are the only writers."""
    check(lib().unflow_set_option(name.encode(), int(value)), "set_option(%s)" % name)


def get_option(name):
    v = ctypes.c_int(0)
    check(lib().unflow_get_option(name.encode(), ctypes.byref(v)), "get_option(%s)" % name)
    return v.value


def _apply_env_options(L):
    """Host-side bridge from the process environment to the library's option table, applied once at load time:
    UNFLOW_CONV_MATH / UNFLOW_WGRAD_MATH / UNFLOW_CORR_MATH = fp32 select the fp32-MFMA kernels (core/engine.py reads
    UNFLOW_CONV_MATH too, for the operand-plane layout), and UNFLOW_OPT_<NAME>=<int> sets any option of csrc/options.h
    (A/B runs of tools/ and profiles/)."""
    if os.environ.get("UNFLOW_CONV_MATH") == "fp32":
        L.unflow_set_option(b"conv_math_fp32", 1)
    if os.environ.get("UNFLOW_WGRAD_MATH") == "fp32":
        L.unflow_set_option(b"wgrad_math_fp32", 1)
    if os.environ.get("UNFLOW_CORR_MATH") == "fp32":
        L.unflow_set_option(b"corr_math_fp32", 1)
    for name in L.unflow_option_names().decode().split("\n"):
        v = os.environ.get("UNFLOW_OPT_" + name.upper()) if name else None
        if v is not None:
            if L.unflow_set_option(name.encode(), int(v)) != 0:
                raise RuntimeError("unflow_set_option(%s) failed" % name)


def check(status, where=""):
    if status != 0:
        raise UnflowError(status, where)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def cf(x):
    return ctypes.c_float(float(x))


def cl(x):
    return ctypes.c_long(int(x))


def csz(x):
    return ctypes.c_size_t(int(x))


def stream(device=None):
    """The current HIP stream of `device` (default: the current device) — pass the tensors' device when it may differ."""
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Planes(ctypes.Structure):
    """unflow_planes of include/unflow_hip.h: 16-bit operand planes of a tensor / channel slice."""
    _fields_ = [('base', ctypes.c_void_p), ('plane_stride', ctypes.c_long), ('ld', ctypes.c_int), ('n_planes', ctypes.c_int),
                ('scale', ctypes.c_float)]


def planes_of(pl, scale=0.0):
    """ctypes pointer to the unflow_planes of an int16 planes view [P, N, H, W, C] (or [P, ..., C] weights), or NULL.
    scale: what the planes hold relative to the tensor's values (0 = 1; fp16 gradient planes carry a power of two)."""
    if pl is None:
        return None
    return ctypes.byref(Planes(pl.data_ptr(), pl.stride(0), pl.stride(-2), pl.shape[0], float(scale)))


class PyrLevel(ctypes.Structure):
    """unflow_pyr_level of include/unflow_hip.h (argument of unflow_loss_pyramid_default)."""
    _fields_ = [(k, ctypes.c_void_p) for k in ('im', 'flow', 'gray1', 'gray2w', 'mask', 'dist', 'd_flow')] + \
               [(k, ctypes.c_int) for k in ('H', 'W', 'n_mask', 'max_distance')] + \
               [(k, ctypes.c_float) for k in ('flow_scale', 'ternary_scale', 'smooth_scale')]
