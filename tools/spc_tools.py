"""Loader of tools/libspc_tools.so (measurement instruments: streaming copies, bandwidth probe; tools/csrc/spc_tools.h).
Not part of the product: only bench.py's copy-rate yardstick and the scripts in tools/ use it."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspc_tools.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not built: python -c 'import __graft_entry__ as g; g.build()'" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        lib.spc_tools_last_error.restype = ctypes.c_char_p
        lib.spc_stream_copy.argtypes = [vp, vp, i64, vp]
        lib.spc_stream_copy_f64.argtypes = [vp, vp, i64, vp]
        lib.spc_stream_copy_f32.argtypes = [vp, vp, i64, ctypes.c_int, vp]
        lib.spc_stream_probe.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, i64, ctypes.c_int, vp]
        lib.spc_probe_add_chain.argtypes = [vp, vp, ctypes.c_int, vp]
        lib.spc_probe_issue.argtypes = [ctypes.c_int, vp, vp, ctypes.c_int, vp]
        _lib = lib
    return _lib


def _check(rc):
    if rc:
        raise RuntimeError("spc_tools error %d: %s" % (rc, load().spc_tools_last_error().decode()))


def stream_copy(dst, src, stream=None, f64=False):
    """dst <- src (two contiguous torch tensors of equal byte size on one device) with the 16 B/lane (or 8 B/lane)
    streaming kernel, on ``stream`` (default: torch's current stream of that device)."""
    import torch
    nbytes = src.numel() * src.element_size()
    if dst.numel() * dst.element_size() != nbytes or not (src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("stream_copy needs two contiguous tensors of equal byte size")
    if stream is None:
        stream = torch.cuda.current_stream(src.device)
    fn = load().spc_stream_copy_f64 if f64 else load().spc_stream_copy
    with torch.cuda.device(src.device):
        _check(fn(dst.data_ptr(), src.data_ptr(), nbytes, ctypes.c_void_p(stream.cuda_stream)))
