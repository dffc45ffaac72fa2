"""ctypes mirror of include/spc.h (struct layouts and prototypes) and the library loader.

The product path has NO CPU fallback: ``load_library()`` raises ``SpcLibraryError`` when
``libspc_hip.so`` has not been built (``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes
import os

ABI_VERSION = 4
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libspc_hip.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

c_void_p, c_double, c_int32, c_int64 = ctypes.c_void_p, ctypes.c_double, ctypes.c_int32, ctypes.c_int64
c_int32_p = ctypes.POINTER(ctypes.c_int32)

SPC_OK, SPC_ERR_INVALID_ARGUMENT, SPC_ERR_UNSUPPORTED, SPC_ERR_LAUNCH, SPC_ERR_NO_DEVICE = 0, -1, -2, -3, -4


class SpcError(RuntimeError):
    """A C-ABI call returned a negative spc_status."""

    def __init__(self, code, text):
        super().__init__("spc error %d: %s" % (code, text))
        self.code = code


class SpcInvalidArgument(SpcError, ValueError):
    pass


class SpcLibraryError(ImportError):
    """The HIP extension is missing or does not match include/spc.h."""


class Dims(ctypes.Structure):
    _fields_ = [("n_cols", c_int64), ("nG", c_int32), ("nL", c_int32), ("pitchG", c_int64),
                ("pitchGh", c_int64), ("pitchL", c_int64), ("les_grid_shared", c_int32),
                ("cols_per_block", c_int32)]


def _ptrs(*names):
    return [(n, c_void_p) for n in names]


class ForwardArgs(ctypes.Structure):
    _fields_ = (_ptrs("U", "V", "T", "SH", "QL", "QI", "Pf", "Ph", "Zgfull", "Zghalf", "zf", "zh",
                      "u_d", "v_d", "thl_d", "qt_d", "ql_d", "ps_d", "rain", "rain_last")
                + [("factor", c_double), ("dt", c_double)]
                + _ptrs("f_u", "f_v", "f_thl", "f_qt", "f_ql", "ql_ref", "f_ps", "u", "v", "thl", "qt", "ps",
                        "Zf", "Zh", "rainrate", "idx", "Z0M", "Z0H", "QLflux", "QIflux", "SHflux", "TSflux",
                        "z0m", "z0h", "wthl", "wqt"))


class BackwardArgs(ctypes.Structure):
    _fields_ = (_ptrs("T", "SH", "QL", "QI", "U", "V", "A", "Zf", "Zgfull", "Zghalf", "zf",
                      "t_d", "qt_d", "ql_d", "ql_ice_d", "u_d", "v_d", "A_prof", "zh", "Zh", "rhobf_d")
                + [("conservative", c_int32), ("reserved", c_int32), ("factor", c_double), ("dt", c_double)]
                + _ptrs("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A", "start_index"))


class DiagnosticsArgs(ctypes.Structure):
    _fields_ = _ptrs("T", "SH", "QL", "QI", "Pf", "Zgfull", "Zghalf", "zf", "thl_d", "ql_d", "ql_ice_d",
                     "Tv", "THL", "QT", "Zf", "Zh", "pf", "t", "ql_water")


class VnudgeArgs(ctypes.Structure):
    _fields_ = ([("n_cols", c_int64), ("itot", c_int32), ("jtot", c_int32), ("ktot", c_int32), ("constantT", c_int32)]
                + _ptrs("qt", "qsat", "thl", "ql", "R", "ql_av", "qt_av", "presf", "ql_ref", "beta", "a_add", "qt_std", "status", "work")
                + [("work_bytes", c_int64)])


class InterpArgs(ctypes.Structure):
    _fields_ = ([("n_rows", c_int64), ("n_x", c_int32), ("n_xp", c_int32), ("pitch_x", c_int64), ("pitch_xp", c_int64),
                 ("pitch_fp", c_int64), ("pitch_out", c_int64)] + _ptrs("x", "xp", "fp", "out"))


class SearchsortedArgs(ctypes.Structure):
    _fields_ = ([("n_rows", c_int64), ("n_a", c_int32), ("n_v", c_int32), ("pitch_a", c_int64), ("pitch_v", c_int64),
                 ("pitch_out", c_int64)] + _ptrs("a", "v", "out") + [("side_right", c_int32), ("reserved", c_int32)])


class InterpCArgs(ctypes.Structure):
    _fields_ = ([("n_rows", c_int64), ("nG", c_int32), ("nL", c_int32), ("pitch_Zh", c_int64), ("pitch_zh", c_int64),
                 ("pitch_q", c_int64), ("pitch_out", c_int64)] + _ptrs("Zh", "zh", "q", "rho", "out")
                + [("mode", c_int32), ("reserved", c_int32)])


#: every symbol include/spc.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "spc_forward_f64": (ctypes.c_int, [ctypes.POINTER(Dims), ctypes.POINTER(ForwardArgs), c_void_p]),
    "spc_forward_f32": (ctypes.c_int, [ctypes.POINTER(Dims), ctypes.POINTER(ForwardArgs), c_void_p]),
    "spc_cloud_indices_f64": (ctypes.c_int, [ctypes.POINTER(Dims), c_void_p, c_void_p, c_void_p, c_void_p]),
    "spc_cloud_indices_f32": (ctypes.c_int, [ctypes.POINTER(Dims), c_void_p, c_void_p, c_void_p, c_void_p]),
    "spc_backward_f64": (ctypes.c_int, [ctypes.POINTER(Dims), ctypes.POINTER(BackwardArgs), c_void_p]),
    "spc_backward_f32": (ctypes.c_int, [ctypes.POINTER(Dims), ctypes.POINTER(BackwardArgs), c_void_p]),
    "spc_diagnostics_f64": (ctypes.c_int, [ctypes.POINTER(Dims), ctypes.POINTER(DiagnosticsArgs), c_void_p]),
    "spc_diagnostics_f32": (ctypes.c_int, [ctypes.POINTER(Dims), ctypes.POINTER(DiagnosticsArgs), c_void_p]),
    "spc_surface_fluxes_f64": (ctypes.c_int, [c_int64] + [c_void_p] * 9),
    "spc_surface_fluxes_f32": (ctypes.c_int, [c_int64] + [c_void_p] * 9),
    "spc_variability_nudge_f64": (ctypes.c_int, [ctypes.POINTER(VnudgeArgs), c_void_p]),
    "spc_exner_f64": (ctypes.c_int, [c_int64, c_void_p, c_void_p, c_int32, c_void_p]),
    "spc_exner_f32": (ctypes.c_int, [c_int64, c_void_p, c_void_p, c_int32, c_void_p]),
    "spc_interp_f64": (ctypes.c_int, [ctypes.POINTER(InterpArgs), c_void_p]),
    "spc_interp_f32": (ctypes.c_int, [ctypes.POINTER(InterpArgs), c_void_p]),
    "spc_searchsorted_f64": (ctypes.c_int, [ctypes.POINTER(SearchsortedArgs), c_void_p]),
    "spc_searchsorted_f32": (ctypes.c_int, [ctypes.POINTER(SearchsortedArgs), c_void_p]),
    "spc_interp_c_f64": (ctypes.c_int, [ctypes.POINTER(InterpCArgs), c_void_p]),
    "spc_interp_c_f32": (ctypes.c_int, [ctypes.POINTER(InterpCArgs), c_void_p]),
    "spc_rms_f64": (ctypes.c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "spc_rms_f32": (ctypes.c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "spc_abi_version": (ctypes.c_int, []),
    "spc_last_error": (ctypes.c_char_p, []),
    "spc_device_count": (ctypes.c_int, []),
    "spc_pick_cols_per_block": (ctypes.c_int, [ctypes.POINTER(Dims), ctypes.c_int]),
    "spc_describe_launch": (ctypes.c_int, [ctypes.POINTER(Dims), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                           ctypes.c_int]),
    "spc_vnudge_workspace_bytes": (c_int64, [c_int64, c_int32, c_int32, c_int32]),
}

_lib = None


def bind(lib, prototypes=PROTOTYPES):
    for name, (res, args) in prototypes.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SpcLibraryError("%s does not export %s (declared in include/spc.h)" % (lib._name, name)) from e
        fn.restype, fn.argtypes = res, args
    return lib


def load_library(path=None):
    """dlopen libspc_hip.so and bind every prototype. Raises SpcLibraryError if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("SPC_LIB") or LIB_PATH   # SPC_LIB: A/B a variant build of the same ABI
    if not os.path.exists(p):
        raise SpcLibraryError(
            "HIP extension %s not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback on the product path." % p)
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:
        raise SpcLibraryError("cannot load %s: %s" % (p, e)) from e
    bind(lib)
    v = lib.spc_abi_version()
    if v != ABI_VERSION:
        raise SpcLibraryError("%s has ABI version %d, python mirror expects %d" % (p, v, ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def describe_launch(lib, dims, pass_, flags=1, elem_size=8):
    """text of spc_describe_launch (include/spc.h): the kernel instantiation + launch shape chosen for ``dims``"""
    buf = ctypes.create_string_buffer(256)
    rc = lib.spc_describe_launch(ctypes.byref(dims), pass_, flags, elem_size, buf, len(buf))
    if rc < 0:
        check(lib, rc)
    return buf.value.decode()


def check(lib, rc):
    if rc == SPC_OK:
        return
    text = lib.spc_last_error().decode("utf-8", "replace")
    if rc == SPC_ERR_INVALID_ARGUMENT:
        raise SpcInvalidArgument(rc, text)
    raise SpcError(rc, text)
