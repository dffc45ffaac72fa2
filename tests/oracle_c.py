"""Test-side loader of the plain-C oracle (oracle/libspc_oracle.so). It takes the SAME ctypes
argument structs as the product ABI, but with HOST (NumPy) pointers."""
import ctypes
import os
import subprocess

import numpy

from sp_coupler_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "libspc_oracle.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_forward_f64.argtypes = [ctypes.POINTER(_abi.Dims), ctypes.POINTER(_abi.ForwardArgs)]
        _LIB.oracle_backward_f64.argtypes = [ctypes.POINTER(_abi.Dims), ctypes.POINTER(_abi.BackwardArgs)]
        _LIB.oracle_cloud_indices_f64.argtypes = [ctypes.POINTER(_abi.Dims)] + [ctypes.c_void_p] * 3
        _LIB.oracle_diagnostics_f64.argtypes = [ctypes.POINTER(_abi.Dims), ctypes.POINTER(_abi.DiagnosticsArgs)]
    return _LIB


def _p(a):
    assert a.flags.c_contiguous and a.dtype in (numpy.float64, numpy.int32)
    return a.ctypes.data


def forward(gcm, zf, zh, prof, factor, dt, couple_surface=True):
    n, nG = gcm["T"].shape
    nL = prof["U"].shape[1]
    a = _abi.ForwardArgs()
    for k, f in (("U", "U"), ("V", "V"), ("T", "T"), ("SH", "SH"), ("QL", "QL"), ("QI", "QI"), ("Pfull", "Pf"),
                 ("Phalf", "Ph"), ("Zgfull", "Zgfull"), ("Zghalf", "Zghalf")):
        setattr(a, f, _p(gcm[k]))
    for k, f in (("U", "u_d"), ("V", "v_d"), ("THL", "thl_d"), ("QT", "qt_d"), ("QL", "ql_d"), ("PS", "ps_d"),
                 ("Rain", "rain"), ("rain_last", "rain_last")):
        setattr(a, f, _p(prof[k]))
    a.zf, a.zh = _p(zf), _p(zh)
    out = {k: numpy.empty((n, nL)) for k in ("f_u", "f_v", "f_thl", "f_qt", "f_ql", "ql_ref", "u", "v", "thl", "qt")}
    out.update({k: numpy.empty(n) for k in ("f_ps", "ps", "rainrate")})
    out["Zf"], out["Zh"] = numpy.empty((n, nG)), numpy.empty((n, nG + 1))
    out["idx"] = numpy.empty((n, nG), dtype=numpy.int32)
    if couple_surface:
        for k in ("Z0M", "Z0H", "QLflux", "QIflux", "SHflux", "TSflux"):
            setattr(a, k, _p(gcm[k]))
        out.update({k: numpy.empty(n) for k in ("z0m", "z0h", "wthl", "wqt")})
    for k, v in out.items():
        setattr(a, k, _p(v))
    a.factor, a.dt = factor, dt
    d = _abi.Dims(n, nG, nL, nG, nG + 1, nL, 1 if zf.ndim == 1 else 0, 0)
    rc = lib().oracle_forward_f64(ctypes.byref(d), ctypes.byref(a))
    assert rc == 0, rc
    return out


def backward(gcm, Zf, zf, prof, factor, dt, conservative=False, zh=None, Zh=None):
    n, nG = gcm["T"].shape
    a = _abi.BackwardArgs()
    if conservative:
        a.conservative = 1
        a.zh, a.rhobf_d = _p(zh), _p(prof["Rhobf"])
        if Zh is not None:
            a.Zh = _p(Zh)
        else:
            a.Zghalf = _p(gcm["Zghalf"])
    for k in ("T", "SH", "QL", "QI", "U", "V", "A"):
        setattr(a, k, _p(gcm[k]))
    if Zf is not None:
        a.Zf = _p(Zf)
    else:
        a.Zgfull, a.Zghalf = _p(gcm["Zgfull"]), _p(gcm["Zghalf"])
    for k, f in (("T", "t_d"), ("QT", "qt_d"), ("QL", "ql_d"), ("QL_ice", "ql_ice_d"), ("U", "u_d"), ("V", "v_d"),
                 ("A", "A_prof")):
        setattr(a, f, _p(prof[k]))
    a.zf = _p(zf)
    out = {k: numpy.empty((n, nG)) for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A")}
    out["start_index"] = numpy.empty(n, dtype=numpy.int32)
    for k, v in out.items():
        setattr(a, k, _p(v))
    a.factor, a.dt = factor, dt
    nL = prof["T"].shape[1]
    d = _abi.Dims(n, nG, nL, nG, nG + 1, nL, 1 if zf.ndim == 1 else 0, 0)
    rc = lib().oracle_backward_f64(ctypes.byref(d), ctypes.byref(a))
    assert rc == 0, rc
    return out


def cloud_indices(zh, Zh):
    n, nG1 = Zh.shape
    nL = zh.shape[-1]
    idx = numpy.empty((n, nG1 - 1), dtype=numpy.int32)
    d = _abi.Dims(n, nG1 - 1, nL, nG1 - 1, nG1, nL, 1 if zh.ndim == 1 else 0, 0)
    rc = lib().oracle_cloud_indices_f64(ctypes.byref(d), _p(zh), _p(Zh), _p(idx))
    assert rc == 0, rc
    return idx
