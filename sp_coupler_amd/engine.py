"""Host launcher for the HIP coupling kernels: torch tensors in HBM -> C ABI (include/spc.h).

PyTorch is plumbing here (device memory, streams); every number is produced by the hand-written
kernels in csrc/spc_hip.hip.  There is no CPU fallback: constructing an ``Engine`` without the built
extension or without a GPU raises.

A *plan* (``ForwardPlan`` / ``BackwardPlan``) validates shapes once, freezes the ctypes argument block
and can then be launched repeatedly with one foreign call -- what the per-step driver and ``bench.py``
use, so host overhead per step stays at two launches.
"""
import contextlib
import ctypes

import torch

from . import _abi

SURF_IN = ("Z0M", "Z0H", "QLflux", "QIflux", "SHflux", "TSflux")      # [n]        (spcpl.py:33,138)
FWD_LES = ("U", "V", "THL", "QT", "QL")                               # profile[...] spcpl.py:310-314
BWD_LES = ("T", "QT", "QL", "QL_ice", "U", "V")                       # profile[...] spcpl.py:393-411

_DTYPES = {torch.float64: "f64", torch.float32: "f32"}


def _stream_ptr(stream, device=None):
    """hipStream_t of ``stream`` (default: torch's current stream ON ``device``, not on whatever device
    happens to be current) as a ctypes handle; a stream of another device is rejected."""
    if stream is None:
        stream = torch.cuda.current_stream(device)
    elif device is not None and stream.device != device:
        raise ValueError("stream is on %s, engine is on %s" % (stream.device, device))
    return ctypes.c_void_p(stream.cuda_stream)


def _on_engine_stream(fn):
    """run an Engine method with torch's current stream set to the stream its kernels launch on (``stream=`` argument, else
    the engine's own): tensors it allocates belong to that stream, torch ops it issues (``.contiguous()``) are ordered
    with the launch.  Nothing changes for an engine without a stream called without one."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        st = kw.get("stream")
        if st is None:
            st = self.stream
        if st is None:
            return fn(self, *a, **kw)
        with torch.cuda.stream(st):
            return fn(self, *a, **kw)
    return wrapper


class _Checker:
    """Shape / layout validation done on the host BEFORE any launch (a kernel never sees a tensor
    whose extent differs from what its grid assumes)."""

    def __init__(self, device, dtype):
        self.device, self.dtype = device, dtype
        self.keep = []

    def mat(self, name, t, n, m, pitch=None):
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s must be a torch.Tensor, got %s" % (name, type(t).__name__))
        if t.device != self.device:
            raise ValueError("%s is on %s, engine is on %s" % (name, t.device, self.device))
        if t.dtype != self.dtype:
            raise ValueError("%s has dtype %s, engine computes in %s" % (name, t.dtype, self.dtype))
        if t.dim() != 2 or t.shape[0] != n or t.shape[1] != m:
            raise ValueError("%s must have shape [%d x %d], got %s" % (name, n, m, tuple(t.shape)))
        if m > 1 and t.stride(1) != 1:
            raise ValueError("%s must be contiguous along levels (stride %s)" % (name, t.stride()))
        p = t.stride(0) if n > 1 else (pitch if pitch is not None else max(m, t.stride(0)))
        if pitch is not None and n > 1 and p != pitch:
            raise ValueError("%s has column pitch %d, batch uses %d" % (name, p, pitch))
        if p < m:
            raise ValueError("%s has column pitch %d < %d levels" % (name, p, m))
        self.keep.append(t)
        return t.data_ptr(), p

    def vec(self, name, t, n, dtype=None):
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s must be a torch.Tensor, got %s" % (name, type(t).__name__))
        if t.device != self.device or t.dtype != (dtype or self.dtype):
            raise ValueError("%s must be %s on %s" % (name, dtype or self.dtype, self.device))
        if t.dim() != 1 or t.shape[0] != n or (n > 1 and t.stride(0) != 1):
            raise ValueError("%s must be a contiguous vector of %d, got %s" % (name, n, tuple(t.shape)))
        self.keep.append(t)
        return t.data_ptr()


class _Plan:
    """A validated, frozen launch: shapes checked once, ctypes argument block built once, tensors kept alive; ``launch``
    is then ONE foreign call.  ``fn(*call_args, stream)`` is the C-ABI entry point."""

    def __init__(self, engine, fn, call_args, keep, outputs, dims=None, args=None):
        self.engine, self._fn, self._call, self._keep, self.outputs = engine, fn, tuple(call_args), keep, outputs
        self.dims, self.args = dims, args

    def set_scalars(self, factor, dt):
        """forcing factor and time step of the next launches (they change between spin-up and coupled steps)"""
        self.args.factor, self.args.dt = float(factor), float(dt)

    def launch(self, stream=None):
        """Enqueue the kernel on ``stream`` (default: the engine's own stream if it has one, else torch's current
        stream of the engine's device). Returns outputs dict."""
        eng = self.engine
        if stream is None:
            stream = eng.stream
        with torch.cuda.device(eng.device):          # occupancy queries + launch on the engine's device
            rc = self._fn(*self._call, _stream_ptr(stream, eng.device))
        if rc:
            _abi.check(eng.lib, rc)
        return self.outputs

    def launch_raw(self, stream_ptr):
        """Same with a pre-fetched ``ctypes.c_void_p`` stream handle (hot loops). The caller guarantees that the
        handle belongs to the engine's device and that this device is current (``torch.cuda.set_device``)."""
        rc = self._fn(*self._call, stream_ptr)
        if rc:
            _abi.check(self.engine.lib, rc)

    def describe(self):
        """which kernel instantiation / slab size / grid the library picks for this plan (spc_describe_launch)"""
        return _abi.describe_launch(self.engine.lib, self.dims, self._pass, self._flags(), 8 if self.engine.dtype == torch.float64 else 4)


class ForwardPlan(_Plan):
    _pass = 0

    def _flags(self):
        a = self.args
        full = any(getattr(a, f) for f in ("u", "v", "thl", "qt", "ps", "Zf", "Zh", "rainrate", "wthl"))
        return (1 if a.idx else 0) | (2 if full else 0)


class BackwardPlan(_Plan):
    @property
    def _pass(self):
        return 4 if self.args.conservative else 1

    def _flags(self):
        return 0


class DiagnosticsPlan(_Plan):
    _pass = 3

    def _flags(self):
        return 0


class CloudIndexPlan(_Plan):
    _pass = 2

    def _flags(self):
        return 0


class SurfacePlan(_Plan):
    def describe(self):
        return "k_surface"


class OperatorPlan(_Plan):
    """a K7 operator (exner / interp / searchsorted / interp_c / rms) with its arguments frozen: ``run()`` launches and
    returns the result tensor (the caller's ``out=`` or the one allocated when the plan was made)"""

    def __init__(self, engine, fn, call_args, keep, outputs, result, refresh=()):
        super().__init__(engine, fn, call_args, keep, outputs)
        self.result = result
        # (private contiguous copy, the caller's tensor) of every argument whose layout the kernels cannot read in place: the
        # copy is REDONE in front of every launch, so a plan never computes from a stale snapshot of a tensor the caller
        # has updated since (round-4 advisor).  Empty for row-contiguous arguments: run() is then one foreign call.
        self._refresh = tuple(refresh)

    def run(self, stream=None):
        if self._refresh:
            eng = self.engine
            st = stream if stream is not None else eng.stream
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                for snap, src in self._refresh:
                    snap.copy_(src if src.shape == snap.shape else src.expand_as(snap))
        self.launch(stream)
        return self.result

    def launch_raw(self, stream_ptr):
        """the bare foreign call (hot loops) -- only for plans that read every argument in place: a plan holding private
        packed copies must go through run(), which refreshes them, and says so instead of serving a stale snapshot"""
        if self._refresh:
            raise RuntimeError("this plan reads %d argument(s) through private packed copies (non-contiguous or spread inputs): "
                               "use run(), which refreshes them before the launch" % len(self._refresh))
        super().launch_raw(stream_ptr)

    def describe(self):
        return getattr(self._fn, "__name__", "operator")


class Engine:
    """One engine per (device, dtype). ``dtype`` is the arithmetic type of the path: float64 as in
    the reference (AMUSE quantities wrap float64 arrays), float32 for the tolerance sweep."""

    def __init__(self, device=None, dtype=torch.float64, stream=None, lib_path=None):
        """``stream``: a ``torch.cuda.Stream`` every plan and every K7 operator of this engine launches on (None: torch's
        current stream of the device at launch time).  The engine's transfer buffers (``arena``) order their copies
        against the same stream, and the tensors the engine allocates are allocated under it (``on_stream``), so an
        engine with a stream of its own never depends on what stream is current in the caller.  Several such engines on
        ONE device run the row blocks of a ``multi.MultiDeviceEngine`` batch concurrently."""
        self.stream = stream
        # ``lib_path``: another build of the SAME ABI (A/B runs, tools/mutation_control.py); default: the shipped library
        self.lib = _abi.load_library(lib_path)  # raises SpcLibraryError if the HIP extension is missing
        if dtype not in _DTYPES:
            raise ValueError("dtype must be torch.float64 or torch.float32")
        if not torch.cuda.is_available() or self.lib.spc_device_count() < 1:
            raise RuntimeError("sp_coupler_amd.Engine needs a HIP device (MI355X); there is no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        if self.device.type != "cuda":
            raise ValueError("Engine device must be a cuda/HIP device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.dtype = dtype
        sfx = _DTYPES[dtype]
        self._vn_work = None                    # variability_nudge's transposed-plane scratch, grown on demand
        self._fwd = getattr(self.lib, "spc_forward_" + sfx)
        self._bwd = getattr(self.lib, "spc_backward_" + sfx)
        self._idx = getattr(self.lib, "spc_cloud_indices_" + sfx)
        self._diag = getattr(self.lib, "spc_diagnostics_" + sfx)

    # -- helpers ----------------------------------------------------------------------------
    def _grid(self, ck, name, z, n, nL, pitchL):
        """LES grid: [nL] shared by all columns or [n x nL]."""
        if z.dim() == 1:
            return ck.vec(name, z, nL), 1
        ptr, _ = ck.mat(name, z, n, nL, pitchL)
        return ptr, 0

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, device=self.device, dtype=dtype or self.dtype)

    def on_stream(self):
        """context in which torch's current stream IS this engine's stream (allocations, ``.to()`` / ``.cpu()`` copies and
        torch ops issued inside are then ordered with the engine's launches); a no-op for an engine without a stream"""
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def synchronize(self):
        (self.stream if self.stream is not None else torch.cuda.current_stream(self.device)).synchronize()

    # -- transfer plumbing shared with multi.MultiDeviceEngine (which shards the same things over several devices) ----
    def arena(self, specs, rows=None):
        """named arrays in ONE pinned host buffer mirrored by ONE device buffer (transfer.Arena)"""
        from .transfer import Arena
        return Arena(self.device, specs, stream=self.stream)

    def to_devices(self, host_array, rows=None, n_cols=None):
        import numpy
        with self.on_stream():
            return torch.from_numpy(numpy.ascontiguousarray(host_array)).to(self.device, self.dtype)

    # -- K1 (+K2) ---------------------------------------------------------------------------
    @_on_engine_stream
    def plan_forward(self, gcm, zf, prof, factor, dt, zh=None, *, want_profiles=False, want_heights=True,
                     want_idx=None, couple_surface=False, cols_per_block=0, out=None):
        """Arithmetic of convert_profiles + set_les_forcings (splib/spcpl.py:171-246, 299-385) for all
        columns. ``gcm``: dict of the gcm_vars tensors; ``prof``: dict U,V,THL,QT,QL [n x nL], PS [n]
        (+ Rain, rain_last [n]). Output tensors are allocated here (or taken from ``out``)."""
        T_ = gcm["T"]
        n, nG = int(T_.shape[0]), int(T_.shape[1])
        nL = int(prof["U"].shape[1])
        ck = _Checker(self.device, self.dtype)
        a = _abi.ForwardArgs()
        pitchG = pitchGh = pitchL = None
        for key, field in (("U", "U"), ("V", "V"), ("T", "T"), ("SH", "SH"), ("QL", "QL"), ("QI", "QI"),
                           ("Pfull", "Pf"), ("Zgfull", "Zgfull")):
            ptr, pitchG = ck.mat("gcm[%s]" % key, gcm[key], n, nG, pitchG)
            setattr(a, field, ptr)
        for key, field in (("Phalf", "Ph"), ("Zghalf", "Zghalf")):
            ptr, pitchGh = ck.mat("gcm[%s]" % key, gcm[key], n, nG + 1, pitchGh)
            setattr(a, field, ptr)
        for key, field in zip(FWD_LES, ("u_d", "v_d", "thl_d", "qt_d", "ql_d")):
            ptr, pitchL = ck.mat("prof[%s]" % key, prof[key], n, nL, pitchL)
            setattr(a, field, ptr)
        a.ps_d = ck.vec("prof[PS]", prof["PS"], n)
        a.zf, shared = self._grid(ck, "zf", zf, n, nL, pitchL)
        if want_idx is None:
            want_idx = zh is not None
        out = dict(out or {})
        res = {}

        def omat(name, m, pitch):
            t = out.get(name)
            if t is None:
                t = torch.empty(n, pitch or m, device=self.device, dtype=self.dtype)[:, :m]
            ptr, _ = ck.mat("out[%s]" % name, t, n, m, pitch)
            res[name] = t
            return ptr

        def ovec(name):
            t = out.get(name)
            if t is None:
                t = self.empty(n)
            res[name] = t
            return ck.vec("out[%s]" % name, t, n)

        for name in ("f_u", "f_v", "f_thl", "f_qt", "f_ql", "ql_ref"):
            setattr(a, name, omat(name, nL, pitchL))
        a.f_ps = ovec("f_ps")
        if want_profiles:
            for name in ("u", "v", "thl", "qt"):
                setattr(a, name, omat(name, nL, pitchL))
            a.ps = ovec("ps")
        if want_heights:
            a.Zf = omat("Zf", nG, pitchG)
            a.Zh = omat("Zh", nG + 1, pitchGh)
        if want_idx:
            if zh is None:
                raise ValueError("want_idx needs the LES half-level heights zh")
            a.zh, sh2 = self._grid(ck, "zh", zh, n, nL, pitchL)
            if sh2 != shared:
                raise ValueError("zf and zh must both be shared [nL] or both per column [n x nL]")
            idx = out.get("idx")
            if idx is None:  # idx shares the column pitch of the GCM full-level arrays
                idx = torch.empty(n, pitchG or nG, device=self.device, dtype=torch.int32)[:, :nG]
            if (idx.dtype != torch.int32 or idx.device != self.device or tuple(idx.shape) != (n, nG)
                    or (nG > 1 and idx.stride(1) != 1) or (n > 1 and idx.stride(0) != pitchG)):
                raise ValueError("out[idx] must be int32 [%d x %d] with column pitch %d" % (n, nG, pitchG))
            ck.keep.append(idx)
            a.idx = idx.data_ptr()
            res["idx"] = idx
        if "Rain" in prof and "rain_last" in prof:
            a.rain = ck.vec("prof[Rain]", prof["Rain"], n)
            a.rain_last = ck.vec("prof[rain_last]", prof["rain_last"], n)
            a.rainrate = ovec("rainrate")
        if couple_surface:
            for key in SURF_IN:
                setattr(a, key, ck.vec("gcm[%s]" % key, gcm[key], n))
            for name in ("z0m", "z0h", "wthl", "wqt"):
                setattr(a, name, ovec(name))
        a.factor, a.dt = float(factor), float(dt)
        dims = _abi.Dims(n, nG, nL, pitchG or nG, pitchGh or nG + 1, pitchL or nL, shared, int(cols_per_block))
        return ForwardPlan(self, self._fwd, (ctypes.byref(dims), ctypes.byref(a)), ck.keep, res, dims, a)

    @_on_engine_stream
    def forward(self, *args, stream=None, **kw):
        return self.plan_forward(*args, **kw).launch(stream)

    # -- K2 standalone ----------------------------------------------------------------------
    @_on_engine_stream
    def plan_cloud_indices(self, zh, Zh, out=None, cols_per_block=0):
        """searchsorted(zh, Zh, 'right')[:-1][::-1] per column (splib/spcpl.py:26, 764); ``out``: int32 [n x nG]."""
        n, nGp1 = int(Zh.shape[0]), int(Zh.shape[1])
        nG = nGp1 - 1
        nL = int(zh.shape[-1])
        ck = _Checker(self.device, self.dtype)
        Zh_ptr, pitchGh = ck.mat("Zh", Zh, n, nG + 1)
        zh_ptr, shared = self._grid(ck, "zh", zh, n, nL, None)
        pitchL = zh.stride(0) if (zh.dim() == 2 and n > 1) else nL
        idx = out if out is not None else self.empty(n, nG, dtype=torch.int32)
        if (idx.dtype != torch.int32 or idx.device != self.device or tuple(idx.shape) != (n, nG)
                or (nG > 1 and idx.stride(1) != 1)):
            raise ValueError("out must be int32 [%d x %d] on %s, contiguous along levels" % (n, nG, self.device))
        pitchI = idx.stride(0) if n > 1 else nG
        ck.keep.append(idx)
        dims = _abi.Dims(n, nG, nL, pitchI, pitchGh, pitchL, shared, int(cols_per_block))
        return CloudIndexPlan(self, self._idx, (ctypes.byref(dims), zh_ptr, Zh_ptr, idx.data_ptr()), ck.keep, {"idx": idx}, dims)

    @_on_engine_stream
    def cloud_indices(self, zh, Zh, stream=None, cols_per_block=0):
        return self.plan_cloud_indices(zh, Zh, cols_per_block=cols_per_block).launch(stream)["idx"]

    # -- K3 ---------------------------------------------------------------------------------
    @_on_engine_stream
    def plan_backward(self, gcm, zf, prof, factor, dt, Zf=None, *, want_start_index=True, conservative=False,
                      zh=None, Zh=None, cols_per_block=0, out=None):
        """Arithmetic of set_gcm_tendencies (splib/spcpl.py:388-555) for all columns. ``prof``: dict
        T,QT,QL,QL_ice,U,V [n x nL] and A [n x nG] (order of get_cloudfraction(indices))."""
        T_ = gcm["T"]
        n, nG = int(T_.shape[0]), int(T_.shape[1])
        nL = int(prof["T"].shape[1])
        ck = _Checker(self.device, self.dtype)
        a = _abi.BackwardArgs()
        pitchG = pitchGh = pitchL = None
        for key in ("T", "SH", "QL", "QI", "U", "V", "A"):
            ptr, pitchG = ck.mat("gcm[%s]" % key, gcm[key], n, nG, pitchG)
            setattr(a, key, ptr)
        if Zf is not None:
            a.Zf, pitchG = ck.mat("Zf", Zf, n, nG, pitchG)
        else:
            a.Zgfull, pitchG = ck.mat("gcm[Zgfull]", gcm["Zgfull"], n, nG, pitchG)
            a.Zghalf, pitchGh = ck.mat("gcm[Zghalf]", gcm["Zghalf"], n, nG + 1, pitchGh)
        for key, field in zip(BWD_LES, ("t_d", "qt_d", "ql_d", "ql_ice_d", "u_d", "v_d")):
            ptr, pitchL = ck.mat("prof[%s]" % key, prof[key], n, nL, pitchL)
            setattr(a, field, ptr)
        a.A_prof, pitchG = ck.mat("prof[A]", prof["A"], n, nG, pitchG)
        a.zf, shared = self._grid(ck, "zf", zf, n, nL, pitchL)
        a.conservative = 1 if conservative else 0
        if conservative:   # sputils.interp_c needs the half levels of both grids and the LES base density
            if zh is None:
                raise ValueError("conservative coarsening needs the LES half-level heights zh")
            a.zh, sh2 = self._grid(ck, "zh", zh, n, nL, pitchL)
            if sh2 != shared:
                raise ValueError("zf and zh must both be shared [nL] or both per column [n x nL]")
            a.rhobf_d, pitchL = ck.mat("prof[Rhobf]", prof["Rhobf"], n, nL, pitchL)
            if Zh is not None:
                a.Zh, pitchGh = ck.mat("Zh", Zh, n, nG + 1, pitchGh)
            else:
                a.Zghalf, pitchGh = ck.mat("gcm[Zghalf]", gcm["Zghalf"], n, nG + 1, pitchGh)
        a.factor, a.dt = float(factor), float(dt)
        out = dict(out or {})
        res = {}
        for name in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A"):
            t = out.get(name)
            if t is None:   # outputs share the column pitch of the GCM full-level inputs
                t = torch.empty(n, pitchG or nG, device=self.device, dtype=self.dtype)[:, :nG]
            ptr, pitchG = ck.mat("out[%s]" % name, t, n, nG, pitchG)
            setattr(a, name, ptr)
            res[name] = t
        if want_start_index:
            t = out.get("start_index")
            if t is None:
                t = self.empty(n, dtype=torch.int32)
            a.start_index = ck.vec("out[start_index]", t, n, dtype=torch.int32)
            res["start_index"] = t
        dims = _abi.Dims(n, nG, nL, pitchG or nG, pitchGh or nG + 1, pitchL or nL, shared, int(cols_per_block))
        return BackwardPlan(self, self._bwd, (ctypes.byref(dims), ctypes.byref(a)), ck.keep, res, dims, a)

    @_on_engine_stream
    def backward(self, *args, stream=None, **kw):
        return self.plan_backward(*args, **kw).launch(stream)

    # -- the hot-path pair: what a steady-state step launches, and what bench.py / tools time -------
    def plan_exchange(self, gcm, zf, zh, prof, factor_les, factor_gcm, dt, cols_per_block=0):
        """(ForwardPlan, BackwardPlan) of one column-exchange with ONLY the outputs SURVEY 8(d) counts: K1 writes
        the six setter arrays + f_ps + the fused index map (no optional profiles / heights / rain rate), K3
        recomputes Zf from the geopotential (no Zf round trip) and writes the seven tendencies."""
        lean = {k: v for k, v in prof.items() if k not in ("Rain", "rain_last")}
        fp = self.plan_forward(gcm, zf, lean, factor_les, dt, zh=zh, want_profiles=False, want_heights=False,
                               cols_per_block=cols_per_block)
        bp = self.plan_backward(gcm, zf, prof, factor_gcm, dt, Zf=None, want_start_index=False,
                                cols_per_block=cols_per_block)
        return fp, bp

    # -- K5 ---------------------------------------------------------------------------------
    @_on_engine_stream
    def plan_diagnostics(self, gcm, zf=None, prof=None, out=None, cols_per_block=0):
        """spifs.nc diagnostics: Tv, THL, QT, Zf, Zh (splib/spcpl.py:176,197-198,214-215) and, when
        ``zf``/``prof`` are given, pf, t, ql_water on LES levels (splib/spcpl.py:402,408-409).  Output tensors are
        taken from ``out`` where given (views of a transfer buffer), else allocated ONCE here."""
        T_ = gcm["T"]
        n, nG = int(T_.shape[0]), int(T_.shape[1])
        ck = _Checker(self.device, self.dtype)
        a = _abi.DiagnosticsArgs()
        out = dict(out or {})
        pitchG = pitchGh = pitchL = None
        for key, field in (("T", "T"), ("SH", "SH"), ("QL", "QL"), ("QI", "QI"), ("Pfull", "Pf"), ("Zgfull", "Zgfull")):
            ptr, pitchG = ck.mat("gcm[%s]" % key, gcm[key], n, nG, pitchG)
            setattr(a, field, ptr)
        a.Zghalf, pitchGh = ck.mat("gcm[Zghalf]", gcm["Zghalf"], n, nG + 1, pitchGh)
        res = {}
        for name in ("Tv", "THL", "QT", "Zf"):
            t = out.get(name)
            res[name] = t if t is not None else torch.empty(n, pitchG or nG, device=self.device, dtype=self.dtype)[:, :nG]
            ptr, pitchG = ck.mat(name, res[name], n, nG, pitchG)
            setattr(a, name, ptr)
        t = out.get("Zh")
        res["Zh"] = t if t is not None else torch.empty(n, pitchGh or nG + 1, device=self.device, dtype=self.dtype)[:, :nG + 1]
        a.Zh, pitchGh = ck.mat("Zh", res["Zh"], n, nG + 1, pitchGh)
        nL, shared = 1, 1
        if zf is not None and prof is not None:
            nL = int(prof["THL"].shape[1])
            for key, field in (("THL", "thl_d"), ("QL", "ql_d"), ("QL_ice", "ql_ice_d")):
                ptr, pitchL = ck.mat("prof[%s]" % key, prof[key], n, nL, pitchL)
                setattr(a, field, ptr)
            a.zf, shared = self._grid(ck, "zf", zf, n, nL, pitchL)
            for name in ("pf", "t", "ql_water"):
                t = out.get(name)
                res[name] = t if t is not None else torch.empty(n, pitchL or nL, device=self.device, dtype=self.dtype)[:, :nL]
                ptr, pitchL = ck.mat(name, res[name], n, nL, pitchL)
                setattr(a, name, ptr)
        dims = _abi.Dims(n, nG, nL, pitchG or nG, pitchGh or nG + 1, pitchL or nL, shared, int(cols_per_block))
        return DiagnosticsPlan(self, self._diag, (ctypes.byref(dims), ctypes.byref(a)), ck.keep, res, dims, a)

    @_on_engine_stream
    def diagnostics(self, gcm, zf=None, prof=None, stream=None, cols_per_block=0):
        return self.plan_diagnostics(gcm, zf, prof, cols_per_block=cols_per_block).launch(stream)

    # -- variability nudge (qt_forcing == 'variance') ------------------------------------------------
    @_on_engine_stream
    def variability_nudge(self, qt, qsat, R, ql_av, qt_av, ql_ref, presf=None, thl=None, ql=None, constantT=False,
                          stream=None):
        """spcpl.variability_nudge (splib/spcpl.py:613-744) for all columns: ``qt`` [n x itot x jtot x k] is updated
        IN PLACE (``thl`` too with ``constantT``); returns dict beta, a, qt_std [n x k] and status [n x k] int32."""
        if self.dtype != torch.float64:
            raise ValueError("variability_nudge computes in float64 only (bit parity with numpy / scipy)")
        if qt.dim() != 4:
            raise ValueError("qt must be [n x itot x jtot x ktot]")
        n, itot, jtot, ktot = (int(x) for x in qt.shape)
        ck = _Checker(self.device, self.dtype)

        def field(name, t):
            if not isinstance(t, torch.Tensor) or t.device != self.device or t.dtype != self.dtype or \
                    tuple(t.shape) != (n, itot, jtot, ktot) or not t.is_contiguous():
                raise ValueError("%s must be a contiguous float64 [%d x %d x %d x %d] tensor on %s" % (name, n, itot, jtot, ktot, self.device))
            ck.keep.append(t)
            return t.data_ptr()
        a = _abi.VnudgeArgs()
        a.n_cols, a.itot, a.jtot, a.ktot, a.constantT = n, itot, jtot, ktot, 1 if constantT else 0
        a.qt, a.qsat = field("qt", qt), field("qsat", qsat)
        a.R, _ = ck.mat("R", R.reshape(n, itot * jtot), n, itot * jtot)
        for name, t in (("ql_av", ql_av), ("qt_av", qt_av), ("ql_ref", ql_ref)):
            ptr, _ = ck.mat(name, t, n, ktot, ktot)
            setattr(a, name, ptr)
        if constantT:
            a.thl, a.ql = field("thl", thl), field("ql", ql)
            a.presf, _ = ck.mat("presf", presf, n, ktot, ktot)
        res = {k: self.empty(n, ktot) for k in ("beta", "a", "qt_std")}
        res["status"] = self.empty(n, ktot, dtype=torch.int32)
        a.beta, a.a_add, a.qt_std, a.status = (res["beta"].data_ptr(), res["a"].data_ptr(), res["qt_std"].data_ptr(),
                                               res["status"].data_ptr())
        # scratch for the transposed qt / qsat planes (include/spc.h: spc_vnudge_args.work), kept between calls and sized
        # by the library
        need = int(self.lib.spc_vnudge_workspace_bytes(n, itot, jtot, ktot))
        if need < 0:
            _abi.check(self.lib, need)
        if self._vn_work is None or self._vn_work.numel() < need:
            self._vn_work = None
            self._vn_work = torch.empty(need, dtype=torch.uint8, device=self.device)
        a.work, a.work_bytes = self._vn_work.data_ptr(), self._vn_work.numel()
        with torch.cuda.device(self.device):
            rc = self.lib.spc_variability_nudge_f64(ctypes.byref(a), _stream_ptr(stream if stream is not None else self.stream, self.device))
        _abi.check(self.lib, rc)
        return res

    # -- K7: the helpers of splib/sputils.py as batched operators (sp_coupler_amd/sputils.py keeps their names) -------
    # Each operator has a ``plan_*`` form (arguments checked and the C argument block frozen ONCE, output allocated once
    # or taken from ``out=``: ``plan.run()`` is then one foreign call, no allocation) and a convenience form that builds
    # the plan and runs it.  Arguments are read IN PLACE when they are contiguous along their rows; anything else (a
    # transposed view, a shared fp of interp, q / rho of different pitch) goes through a private packed copy that run()
    # refreshes from the caller's tensor before every launch -- a plan never serves a stale snapshot.
    def _rows(self, name, t, n_rows=None, shared_ok=False, refresh=None):
        """(tensor, data_ptr, pitch, n): a [n_rows x n] matrix contiguous along n (pitch = row stride), or -- where the
        operator allows it -- ONE [n] row shared by all rows (pitch 0).  A tensor the kernels cannot read in place is
        copied; the (copy, original) pair is appended to ``refresh`` so that a plan can redo the copy before every run."""
        orig = t
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s must be a torch.Tensor, got %s" % (name, type(t).__name__))
        if t.device != self.device or t.dtype != self.dtype:
            raise ValueError("%s must be %s on %s (got %s on %s)" % (name, self.dtype, self.device, t.dtype, t.device))
        if t.dim() == 1:
            if not shared_ok and n_rows not in (None, 1):
                raise ValueError("%s must have %d rows, got one" % (name, n_rows))
            t = t.contiguous()
            if refresh is not None and t is not orig:
                refresh.append((t, orig))
            return t, t.data_ptr(), (0 if shared_ok and n_rows not in (None, 1) else max(1, t.shape[0])), int(t.shape[0])
        if t.dim() != 2:
            raise ValueError("%s must be [n] or [n_rows x n], got %s" % (name, tuple(t.shape)))
        if n_rows is not None and t.shape[0] != n_rows:
            raise ValueError("%s must have %d rows, got %d" % (name, n_rows, t.shape[0]))
        if t.shape[1] > 1 and t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
            t = t.contiguous()
            if refresh is not None:
                refresh.append((t, orig))
        return t, t.data_ptr(), (int(t.stride(0)) if t.shape[0] > 1 else max(1, int(t.shape[1]))), int(t.shape[1])

    def _out(self, out, n_rows, n, dtype=None):
        """the operator's [n_rows x n] result: the caller's ``out`` (checked; may be [n] when n_rows == 1, may be pitched)
        or a fresh tensor.  Returns (2-D tensor, pitch)."""
        dtype = dtype or self.dtype
        if out is None:
            out = torch.empty(n_rows, n, dtype=dtype, device=self.device)
        else:
            if not isinstance(out, torch.Tensor) or out.device != self.device or out.dtype != dtype:
                raise ValueError("out must be a %s tensor on %s" % (dtype, self.device))
            if out.dim() == 1 and n_rows == 1:
                out = out.unsqueeze(0)
            if tuple(out.shape) != (n_rows, n) or (n > 1 and out.stride(1) != 1) or (n_rows > 1 and out.stride(0) < n):
                raise ValueError("out must be [%d x %d], contiguous along its rows, got %s (strides %s)"
                                 % (n_rows, n, tuple(out.shape), out.stride()))
        return out, (int(out.stride(0)) if n_rows > 1 else max(1, n))

    def _call(self, fn, *args, stream=None):
        with torch.cuda.device(self.device):
            rc = fn(*args, _stream_ptr(stream if stream is not None else self.stream, self.device))
        _abi.check(self.lib, rc)

    @_on_engine_stream
    def plan_exner(self, p, inverse=False, out=None):
        """sputils.exner / iexner (splib/sputils.py:28-34), elementwise on a device tensor of any shape"""
        if p.device != self.device or p.dtype != self.dtype:
            raise ValueError("p must be %s on %s" % (self.dtype, self.device))
        refresh = []
        if not p.is_contiguous():
            src, p = p, p.contiguous()
            refresh.append((p, src))
        if out is None:
            out = torch.empty_like(p)
        elif not isinstance(out, torch.Tensor) or out.device != self.device or out.dtype != self.dtype or out.shape != p.shape or not out.is_contiguous():
            raise ValueError("out must be a contiguous %s tensor of shape %s on %s" % (self.dtype, tuple(p.shape), self.device))
        fn = getattr(self.lib, "spc_exner_" + _DTYPES[self.dtype])
        return OperatorPlan(self, fn, (p.numel(), p.data_ptr(), out.data_ptr(), 1 if inverse else 0), [p, out], {"out": out}, result=out,
                            refresh=refresh)

    def exner(self, p, inverse=False, stream=None, out=None):
        return self.plan_exner(p, inverse, out=out).run(stream)

    @_on_engine_stream
    def plan_interp(self, x, xp, fp, out=None):
        """sputils.interp == numpy.interp (splib/sputils.py:82-86) for every row: fp [n_rows x n_xp] (or [n_xp]), xp the
        same shape or one shared [n_xp], x [n_rows x n_x] or one shared [n_x].  Result [n_rows x n_x] ([n_x] when every
        argument is 1-D)."""
        one = fp.dim() == 1 and xp.dim() == 1 and x.dim() == 1
        n_rows = max(int(t.shape[0]) if t.dim() == 2 else 1 for t in (x, xp, fp))
        refresh = []
        if fp.dim() == 1 and n_rows > 1:          # the ABI has one fp row per result row: a shared fp is spread (and re-spread per run)
            fp2 = fp.unsqueeze(0).expand(n_rows, -1).contiguous()
            refresh.append((fp2, fp.unsqueeze(0)))
        else:
            fp2 = fp
        fp2, p_fp, pitch_fp, n_xp = self._rows("fp", fp2, n_rows, refresh=refresh)
        xp2, p_xp, pitch_xp, n_xp2 = self._rows("xp", xp, n_rows, shared_ok=True, refresh=refresh)
        x2, p_x, pitch_x, n_x = self._rows("x", x, n_rows, shared_ok=True, refresh=refresh)
        if n_xp2 != n_xp:
            raise ValueError("fp and xp are not of the same length")          # numpy.interp's message
        out, pitch_out = self._out(out, n_rows, n_x)
        a = _abi.InterpArgs(n_rows, n_x, n_xp, pitch_x, pitch_xp, pitch_fp, pitch_out, p_x, p_xp, p_fp, out.data_ptr())
        fn = getattr(self.lib, "spc_interp_" + _DTYPES[self.dtype])
        return OperatorPlan(self, fn, (ctypes.byref(a),), [fp2, xp2, x2, out, a], {"out": out}, result=out[0] if one else out, refresh=refresh)

    def interp(self, x, xp, fp, stream=None, out=None):
        return self.plan_interp(x, xp, fp, out=out).run(stream)

    @_on_engine_stream
    def plan_searchsorted(self, a, v, side="left", out=None):
        """sputils.searchsorted == numpy.searchsorted (splib/sputils.py:88-91) per row; int64 indices"""
        if side not in ("left", "right"):
            raise ValueError("side must be 'left' or 'right'")
        one = a.dim() == 1 and v.dim() == 1
        n_rows = max(int(t.shape[0]) if t.dim() == 2 else 1 for t in (a, v))
        refresh = []
        a2, p_a, pitch_a, n_a = self._rows("a", a, n_rows, shared_ok=True, refresh=refresh)
        v2, p_v, pitch_v, n_v = self._rows("v", v, n_rows, shared_ok=True, refresh=refresh)
        out, pitch_out = self._out(out, n_rows, n_v, dtype=torch.int64)
        args = _abi.SearchsortedArgs(n_rows, n_a, n_v, pitch_a, pitch_v, pitch_out, p_a, p_v, out.data_ptr(), 1 if side == "right" else 0, 0)
        fn = getattr(self.lib, "spc_searchsorted_" + _DTYPES[self.dtype])
        return OperatorPlan(self, fn, (ctypes.byref(args),), [a2, v2, out, args], {"out": out}, result=out[0] if one else out, refresh=refresh)

    def searchsorted(self, a, v, side="left", stream=None, out=None):
        return self.plan_searchsorted(a, v, side, out=out).run(stream)

    @_on_engine_stream
    def plan_interp_c(self, Zh, zh, q, rho=None, mode="interp_c", out=None):
        """sputils.interp_c / interp_rho / integral (splib/sputils.py:94-197) per row: Zh [n_rows x (nG+1)] layer bounds, zh
        [nL] (shared) or [n_rows x nL] grid points, q (and rho) [n_rows x nL'] with nL' >= nL - 1 cell values.
        mode 'interp_c' (rho required), 'interp_rho' (q is the density), 'integral' (rho optional)."""
        modes = {"interp_c": 0, "interp_rho": 1, "integral": 2}
        if mode not in modes:
            raise ValueError("mode must be one of %s" % sorted(modes))
        one = Zh.dim() == 1 and q.dim() == 1
        n_rows = max(int(t.shape[0]) if t.dim() == 2 else 1 for t in (Zh, q))
        refresh = []
        Zh2, p_Zh, pitch_Zh, nGh = self._rows("Zh", Zh, n_rows, refresh=refresh)
        zh2, p_zh, pitch_zh, nL = self._rows("zh", zh, n_rows, shared_ok=True, refresh=refresh)
        q2, p_q, pitch_q, nq = self._rows("q", q, n_rows, refresh=refresh)
        if nq < nL - 1:
            raise ValueError("q has %d values, the %d grid points of zh bound %d cells" % (nq, nL, nL - 1))
        p_rho, rho2 = None, None
        if rho is not None and mode != "interp_rho":
            rho2, p_rho, pitch_rho, nr = self._rows("rho", rho, n_rows, refresh=refresh)
            if nr != nq:
                raise ValueError("rho and q must have the same shape")
            if n_rows > 1 and pitch_rho != pitch_q:            # the ABI has ONE pitch for q and rho: both packed (and re-packed per run)
                refresh[:] = [pr for pr in refresh if pr[0] is not rho2 and pr[0] is not q2]
                rho2, q2 = rho.contiguous(), q.contiguous()
                if rho2 is rho:
                    rho2 = rho.clone()
                if q2 is q:
                    q2 = q.clone()
                refresh += [(rho2, rho), (q2, q)]
                p_rho, p_q, pitch_q = rho2.data_ptr(), q2.data_ptr(), max(1, nq)
        elif mode == "interp_c":
            raise ValueError("interp_c needs the weights rho")
        nG = nGh - 1
        out, pitch_out = self._out(out, n_rows, max(nG, 0))
        a = _abi.InterpCArgs(n_rows, nG, nL, pitch_Zh, pitch_zh, pitch_q, pitch_out, p_Zh, p_zh, p_q, p_rho, out.data_ptr(), modes[mode], 0)
        fn = getattr(self.lib, "spc_interp_c_" + _DTYPES[self.dtype])
        return OperatorPlan(self, fn, (ctypes.byref(a),), [Zh2, zh2, q2, rho2, out, a], {"out": out}, result=out[0] if one else out, refresh=refresh)

    def interp_c(self, Zh, zh, q, rho=None, mode="interp_c", stream=None, out=None):
        return self.plan_interp_c(Zh, zh, q, rho, mode, out=out).run(stream)

    @_on_engine_stream
    def plan_rms(self, a, out=None):
        """sputils.rms (splib/sputils.py:23-24) of every row of a [n_rows x n] tensor (of the one row of a 1-D tensor)"""
        one = a.dim() == 1
        refresh = []
        a2, p_a, pitch, n = self._rows("a", a, refresh=refresh)
        n_rows = 1 if one else int(a2.shape[0])
        if out is None:
            out = self.empty(n_rows)
        elif not isinstance(out, torch.Tensor) or out.device != self.device or out.dtype != self.dtype or tuple(out.shape) != (n_rows,) \
                or not out.is_contiguous():
            raise ValueError("out must be a contiguous %s vector of %d on %s" % (self.dtype, n_rows, self.device))
        fn = getattr(self.lib, "spc_rms_" + _DTYPES[self.dtype])
        return OperatorPlan(self, fn, (n_rows, n, max(pitch, n), p_a, out.data_ptr()), [a2, out], {"out": out}, result=out[0] if one else out,
                            refresh=refresh)

    def rms(self, a, stream=None, out=None):
        return self.plan_rms(a, out=out).run(stream)

    # -- surface fluxes of columns without an LES -------------------------------------------------
    @_on_engine_stream
    def plan_surface_fluxes(self, Ph_s, T_s, QLflux, QIflux, SHflux, TSflux, out=None):
        """(wthl, wqt) of spcpl.convert_surface_fluxes (splib/spcpl.py:153-161) for [n] scalars."""
        n = int(Ph_s.shape[0])
        ck = _Checker(self.device, self.dtype)
        ptrs = [ck.vec(nm, t, n) for nm, t in (("Ph_s", Ph_s), ("T_s", T_s), ("QLflux", QLflux), ("QIflux", QIflux),
                                               ("SHflux", SHflux), ("TSflux", TSflux))]
        out = dict(out or {})
        wthl = out["wthl"] if "wthl" in out else self.empty(n)
        wqt = out["wqt"] if "wqt" in out else self.empty(n)
        optr = [ck.vec("wthl", wthl, n), ck.vec("wqt", wqt, n)]
        fn = getattr(self.lib, "spc_surface_fluxes_" + _DTYPES[self.dtype])
        return SurfacePlan(self, fn, [n] + ptrs + optr, ck.keep, {"wthl": wthl, "wqt": wqt})

    @_on_engine_stream
    def surface_fluxes(self, Ph_s, T_s, QLflux, QIflux, SHflux, TSflux, stream=None):
        r = self.plan_surface_fluxes(Ph_s, T_s, QLflux, QIflux, SHflux, TSflux).launch(stream)
        return r["wthl"], r["wqt"]
