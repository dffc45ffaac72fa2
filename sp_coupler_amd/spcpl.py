"""Reference-named host API of the coupling step (mirror of ``splib/spcpl.py``), batched on the GPU.

Same function names, argument meaning and return values as the reference module, so that
``splib.step`` (``splib/splib.py:267-352``) can call it unchanged:

    gather_gcm_data(gcm, les_models, couple_surface, output_column_indices=None, write=True)   spcpl.py:55
    convert_profiles(les, write=True) -> (u, v, thl, qt, ps, ql)                               spcpl.py:171
    set_les_forcings(les, gcm, asynchronous, firststep, profile, dt_gcm, factor, couple_surface,
                     qt_forcing='sp', write=True, variability_nudge_constant_T=False) -> dict  spcpl.py:299
    get_les_profiles(les, asynchronous) -> dict                                                spcpl.py:747
    get_cloud_fraction(les)                                                                    spcpl.py:22
    set_gcm_tendencies(gcm, les, profile, dt_gcm, factor=1, write=True, conservative=False)    spcpl.py:388
    write_les_profiles(les) / set_les_state(les, u, v, thl, qt, ps=None)                       spcpl.py:574/274
    set_gcm_tendencies_from_file(gcm, les)                                                     spcpl.py:558
    convert_surface_fluxes(les) / output_column_conversion(profile)                            spcpl.py:136/251

What differs is WHERE the arithmetic runs: ``gather_gcm_data`` packs every SP column into
``[n_cols x n_lev]`` tensors in HBM (a ``ColumnBatch``); the first per-``les`` call of a step launches
ONE HIP kernel for all columns (``*_batched`` twins, usable directly), later per-``les`` calls only
fan rows out to the model setters.  Nothing here computes on the CPU: without the HIP extension and
a GPU every entry point raises.

Units: the reference passes AMUSE quantities.  Every unit on this path is SI-coherent with factor 1
(SURVEY.md section 8(c)), so values are taken with ``.number`` when present; results handed to model
setters are plain float64 NumPy arrays unless ``set_unit_wrapper`` installs a wrapper
(INTEGRATION.md shows the OMUSE one).
"""
import logging
import time

import numpy
import torch

from . import transfer
from .engine import Engine
from .transfer import Arena, StepTrace  # noqa: F401  (re-exported: spcpl.Arena)

log = logging.getLogger(__name__)

# splib/spcpl.py:32-33
gcm_vars = ["U", "V", "T", "SH", "QL", "QI", "Pfull", "Phalf", "A", "Zgfull", "Zghalf"]
surf_vars = ["Z0M", "Z0H", "QLflux", "QIflux", "SHflux", "TLflux", "TSflux"]
# splib/spcpl.py:47-51
var_to_netcdf_name = {"Z0M": "z0m", "Z0H": "z0h", "Phalf": "Ph", "Pfull": "Pf"}
# keys of the dict get_les_profiles returns (splib/spcpl.py:767)
les_profile_keys = ["U", "V", "presf", "Rhof", "Rhobf", "THL", "QT", "QL", "QL_ice", "QR", "PS", "T", "A", "Rain"]
# setter unit names, for an optional unit wrapper (splib/spcpl.py:353-358, 546-554)
output_units = {"f_u": "m/s**2", "f_v": "m/s**2", "f_thl": "K/s", "f_qt": "mfu/s", "f_ql": "mfu/s", "f_ps": "Pa/s",
                "ql_ref": "mfu", "z0m": "m", "z0h": "m", "wthl": "m*K/s", "wqt": "m/s", "f_U": "m/s**2",
                "f_V": "m/s**2", "f_T": "K/s", "f_SH": "shu/s", "f_QL": "mfu/s", "f_QI": "mfu/s", "f_A": "ccu/s",
                "u": "m/s", "v": "m/s", "thl": "K", "qt": "mfu", "ql": "mfu", "ps": "Pa", "Zf": "m", "Zh": "m"}

_engine = None
_unit_wrapper = None
writer = None      # optional spifs writer (sp_coupler_amd.spio.SpifsWriter); None = no output
writer_rows = {}   # GCM grid index -> column index in the writer's file, for the extra output columns


def get_engine():
    """the engine of this process.  Default: ONE plain ``Engine`` -- inside a one-rank-per-GPU job (WORLD_SIZE > 1) on
    that rank's LOCAL_RANK device, else on the current device.  Every visible GPU behind this ONE master process
    (multi.MultiDeviceEngine: the column batch split into row blocks, one per GPU -- the reference's master is one
    process holding all LES objects, splib/splib.py:146-154) is OPT-IN: ``SPC_DEVICES=all`` or ``SPC_DEVICES=0,1,...``
    in the environment, or ``set_engine(MultiDeviceEngine(...))``; the choice and the row partition are logged.  (It
    was the default in round 3; it has only ever run on two engines sharing one card, so a multi-GPU node no longer
    gets it silently.)"""
    global _engine
    if _engine is None:
        import os
        want = os.environ.get("SPC_DEVICES", "").strip()
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            _engine = Engine("cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")))
        elif want and os.environ.get("SPC_SINGLE_GPU") != "1":
            from .multi import MultiDeviceEngine
            _engine = MultiDeviceEngine()
            log.warning("sp_coupler_amd: SPC_DEVICES=%s -> MultiDeviceEngine on %s (row blocks of >= %d columns per device)",
                        want, ", ".join(str(e.device) for e in _engine.engines), _engine.min_cols_per_device)
        else:
            _engine = Engine()
            if torch.cuda.device_count() > 1:
                log.info("sp_coupler_amd: %d GPUs visible, using %s only (SPC_DEVICES=all spreads the column batch over all of them)",
                         torch.cuda.device_count(), _engine.device)
    return _engine


def set_engine(engine):
    """install (or, with None, drop) the engine; transfer buffers and launch plans bound to the old one are released"""
    global _engine, _current
    _engine = engine
    _buffers.clear()
    _current = None


def set_unit_wrapper(fn):
    """fn(name, ndarray) -> object handed to model setters (e.g. ``arr | unit``); None = plain arrays."""
    global _unit_wrapper
    _unit_wrapper = fn


def _wrap(name, value):
    return value if _unit_wrapper is None else _unit_wrapper(name, value)


def _num(q):
    """bare float64 numbers of a (possibly unit-carrying) value"""
    if hasattr(q, "number"):
        q = q.number
    return numpy.asarray(q, dtype=numpy.float64)


def _to_host(t):
    """device tensor (or a transfer.Sharded array of a multi-device engine) -> NumPy array.  Callers on the slow paths run
    under ``engine.on_stream()``, so the copy is ordered behind the engine's launches."""
    if isinstance(t, transfer.Sharded):
        return t.to_host()          # block by block, concatenated on the host: no device holds the whole batch
    return t.cpu().numpy()


def _result(x):
    """value of an async request (``.result()``) or the value itself"""
    return x.result() if hasattr(x, "result") and callable(x.result) else x


_F64, _I32 = torch.float64, torch.int32
#: LES slab means the kernels consume (forward: spcpl.py:310-315; backward: spcpl.py:393-411); scalars last
_LES_FWD_LEVELS = ("U", "V", "THL", "QT", "QL")
_LES_BWD_LEVELS = ("T", "QL_ice")
_LES_IN_LEVELS = _LES_FWD_LEVELS + _LES_BWD_LEVELS
_LES_DIAG_LEVELS = ("Rhobf", "presf", "Rhof", "QR")          # conservative coarsening / spifs diagnostics only
_LES_IN_SCALARS = ("PS", "Rain", "rain_last")


class StepBuffers:
    """Transfer buffers and launch plans of one batch geometry, allocated ONCE and reused every step (pinned
    allocations cost milliseconds): GCM state up, LES slab means up, forcings down, tendencies down."""

    def __init__(self, engine, n, n_total, nG, nL, with_surf):
        self.engine, self.n, self.n_total, self.nG, self.nL = engine, n, n_total, nG, nL
        dt = engine.dtype
        g = [(v, (n_total, nG + 1 if v in ("Phalf", "Zghalf") else nG), dt) for v in gcm_vars]
        if with_surf:
            g += [(v, (n_total,), dt) for v in surf_vars]
        self.gcm_in = engine.arena(g, rows=n)
        # upload order = order of need: what K1 reads (first step: only these are known), then what only K3 reads, then
        # the diagnostics of conservative coarsening / spifs -- a copy "up to X" never carries more than it must
        self.les_in = engine.arena([(k, (n, nL), dt) for k in _LES_FWD_LEVELS] + [(k, (n,), dt) for k in _LES_IN_SCALARS]
                            + [(k, (n, nL), dt) for k in _LES_BWD_LEVELS] + [("A", (n, nG), dt)]
                            + [(k, (n, nL), dt) for k in _LES_DIAG_LEVELS], rows=n)
        # what the 7 LES setters + get_cloudfraction need comes first ("core": downloaded every step); heights
        # and surface fluxes follow and cross PCIe only when somebody asks for them
        self.fwd_out = engine.arena([(k, (n, nL), dt) for k in ("f_u", "f_v", "f_thl", "f_qt", "f_ql", "ql_ref")]
                             + [("f_ps", (n,), dt), ("idx", (n, nG), _I32)]
                             + [("wthl", (n,), dt), ("wqt", (n,), dt), ("Zf", (n, nG), dt), ("Zh", (n, nG + 1), dt),
                                ("Tv", (n, nG), dt), ("THL", (n, nG), dt), ("QT", (n, nG), dt)]
                             + [(k, (n, nL), dt) for k in ("u", "v", "thl", "qt")] + [("ps", (n,), dt)], rows=n)
        self.bwd_out = engine.arena([(k, (n, nG), dt) for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A")]
                                    + [("start_index", (n,), _I32)], rows=n)
        # large batches: every array crosses PCIe on its own, overlapped with the model's next getter / setter call; small
        # ones (arrays under transfer.Arena.PIECE_MIN): one call, one copy per group as before (the per-array calls and
        # events would cost more host time than the overlap hides)
        self.piecewise = n * nL * 8 >= transfer.Arena.PIECE_MIN
        self.plans = {}
        self.grid_key = None
        self.zf = self.zh = self.zf_host = self.zh_host = None

    def set_grid(self, zf_host, zh_host):
        key = (zf_host.shape, zf_host.tobytes(), zh_host.tobytes()) if zf_host.size <= 4096 else None
        if key is None or key != self.grid_key:
            self.zf_host, self.zh_host = zf_host, zh_host
            rows = None if zf_host.ndim == 1 else self.n           # shared grid: replicated; per-column grid: row-sharded
            self.zf = self.engine.to_devices(zf_host, rows, n_cols=self.n)
            self.zh = self.engine.to_devices(zh_host, rows, n_cols=self.n)
            self.grid_key = key
            self.plans = {}


_buffers = {}


def _get_buffers(engine, n, n_total, nG, nL, with_surf):
    key = (id(engine), n, n_total, nG, nL, bool(with_surf))
    b = _buffers.get(key)
    if b is None:
        if len(_buffers) >= 4:      # a run has one geometry (plus spin-up variants): do not hoard pinned memory
            _buffers.clear()
        b = _buffers[key] = StepBuffers(engine, n, n_total, nG, nL, with_surf)
        b.key = key
    return b


def _is_ensemble(les_models):
    """the optional batched LES protocol (sp_coupler_amd.models docstring)"""
    return bool(getattr(les_models, "batched", False))


class LazyRow:
    """``les.gcm_Zf`` / ``les.gcm_Zh`` (splib/spcpl.py:200-201) of one column, computed on FIRST READ: the heights of a
    step come from the diagnostics kernel K5 + one device->host copy for all columns, which a step whose caller never
    looks at them does not pay.  Behaves like the float64 row it stands for (``numpy.asarray``, indexing, ``len``,
    arithmetic, attribute access such as ``.shape`` / ``.number``); bound once to (batch, column) and valid for every
    later step of that batch."""

    __slots__ = ("_batch", "_key", "_i", "_step", "_v")

    def __init__(self, batch, key, i):
        self._batch, self._key, self._i, self._step, self._v = batch, key, i, -1, None

    def _get(self):
        b = self._batch
        if self._step != b.step_id:
            self._v = _wrap(self._key, _heights(b)[self._key][self._i])
            self._step = b.step_id
        return self._v

    def __array__(self, dtype=None, copy=None):
        a = numpy.asarray(_num(self._get()))
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getattr__(self, name):
        return getattr(self._get(), name)

    def __getitem__(self, k):
        return self._get()[k]

    def __len__(self):
        return len(self._get())

    def __iter__(self):
        return iter(self._get())

    def __repr__(self):
        return "LazyRow(%s[%d]: %r)" % (self._key, self._i, self._get())


def _lazy_binop(name):
    def op(self, other):
        return getattr(self._get(), name)(other._get() if isinstance(other, LazyRow) else other)
    op.__name__ = name
    return op


for _n in ("add", "sub", "mul", "truediv", "pow", "radd", "rsub", "rmul", "rtruediv", "lt", "le", "gt", "ge", "eq", "ne"):
    setattr(LazyRow, "__%s__" % _n, _lazy_binop("__%s__" % _n))
LazyRow.__neg__ = lambda self: -self._get()
LazyRow.__hash__ = None


class ColumnBatch:
    """All SP columns of a run, resident in HBM as [n_cols x n_lev] tensors (views into the transfer buffers: ONE
    host->device copy per step for the whole GCM state, one for the LES slab means).  The batch object PERSISTS across
    steps while the set of LES objects, their GCM columns and the surface-coupling switch stay the same (``refill``
    re-reads the GCM and resets the per-step results), so the per-LES handles (``les._spc_batch``, ``les._spc_row``, the
    lazy ``les.gcm_Zf`` / ``gcm_Zh``) are attached once, not every step."""

    def __init__(self, engine, les_models, gcm, cols, extra_cols, couple_surface):
        self.engine = engine
        self.ens = les_models if _is_ensemble(les_models) else None
        self.les_models = les_models if self.ens is not None else list(les_models)
        self.n = len(self.les_models)
        self.row = None if self.ens is not None else {id(les): i for i, les in enumerate(self.les_models)}
        self.extra_cols = list(extra_cols)
        self.cols = list(cols)
        self.cols_arr = numpy.asarray(self.cols)          # what the GCM getters are handed (an index array, not a list)
        self.gi_ref = self.ens.grid_indices if self.ens is not None else None
        self.gi_copy = numpy.array(self.gi_ref) if self.ens is not None else None     # an ensemble may re-mask IN PLACE
        self.couple_surface = bool(couple_surface)
        n_total = len(cols)
        self.use_out = bool(getattr(gcm, "supports_out", False))
        first = None
        nL = self._les_levels()
        nG = getattr(gcm, "ktot", None)
        if nG is None:                                    # learn the level count from the first variable
            first = _num(gcm.get_profile_fields(gcm_vars[0], cols))
            nG = first.shape[1]
        self.buf = b = _get_buffers(engine, self.n, n_total, int(nG), nL, couple_surface)
        hn = b.gcm_in.hn
        self.gcm_host = {v: hn[v] for v in gcm_vars}      # rows n.. are the extra output columns
        self.surf_host = {v: hn[v] for v in surf_vars} if couple_surface else {}
        n = self.n
        self.gcm = {k: t[:n] for k, t in b.gcm_in.d.items()}
        self.profiles = {}                # id(les) -> profile dict (values or async requests)
        self.profile_generation = 0
        self.step_id = 0
        self._reset_step()
        self._fill(gcm, first)
        if self.n:
            self._pack_les_grid()

    def _reset_step(self):
        self.step_id += 1
        self.fwd = None                   # host results of the last forward launch
        self.fwd_key = self.fwd_raw = None
        self.fwd_rows = None              # per-LES protocol: row views of this step's snapshot of the forward results
        self.fwd_written = False
        self.bwd = None
        self.bwd_key = self.bwd_raw = None
        self.bwd_rows = None
        self.dev_prof = None              # device views of the LES slab means uploaded for this step
        self.diag_host = self.conv = self.idx_host = self.idx_rows = None
        self.wlp = self.wlp_key = None
        self.ql_ref_host = None

    def _fill(self, gcm, first=None):
        """one ``get_profile_fields`` per variable straight into the pinned buffer; each variable starts crossing PCIe on
        the buffer's copy stream as soon as it is there, overlapped with the fetch of the next one (spcpl.py:62-75)"""
        arena = self.buf.gcm_in
        hn, cols, use_out = arena.hn, self.cols_arr, self.use_out
        for v in gcm_vars:                                                    # spcpl.py:62-67
            arena.writable(v)                       # (last step's upload of it has long left the pinned buffer: free)
            if v == gcm_vars[0] and first is not None:
                numpy.copyto(hn[v], first)
            elif use_out:
                gcm.get_profile_fields(v, cols, out=hn[v])
            else:
                numpy.copyto(hn[v], _num(gcm.get_profile_fields(v, cols)))
            arena.push(v, "h2d_gcm")                # on the wire while the next variable is being fetched from the GCM
        if self.couple_surface:
            for v in surf_vars:                                               # spcpl.py:69-75
                arena.writable(v)
                if use_out:
                    gcm.get_surface_field(v, cols, out=hn[v])
                else:
                    numpy.copyto(hn[v], _num(gcm.get_surface_field(v, cols)))
                arena.push(v, "h2d_gcm")

    def reusable_for(self, engine, les_models, cols, couple_surface):
        """same engine, same LES objects in the same order, same GCM columns, same surface switch, buffers still ours"""
        if engine is not self.engine or bool(couple_surface) != self.couple_surface or cols != self.cols:
            return False
        if _buffers.get(self.buf.key) is not self.buf:
            return False
        if self.ens is not None:
            return les_models is self.ens
        if _is_ensemble(les_models) or len(les_models) != self.n:
            return False
        mine = self.les_models
        return all(a is b for a, b in zip(les_models, mine))

    def refill(self, gcm):
        """next step of the same batch: per-step results dropped, GCM state re-read and uploaded.  The slab means
        fetched by get_les_profiles() after the previous LES run stay (``profiles``)."""
        self.use_out = bool(getattr(gcm, "supports_out", False))
        self._reset_step()
        self._fill(gcm)

    def _les_levels(self):
        if not len(self.les_models):
            return 1
        z = self.ens.zf_cache if self.ens is not None else self.les_models[0].zf_cache
        return int(_num(z).shape[-1])

    def _pack_les_grid(self):
        """les.zf_cache / les.zh_cache (splib/splib.py:152-153, set once per LES): one shared [nL] grid when all LES
        instances agree (the normal case), else [n x nL]."""
        if self.ens is not None:
            zf_host, zh_host = _num(self.ens.zf_cache), _num(self.ens.zh_cache)
        else:
            zfs = [_num(les.zf_cache) for les in self.les_models]
            zhs = [_num(les.zh_cache) for les in self.les_models]
            shared = all(z is zfs[0] or (z.shape == zfs[0].shape and numpy.array_equal(z, zfs[0])) for z in zfs) and \
                all(z is zhs[0] or numpy.array_equal(z, zhs[0]) for z in zhs)
            zf_host = zfs[0] if shared else numpy.stack(zfs)
            zh_host = zhs[0] if shared else numpy.stack(zhs)
        self.buf.set_grid(zf_host, zh_host)
        self.zf, self.zh, self.zf_host, self.zh_host = self.buf.zf, self.buf.zh, self.buf.zf_host, self.buf.zh_host

    def attach(self, attach_rows=False):
        """hand every LES object its handles (once per batch; the reference scatters rows here, spcpl.py:81-86)"""
        if self.ens is not None:
            self.ens._spc_batch = self
            return
        for i, les in enumerate(self.les_models):
            les._spc_batch, les._spc_row = self, i
            les.gcm_Zf, les.gcm_Zh = LazyRow(self, "Zf", i), LazyRow(self, "Zh", i)         # spcpl.py:200-201, on first read
        if attach_rows:
            self.attach_gcm_rows()

    def attach_gcm_rows(self):
        for i, les in enumerate(self.les_models):
            for v in gcm_vars:
                setattr(les, v, self.gcm_host[v][i].copy())
            for v in self.surf_host:
                setattr(les, v, self.surf_host[v][i])

    def index_of(self, les):
        if self.ens is not None:
            return les._i
        return self.row[id(les)]

    # ---- LES slab means -> device -------------------------------------------------------------
    def stack_profiles(self, keys, source):
        """per-LES protocol: source(les) -> dict (values, AMUSE-style quantities or async requests).  Rows are written
        straight into the pinned upload buffer (one C loop per variable; the kind of value -- request / quantity / bare
        array -- is looked at on the FIRST column only); each variable goes on the wire as soon as it is packed."""
        hn = self.buf.les_in.hn
        rows = [source(les) for les in self.les_models]
        for k in keys:
            dst = hn[k]
            self.buf.les_in.writable(k)
            try:                                    # fast: all columns hold the same kind of value
                vals = [r[k] for r in rows]
                v0 = vals[0]
                if hasattr(v0, "result") and callable(v0.result):
                    vals = [v.result() for v in vals]
                    v0 = vals[0]
                if hasattr(v0, "number"):
                    vals = [v.number for v in vals]
                if dst.ndim == 2:
                    numpy.concatenate(vals, out=dst.reshape(-1))      # n rows of nL float64 -> the [n x nL] block
                else:
                    dst[:] = vals
            except (TypeError, ValueError, AttributeError):           # mixed kinds / dtypes / shapes: column by column
                vals = [_num(_result(r[k])) for r in rows]
                if dst.ndim == 2:
                    numpy.stack(vals, out=dst)
                else:
                    dst[:] = vals
            self.buf.les_in.push(k, "h2d_les")       # on the wire while the next variable is being packed
        return {k: self.buf.les_in.d[k] for k in keys}

    def upload_profiles(self, keys, arrays=None):
        """batched protocol: ``arrays`` (dict key -> [n x ...]) are copied into the upload buffer unless they ARE its
        views already (ensemble getters write there directly); one host->device copy."""
        hn = self.buf.les_in.hn
        for k in keys:
            if arrays is not None and arrays[k] is not hn[k]:
                self.buf.les_in.writable(k)
                numpy.copyto(hn[k], _num(arrays[k]))
            self.buf.les_in.push(k, "h2d_les")
        return {k: self.buf.les_in.d[k] for k in keys}


_current = None


def current_batch():
    return _current


def _batch_of(les):
    b = getattr(les, "_spc_batch", None)
    if b is None and hasattr(les, "_e"):               # per-column face of an ensemble
        b = getattr(les._e, "_spc_batch", None)
    if b is None:
        raise RuntimeError("gather_gcm_data() must be called before the per-les coupling functions")
    return b


# ---------------------------------------------------------------------------------------------
# gather: splib/spcpl.py:55-132
# ---------------------------------------------------------------------------------------------
def gather_gcm_data(gcm, les_models, couple_surface, output_column_indices=None, write=True, attach_rows=False):
    """Pull all SP columns' profiles with ONE ``gcm.get_profile_fields`` call per variable (as the
    reference does, spcpl.py:66) straight into the pinned upload buffer and send them to HBM in ONE copy.  The
    reference then scatters rows onto the ``les`` objects (spcpl.py:81-86); here each ``les`` gets a handle to the
    batch instead (``attach_rows=True`` also sets the per-variable row views for code that reads ``les.T`` etc.)."""
    global _current
    extra_cols = [] if output_column_indices is None else list(output_column_indices)
    ens = les_models if _is_ensemble(les_models) else None
    start = time.time()
    eng = get_engine()
    prev = _current
    if (ens is not None and prev is not None and prev.ens is ens and prev.gi_ref is ens.grid_indices and prev.engine is eng
            and prev.extra_cols == extra_cols and prev.couple_surface == bool(couple_surface)
            and _buffers.get(prev.buf.key) is prev.buf and numpy.array_equal(prev.gi_copy, ens.grid_indices)):
        cols = prev.cols                     # the same ensemble object with the same columns as last step: nothing to rebuild
    else:
        cols = ([int(g) for g in ens.grid_indices] if ens is not None else [les.grid_index for les in les_models]) + extra_cols
    if not any(cols):                                                        # quirk kept: spcpl.py:63,71
        _current = None
        return None
    if prev is not None and (cols is prev.cols or prev.reusable_for(eng, les_models, cols, couple_surface)):
        batch = prev                                     # same columns as last step: handles stay, buffers are refilled
        batch.refill(gcm)
        if attach_rows and batch.ens is None:
            batch.attach_gcm_rows()
    else:
        batch = ColumnBatch(eng, les_models, gcm, cols, extra_cols, couple_surface)
        if prev is not None and batch.row is not None:   # slab means fetched by get_les_profiles() after the previous LES run
            batch.profiles = {k: v for k, v in prev.profiles.items() if k in batch.row}
        batch.attach(attach_rows)
    log.info("Fetching gcm data took %d s" % (time.time() - start))
    profile_data, surface_data = batch.gcm_host, batch.surf_host
    # extra output columns: spcpl.py:89-129
    if extra_cols and write and writer is not None:
        n0 = batch.n
        C = {var_to_netcdf_name.get(v, v): profile_data[v][n0:] for v in gcm_vars}
        for v in surface_data:
            C[v] = surface_data[v][n0:]
        D = output_column_conversion(C)
        rows = [writer_rows[c] for c in extra_cols]      # column index of each extra output column in the file
        writer.write(rows=rows, **{k: D[k] for k in ("U", "V", "T", "SH", "QL", "QI", "Pf", "Ph", "Zf", "Zh", "Psurf",
                                                      "Tv", "THL", "QT", "A")})
        if couple_surface:                                                   # spcpl.py:112-129
            Cs = dict(C, Ph=profile_data["Phalf"][n0:], T=profile_data["T"][n0:])
            z0m, z0h, wthl, wqt = convert_surface_fluxes(Cs)
            writer.write(rows=rows, z0m=_num(z0m), z0h=_num(z0h), wthl=_num(wthl), wqt=_num(wqt),
                         **{k: surface_data[k][n0:] for k in ("TLflux", "TSflux", "SHflux", "QLflux", "QIflux")})
    _current = batch
    return batch


# ---------------------------------------------------------------------------------------------
# forward: convert_profiles + set_les_forcings
# ---------------------------------------------------------------------------------------------
_FWD_KEYS = ("U", "V", "THL", "QT", "QL", "PS", "Rain")
_FWD_CORE = ("f_u", "f_v", "f_thl", "f_qt", "f_ql", "ql_ref", "f_ps", "idx")
_FWD_PULL = ("f_u", "f_v", "f_thl", "f_qt", "f_ps", "f_ql", "ql_ref", "idx")     # download order = setter order (spcpl.py:341-347)
_FWD_SURF = ("z0m", "z0h", "wthl", "wqt")
_DIAG_GCM = ("Zf", "Zh", "Tv", "THL", "QT")       # K5's GCM-level outputs; contiguous in StepBuffers.fwd_out


def _first_step_profile(les):
    """splib/spcpl.py:302-308, 321: live getters on the first step"""
    return {"U": les.get_profile_U(), "V": les.get_profile_V(), "THL": les.get_profile_THL(),
            "QT": les.get_profile_QT(), "QL": les.get_profile_QL(), "PS": les.get_surface_pressure(),
            "Rain": les.get_rain()}


def _plan(batch, kind, flags, make):
    """launch plan of this batch geometry, built once and reused every step (the tensors it binds are views into
    the StepBuffers, which persist)"""
    key = (kind,) + tuple(flags)
    plan = batch.buf.plans.get(key)
    if plan is None:
        plan = batch.buf.plans[key] = make()
    return plan


def _inputs_ready(batch):
    """the current stream waits for the uploads still in flight on the transfer buffers' copy streams (and for downloads
    of the result buffers nobody has waited for: the next kernel overwrites them)"""
    batch.buf.gcm_in.fence()
    batch.buf.les_in.fence()
    batch.buf.fwd_out.settle()
    batch.buf.bwd_out.settle()


def _launch(batch, plan, what):
    _inputs_ready(batch)
    if transfer.trace is not None:
        with transfer.trace.region(what, 0, batch.engine.device):
            plan.launch()
    else:
        plan.launch()
    return plan.outputs


def forward_batched(batch, profiles, dt_gcm, factor, couple_surface=False, wait=True):
    """K1 (+fused K2) for every column of ``batch`` -- the LEAN hot-path kernel bench.py times (the six setter
    arrays, f_ps and the index map; with ``couple_surface`` also the surface fluxes).  ``profiles``: dict of device
    tensors U,V,THL,QT,QL [n x nL], PS [n] (views of the step's upload buffer).  Returns dict of HOST arrays: views
    into the pinned download buffer, filled by ONE device->host copy."""
    eng, b = batch.engine, batch.buf
    dt = float(_num(dt_gcm))
    # the surface outputs live behind idx in the buffer; z0m / z0h are pass-throughs written by the kernel too
    out = {k: b.fwd_out.d[k] for k in _FWD_CORE}
    prof = {k: profiles[k] for k in ("U", "V", "THL", "QT", "QL", "PS")}
    if couple_surface:
        out.update(wthl=b.fwd_out.d["wthl"], wqt=b.fwd_out.d["wqt"])
    plan = _plan(batch, "fwd", (bool(couple_surface),), lambda: eng.plan_forward(
        batch.gcm, batch.zf, prof, float(factor), dt, zh=batch.zh, want_profiles=False, want_heights=False,
        couple_surface=couple_surface, out=out))
    plan.set_scalars(float(factor), dt)
    _launch(batch, plan, "k1")
    # results come back array by array on the buffer's copy stream; `wait=False`: the caller takes each one when it has
    # landed (b.fwd_out.ready(name)) and hands it to the model while the next ones are still on the wire
    b.fwd_out.pull(_FWD_PULL + (("wthl", "wqt") if couple_surface else ()), "d2h_forcings")
    if wait:
        b.fwd_out.ready()
    host = {k: b.fwd_out.hn[k] for k in _FWD_CORE}
    if couple_surface:
        host["wthl"], host["wqt"] = b.fwd_out.hn["wthl"], b.fwd_out.hn["wqt"]
        host["z0m"] = numpy.array(batch.surf_host["Z0M"][:batch.n])      # pass-throughs (spcpl.py:146-147): the host has them
        host["z0h"] = numpy.array(batch.surf_host["Z0H"][:batch.n])
    host["ql"] = host["ql_ref"]
    return host


def _heights(batch):
    """Zf, Zh (+ Tv, THL, QT) of every column on the host: K5 on the batch (cached launch plan writing into the step's
    download buffer, one device->host copy of that span), run only when somebody needs them (a read of les.gcm_Zf /
    gcm_Zh, the spifs writer); spcpl.py:176, 197-198, 214-215."""
    if batch.diag_host is None:
        b = batch.buf
        plan = _plan(batch, "diag", (), lambda: batch.engine.plan_diagnostics(
            batch.gcm, out={k: b.fwd_out.d[k] for k in _DIAG_GCM}))
        _launch(batch, plan, "k5")
        b.fwd_out.download(upto=_DIAG_GCM[-1], start=_DIAG_GCM[0], what="d2h_heights")
        batch.diag_host = {k: numpy.array(b.fwd_out.hn[k]) for k in _DIAG_GCM}
    return batch.diag_host


def _finish_forward(batch, host, rain, rain_last, dt_gcm):
    dt = float(_num(dt_gcm))
    host["rain"] = numpy.array(rain, dtype=numpy.float64)
    host["rainrate"] = (host["rain"] - rain_last) / dt                        # spcpl.py:325 (IEEE: same on host)
    batch.fwd = host
    batch.fwd_written = False
    batch.fwd_rows = None
    return host


def _ensure_forward(batch, les, firststep, profile, dt_gcm, factor, couple_surface):
    key = (batch.profile_generation, bool(firststep), float(_num(dt_gcm)), float(factor), bool(couple_surface))
    if batch.fwd is None or batch.fwd_key != key:
        if firststep:
            src = _first_step_profile
        else:
            if profile is not None:
                batch.profiles.setdefault(id(les), profile)
            missing = [m for m in batch.les_models if id(m) not in batch.profiles]
            if missing:
                raise RuntimeError("set_les_forcings: LES profiles of %d columns are unknown; call get_les_profiles() "
                                   "for every LES after stepping it (as splib.step_les_models does) or use "
                                   "set_les_forcings_batched()" % len(missing))
            src = lambda m: batch.profiles[id(m)]               # noqa: E731
        prof = batch.stack_profiles(_FWD_KEYS, src)
        rain_last = numpy.array([float(_num(getattr(m, "rain", 0.0))) for m in batch.les_models])   # spcpl.py:316-319
        host = forward_batched(batch, prof, dt_gcm, factor, couple_surface)
        _finish_forward(batch, host, batch.buf.les_in.hn["Rain"], rain_last, dt_gcm)
        batch.fwd_key = key
    # the per-LES loop passes the SAME objects for every column of a step: later calls recognise them by identity
    batch.fwd_raw = (firststep, dt_gcm, factor, couple_surface, batch.profile_generation)
    return batch.fwd


def _rows_of(block):
    """fresh copy of a [n x m] (or [n]) result block and the list of its rows: what the per-LES setters receive.  A
    model may keep the array it is handed -- it never aliases a transfer buffer a later step overwrites -- and the
    whole block is copied ONCE per step instead of once per column and variable."""
    return list(numpy.array(block))


def _forward_rows(batch):
    f = batch.fwd
    rows = [_rows_of(f[k]) for k in ("f_u", "f_v", "f_thl", "f_qt", "f_ps", "f_ql", "ql_ref", "rain")]
    if "wthl" in f:
        rows += [_rows_of(f[k]) for k in _FWD_SURF]
    if _unit_wrapper is not None:
        names = ("f_u", "f_v", "f_thl", "f_qt", "f_ps", "f_ql", "ql_ref", None) + _FWD_SURF
        rows = [r if nm is None else [_unit_wrapper(nm, v) for v in r] for nm, r in zip(names, rows)]
    batch.fwd_rows = rows
    return rows


def _write_forward(batch):
    """spifs rows of convert_profiles + set_les_forcings for ALL columns (spcpl.py:230-244, 352-376)."""
    f, g = batch.fwd, batch.gcm_host
    n = batch.n
    d = _heights(batch)                                                                      # K5: Tv, THL, QT, Zf, Zh
    writer.write(U=g["U"][:n], V=g["V"][:n], T=g["T"][:n], SH=g["SH"][:n], QL=g["QL"][:n], QI=g["QI"][:n],
                 Pf=g["Pfull"][:n], Ph=g["Phalf"][:n, 1:], Zf=d["Zf"], Zh=d["Zh"][:, 1:], Psurf=g["Phalf"][:n, -1],
                 Tv=d["Tv"], THL=d["THL"], QT=d["QT"], f_u=f["f_u"], f_v=f["f_v"], f_thl=f["f_thl"], f_qt=f["f_qt"],
                 rain=f["rain"], rainrate=f["rainrate"] * 3600)                              # spcpl.py:358
    if "wthl" in f:
        s = batch.surf_host
        writer.write(z0m=f["z0m"], z0h=f["z0h"], wthl=f["wthl"], wqt=f["wqt"], TLflux=s["TLflux"][:n],
                     TSflux=s["TSflux"][:n], SHflux=s["SHflux"][:n], QLflux=s["QLflux"][:n], QIflux=s["QIflux"][:n])
    batch.fwd_written = True


def _attach_heights(batch, les, i):
    """les.gcm_Zf / les.gcm_Zh (spcpl.py:200-201) as lazy rows of THIS batch, valid for every later step of it (a concrete
    array stored here would go stale: the reference re-assigns them in every set_les_forcings, this path does not)"""
    z = getattr(les, "gcm_Zf", None)
    if not (isinstance(z, LazyRow) and z._batch is batch and z._i == i):
        les.gcm_Zf, les.gcm_Zh = LazyRow(batch, "Zf", i), LazyRow(batch, "Zh", i)


def convert_profiles(les, write=True):
    """splib/spcpl.py:171-246: (u, v, thl, qt, ps, ql) for one column; caches les.gcm_Zf / gcm_Zh."""
    batch = _batch_of(les)
    i = batch.index_of(les)
    if batch.conv is None:
        eng = batch.engine
        nL = batch.zf.shape[-1]
        with eng.on_stream():
            z = eng.to_devices(numpy.zeros((batch.n, nL)), rows=batch.n)        # row blocks per device when the batch is sharded
            dummy = {"U": z, "V": z, "THL": z, "QT": z, "QL": z, "PS": eng.to_devices(numpy.zeros(batch.n), rows=batch.n)}
        _inputs_ready(batch)
        with eng.on_stream():
            res = eng.forward(batch.gcm, batch.zf, dummy, 0.0, 1.0, want_profiles=True, want_heights=True)
            batch.conv = {k: _to_host(res[k]) for k in ("u", "v", "thl", "qt", "ps", "ql_ref", "Zf", "Zh")}
    c = batch.conv
    _attach_heights(batch, les, i)                                           # spcpl.py:200-201
    return (_wrap("u", c["u"][i]), _wrap("v", c["v"][i]), _wrap("thl", c["thl"][i]), _wrap("qt", c["qt"][i]),
            _wrap("ps", c["ps"][i]), _wrap("ql", c["ql_ref"][i]))


def set_les_forcings(les, gcm, asynchronous, firststep, profile, dt_gcm, factor, couple_surface, qt_forcing='sp',
                     write=True, variability_nudge_constant_T=False):
    """splib/spcpl.py:299-385. The first call of a step computes the forcings of ALL columns in one
    launch; this call then pushes column ``les``'s rows to its setters and returns the request dict."""
    try:
        batch, i = les._spc_batch, les._spc_row
    except AttributeError:                       # a per-column face of an ensemble, or gather_gcm_data() not called
        batch = _batch_of(les)
        i = batch.index_of(les)
        _attach_heights(batch, les, i)
    raw = batch.fwd_raw
    if (raw is None or raw[0] is not firststep or raw[1] is not dt_gcm or raw[2] is not factor
            or raw[3] is not couple_surface or raw[4] != batch.profile_generation):
        _ensure_forward(batch, les, firststep, profile, dt_gcm, factor, couple_surface)
    rows = batch.fwd_rows or _forward_rows(batch)
    les.rain = rows[7][i]                                                    # spcpl.py:324
    ql_ref = rows[6][i]
    req = {
        "U": les.set_tendency_U(rows[0][i], return_request=asynchronous),                  # spcpl.py:341
        "V": les.set_tendency_V(rows[1][i], return_request=asynchronous),                  # :342
        "THL": les.set_tendency_THL(rows[2][i], return_request=asynchronous),              # :343
        "QT": les.set_tendency_QT(rows[3][i], return_request=asynchronous),                # :344
        "SP": les.set_tendency_surface_pressure(rows[4][i], return_request=asynchronous),  # :345
        "QL": les.set_tendency_QL(rows[5][i], return_request=asynchronous),                # :346
        "QLp": les.set_ref_profile_QL(ql_ref, return_request=asynchronous),                # :347
    }
    les.ql_ref = ql_ref                                                      # spcpl.py:348
    if write and writer is not None and not batch.fwd_written:
        _write_forward(batch)                    # once per launch, for all columns
    if couple_surface:                                                       # spcpl.py:359-364
        req["Z0M_surf"] = les.set_z0m_surf(rows[8][i], return_request=asynchronous)
        req["Z0H_surf"] = les.set_z0h_surf(rows[9][i], return_request=asynchronous)
        req["WT_surf"] = les.set_wt_surf(rows[10][i], return_request=asynchronous)
        req["WQ_surf"] = les.set_wq_surf(rows[11][i], return_request=asynchronous)
    if qt_forcing == 'variance':                                             # spcpl.py:377-382
        if float(_num(les.get_model_time())) > 0:
            variability_nudge(les, dt_gcm, variability_nudge_constant_T)     # (the reference does not pass `write` on)
    return req


def set_les_forcings_batched(les_models, gcm, asynchronous, firststep, profiles, dt_gcm, factor, couple_surface,
                             qt_forcing='sp', write=True, variability_nudge_constant_T=False):
    """Batched twin: the whole ``for les in les_models`` loop of splib.step (splib/splib.py:317-323).
    ``profiles``: dict les -> profile dict (ignored on the first step), or -- with an LES ensemble offering the
    batched protocol -- what ``get_les_profiles_batched`` returned.  Returns the list of request dicts (empty for
    an ensemble: its setters are synchronous array hand-overs)."""
    if not len(les_models):
        return []
    if _is_ensemble(les_models):
        return _ensemble_forcings(les_models, firststep, profiles, dt_gcm, factor, couple_surface, qt_forcing, write,
                                  variability_nudge_constant_T)
    batch = _batch_of(les_models[0])
    if not firststep:
        reg = batch.profiles
        for les in les_models:
            reg[id(les)] = profiles[les]
    reqs = [set_les_forcings(les, gcm, asynchronous, firststep, None, dt_gcm, factor, couple_surface, 'sp', write)
            for les in les_models]
    if qt_forcing == 'variance':       # spcpl.py:377-382, for all LES in one launch (R drawn in the same les order)
        started = [les for les in les_models if float(_num(les.get_model_time())) > 0]
        if started:
            variability_nudge_batched(started, dt_gcm, variability_nudge_constant_T)
    return reqs


def _ensemble_forcings(ens, firststep, profiles, dt_gcm, factor, couple_surface, qt_forcing, write, constant_T):
    """ONE getter round (first step only), ONE upload, ONE launch, ONE download, ONE setter call per variable for
    all columns: the fast form of splib.py:317-323 when the LES side offers the batched protocol."""
    batch = _batch_of(ens)
    b = batch.buf
    hn = b.les_in.hn
    if qt_forcing == 'variance' and not hasattr(ens, "get_fields_batched"):
        raise NotImplementedError("qt_forcing='variance' needs the 3-D LES fields: an ensemble must offer get_fields_batched / "
                                  "set_fields_batched (sp_coupler_amd.models docstring), or pass the LES objects as a plain list")
    if firststep:                                                            # spcpl.py:302-308, 321
        if b.piecewise:
            for k in _FWD_KEYS:                 # variable by variable: each is on the wire while the next is fetched
                b.les_in.writable(k)
                ens.get_profiles_batched((k,), {k: hn[k]})
                b.les_in.push(k, "h2d_les")
        else:
            for k in _FWD_KEYS:
                b.les_in.writable(k)
            ens.get_profiles_batched(_FWD_KEYS, {k: hn[k] for k in _FWD_KEYS})
            for k in _FWD_KEYS:
                b.les_in.push(k, "h2d_les")
        b.rain_prev = numpy.zeros(batch.n)                                   # `except: rain_last = 0`, spcpl.py:316-319
    elif profiles is None or profiles.get("_buffers") is not b:
        raise RuntimeError("set_les_forcings_batched: pass what get_les_profiles_batched() returned after the last "
                           "LES step (the slab means of this batch geometry are not on the device)")
    dev = b.les_in.d
    host = forward_batched(batch, dev, dt_gcm, factor, couple_surface, wait=False)
    # the setters of spcpl.py:341-347 in their order, each as soon as ITS array has landed: the model takes f_u while
    # f_v ... are still crossing PCIe
    pairs = (("U", "f_u"), ("V", "f_v"), ("THL", "f_thl"), ("QT", "f_qt"), ("SP", "f_ps"), ("QL", "f_ql"), ("QLp", "ql_ref"))
    if b.piecewise:
        for key, name in pairs:
            b.fwd_out.ready(name)
            ens.set_forcings_batched(**{key: host[name]})
        b.fwd_out.ready()
    else:
        b.fwd_out.ready()
        ens.set_forcings_batched(**{key: host[name] for key, name in pairs})
    if couple_surface:                                                       # spcpl.py:359-364
        ens.set_forcings_batched(Z0M_surf=host["z0m"], Z0H_surf=host["z0h"], WT_surf=host["wthl"], WQ_surf=host["wqt"])
    _finish_forward(batch, host, hn["Rain"], getattr(b, "rain_prev", numpy.zeros(batch.n)), dt_gcm)
    b.rain_prev = host["rain"]                                               # les.rain = rain, spcpl.py:324
    batch.ql_ref_host = ens.ql_ref = host["ql_ref"]                          # les.ql_ref = ql, spcpl.py:348
    if write and writer is not None:
        _write_forward(batch)
    if qt_forcing == 'variance' and float(_num(ens.model_time)) > 0:         # spcpl.py:377-382
        variability_nudge_ensemble(ens, dt_gcm, constant_T)
    return []


def convert_surface_fluxes(les):
    """splib/spcpl.py:136-167: (z0m, z0h, wthl, wqt). ``les`` is an LES object of the current batch (values
    come from the fused forward launch) or, as in the reference, a dict of GCM data for columns without
    an LES (keys Z0M, Z0H, QLflux, QIflux, SHflux, TSflux, Ph, T; one column or [n x ...])."""
    if isinstance(les, dict):
        eng = get_engine()
        Ph, T = numpy.atleast_2d(_num(les["Ph"])), numpy.atleast_2d(_num(les["T"]))   # KeyError if missing, spcpl.py:146
        one = numpy.ndim(_num(les["T"])) == 1
        n = Ph.shape[0]
        dev = lambda a: eng.to_devices(numpy.ascontiguousarray(numpy.atleast_1d(_num(a)), dtype=numpy.float64), rows=n)  # noqa: E731
        with eng.on_stream():
            wthl, wqt = eng.surface_fluxes(dev(Ph[:, -1]), dev(T[:, -1]), dev(les["QLflux"]), dev(les["QIflux"]),
                                           dev(les["SHflux"]), dev(les["TSflux"]))
            wthl, wqt = _to_host(wthl), _to_host(wqt)
        sel = (lambda a: a[0]) if one else (lambda a: a)
        return les.get("Z0M"), les.get("Z0H"), _wrap("wthl", sel(wthl)), _wrap("wqt", sel(wqt))
    batch = _batch_of(les)
    if batch.fwd is None or "wthl" not in batch.fwd:
        raise RuntimeError("surface fluxes are computed by set_les_forcings(..., couple_surface=True)")
    i = batch.index_of(les)
    f = batch.fwd
    return _wrap("z0m", f["z0m"][i]), _wrap("z0h", f["z0h"][i]), _wrap("wthl", f["wthl"][i]), _wrap("wqt", f["wqt"][i])


def set_les_state(les, u, v, thl, qt, ps=None):
    """splib/spcpl.py:274-294: broadcast the profiles to 3-D fields with uniform random perturbations
    (DALES defaults; numpy's GLOBAL generator like the reference, seeded by splib.initialize with 42).
    Model initialisation, not part of the per-step path: plain host NumPy."""
    itot, jtot, ktot = les.get_itot(), les.get_jtot(), les.get_ktot()
    vabsmax, thlabsmax, qabsmax = 0.5, 0.1, 2.5e-5                           # spcpl.py:285-287
    les.set_field('U', _wrap("u", vabsmax * numpy.random.uniform(-1., 1., (itot, jtot, ktot)) + _num(u)))
    les.set_field('V', _wrap("v", vabsmax * numpy.random.uniform(-1., 1., (itot, jtot, ktot)) + _num(v)))
    les.set_field('THL', _wrap("thl", thlabsmax * numpy.random.uniform(-1., 1., (itot, jtot, ktot)) + _num(thl)))
    les.set_field('QT', _wrap("qt", qabsmax * numpy.random.uniform(-1., 1., (itot, jtot, ktot)) + _num(qt)))
    if ps:
        les.set_surface_pressure(ps)


# ---------------------------------------------------------------------------------------------
# LES profiles and the cloud-fraction index map
# ---------------------------------------------------------------------------------------------
def _index_map(batch):
    """[n x nG] int32 host array: searchsorted(zh, Zh, side='right')[:-1][::-1] per column (spcpl.py:26 / 764):
    the fused K2 output of this step's forward launch, or the standalone K2 before any forward ran"""
    if batch.fwd is not None and "idx" in batch.fwd:
        return batch.fwd["idx"]
    if batch.idx_host is None:
        _heights(batch)                                   # K5 left Zh of this step in the download buffer's device side
        b = batch.buf
        plan = _plan(batch, "idx", (), lambda: batch.engine.plan_cloud_indices(batch.zh, b.fwd_out.d["Zh"], out=b.fwd_out.d["idx"]))
        _launch(batch, plan, "k2")
        b.fwd_out.download(upto="idx", start="idx", what="d2h_idx")
        batch.idx_host = numpy.array(b.fwd_out.hn["idx"])
    return batch.idx_host


def _index_rows(batch):
    rows = batch.idx_rows
    if rows is None or rows[0] is not batch.fwd:
        rows = batch.idx_rows = (batch.fwd, _rows_of(_index_map(batch)))
    return rows[1]


def cloud_fraction_indices(les):
    """indices = searchsorted(zh, Zh, side='right')[:-1][::-1]  (splib/spcpl.py:26 / 764), from K2"""
    batch = _batch_of(les)
    return _index_rows(batch)[batch.index_of(les)]


def get_cloud_fraction(les):
    """splib/spcpl.py:22-29"""
    indices = cloud_fraction_indices(les)
    return les.get_cloudfraction(indices)[::-1]


def get_les_profiles(les, asynchronous):
    """splib/spcpl.py:747-767: 14 getters; the index map comes from the GPU (K2). The returned dict is
    also remembered so that the next forward / backward launch can batch all columns."""
    try:
        batch, i = les._spc_batch, les._spc_row
    except AttributeError:
        batch = _batch_of(les)
        i = batch.index_of(les)
    rows = batch.idx_rows
    indices = (rows[1] if rows is not None and rows[0] is batch.fwd else _index_rows(batch))[i]
    prof = {"U": les.get_profile_U(return_request=asynchronous), "V": les.get_profile_V(return_request=asynchronous),
            "presf": les.get_presf(return_request=asynchronous), "Rhof": les.get_rhof(return_request=asynchronous),
            "Rhobf": les.get_rhobf(return_request=asynchronous), "THL": les.get_profile_THL(return_request=asynchronous),
            "QT": les.get_profile_QT(return_request=asynchronous), "QL": les.get_profile_QL(return_request=asynchronous),
            "QL_ice": les.get_profile_QL_ice(return_request=asynchronous),
            "QR": les.get_profile_QR(return_request=asynchronous),
            "PS": les.get_surface_pressure(return_request=asynchronous), "T": les.get_profile_T(return_request=asynchronous),
            "A": les.get_cloudfraction(indices, return_request=asynchronous),
            "Rain": les.get_rain(return_request=asynchronous)}
    reg = batch.profiles
    key = id(les)
    if key in reg and len(reg) >= batch.n:       # a new round of fetches begins (every column has been seen once)
        reg = batch.profiles = {}
    if not reg:
        batch.profile_generation += 1
    reg[key] = prof
    return prof


def get_les_profiles_batched(les_models, asynchronous=False, diagnostics=False):
    """The 14 getters of spcpl.py:747-767 for ALL columns.  With an LES ensemble (batched protocol): one call per
    variable, written straight into the pinned upload buffer and sent on as soon as it is there (ONE upload of every
    variable, serving both this step's K3 and the next step's K1); ``diagnostics`` adds presf, Rhof, Rhobf, QR (conservative coarsening / spifs).
    With a plain list of LES objects: dict les -> get_les_profiles(les)."""
    if not _is_ensemble(les_models):
        return {les: get_les_profiles(les, asynchronous) for les in les_models}
    ens = les_models
    batch = _batch_of(ens)
    b = batch.buf
    hn = b.les_in.hn
    keys = _LES_IN_LEVELS + ("PS", "Rain") + (_LES_DIAG_LEVELS if diagnostics else ())
    if b.piecewise:
        for k in keys:                          # variable by variable: each is on the wire while the next is fetched
            b.les_in.writable(k)
            ens.get_profiles_batched((k,), {k: hn[k]})
            b.les_in.push(k, "h2d_les")
    else:
        for k in keys:
            b.les_in.writable(k)
        ens.get_profiles_batched(keys, {k: hn[k] for k in keys})
        for k in keys:
            b.les_in.push(k, "h2d_les")
    b.les_in.writable("A")
    ens.get_cloudfraction_batched(_index_map(batch), hn["A"])                 # spcpl.py:761-765
    b.les_in.push("A", "h2d_les")
    batch.profile_generation += 1
    prof = {k: hn[k] for k in keys + ("A",)}
    prof["_buffers"] = b
    return prof


# ---------------------------------------------------------------------------------------------
# backward: set_gcm_tendencies
# ---------------------------------------------------------------------------------------------
_BWD_KEYS = ("T", "QT", "QL", "QL_ice", "U", "V", "A")
_BWD_OUT = ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A")
_TEND_VARS = ("U", "V", "T", "SH", "QL", "QI", "A")                            # setter order of spcpl.py:535-542


def backward_batched(batch, profiles, dt_gcm, factor=1, conservative=False, wait=True):
    """K3 (K4 when ``conservative``) for every column of ``batch``; ``profiles``: dict of device tensors
    T,QT,QL,QL_ice,U,V [n x nL], A [n x nG] (+ Rhobf for conservative).  Zf is recomputed from the geopotential
    in-kernel (same arithmetic as the forward pass: no height round trip).  Returns dict of HOST arrays, views
    into the pinned download buffer filled by ONE device->host copy."""
    eng, b = batch.engine, batch.buf
    dt = float(_num(dt_gcm))
    out = {k: b.bwd_out.d[k] for k in _BWD_OUT}
    out["start_index"] = b.bwd_out.d["start_index"]
    prof = {k: profiles[k] for k in _BWD_KEYS + (("Rhobf",) if conservative else ())}
    plan = _plan(batch, "bwd", (bool(conservative),), lambda: eng.plan_backward(
        batch.gcm, batch.zf, prof, float(factor), dt, Zf=None, conservative=conservative, zh=batch.zh, out=out))
    plan.set_scalars(float(factor), dt)
    _launch(batch, plan, "k4" if conservative else "k3")
    b.bwd_out.pull(tuple("f_" + v for v in _TEND_VARS) + ("start_index",), "d2h_tendencies")
    if wait:
        b.bwd_out.ready()
    batch.bwd_rows = None
    return {k: b.bwd_out.hn[k] for k in _BWD_OUT + ("start_index",)}


def _ensure_backward(batch, les, profile, dt_gcm, factor, write, conservative):
    key = (batch.profile_generation, float(_num(dt_gcm)), float(factor), bool(conservative))
    if batch.bwd is None or batch.bwd_key != key:
        if profile is not None:
            batch.profiles[id(les)] = profile
        missing = [m for m in batch.les_models if id(m) not in batch.profiles]
        if missing:
            raise RuntimeError("set_gcm_tendencies: LES profiles of %d columns are unknown; call get_les_profiles() "
                               "for every LES first (as splib.step_les_models does)" % len(missing))
        src = lambda m: batch.profiles[id(m)]                # noqa: E731
        with_file = write and writer is not None
        keys = _BWD_KEYS + (("Rhobf",) if conservative else ())
        if with_file:
            keys = keys + tuple(k for k in ("THL", "presf", "Rhof", "Rhobf", "QR") if k not in keys)
        prof = batch.stack_profiles(keys, src)
        batch.bwd = backward_batched(batch, prof, dt_gcm, factor, conservative)
        batch.bwd_key = key
        if with_file:
            _write_backward(batch, prof)         # once per launch, for all columns
    batch.bwd_raw = (dt_gcm, factor, conservative, batch.profile_generation)
    return batch.bwd


def _backward_rows(batch):
    b = batch.bwd
    rows = [_rows_of(b["f_" + v]) for v in _TEND_VARS]
    if _unit_wrapper is not None:
        rows = [[_unit_wrapper("f_" + v, x) for x in r] for v, r in zip(_TEND_VARS, rows)]
    batch.bwd_rows = rows
    return rows


def set_gcm_tendencies(gcm, les, profile, dt_gcm, factor=1, write=True, conservative=False):
    """splib/spcpl.py:388-555. First call of a step: ONE launch for all columns; every call: the seven
    ``gcm.set_profile_tendency`` setters for column ``les`` (spcpl.py:535-542)."""
    try:
        batch, i = les._spc_batch, les._spc_row
    except AttributeError:
        batch = _batch_of(les)
        i = batch.index_of(les)
    raw = batch.bwd_raw
    if (raw is None or raw[0] is not dt_gcm or raw[1] is not factor or raw[2] is not conservative
            or raw[3] != batch.profile_generation):
        _ensure_backward(batch, les, profile, dt_gcm, factor, write, conservative)
    rows = batch.bwd_rows or _backward_rows(batch)
    setter, gi = gcm.set_profile_tendency, les.grid_index
    setter("U", gi, rows[0][i])                                               # spcpl.py:535-542
    setter("V", gi, rows[1][i])
    setter("T", gi, rows[2][i])
    setter("SH", gi, rows[3][i])
    setter("QL", gi, rows[4][i])
    setter("QI", gi, rows[5][i])
    setter("A", gi, rows[6][i])


def _write_backward(batch, prof):
    """spifs rows of set_gcm_tendencies for ALL columns (spcpl.py:412-425, 545-555). ``prof``: device tensors of
    the slab means incl. THL, presf, Rhof, Rhobf, QR."""
    b, n = batch.bwd, batch.n
    _inputs_ready(batch)
    h = _to_host
    with batch.engine.on_stream():
        d = batch.engine.diagnostics(batch.gcm, batch.zf, prof)                              # K5: t, ql_water
        dev = dict(u=h(prof["U"]), v=h(prof["V"]), presf=h(prof["presf"]), rhof=h(prof["Rhof"]),
                   rhobf=h(prof["Rhobf"]), qt=h(prof["QT"]), ql=h(prof["QL"]), ql_ice=h(prof["QL_ice"]),
                   ql_water=h(d["ql_water"]), thl=h(prof["THL"]), t=h(d["t"]), t_=h(prof["T"]), qr=h(prof["QR"]),
                   A_d=h(prof["A"])[:, ::-1])                                                # spcpl.py:404
    writer.write(f_U=b["f_U"], f_V=b["f_V"], f_T=b["f_T"], f_SH=b["f_SH"], f_QL=b["f_QL"], f_QI=b["f_QI"], f_A=b["f_A"],
                 A=batch.gcm_host["A"][:n], **dev)                                           # spcpl.py:550-551


def set_gcm_tendencies_batched(gcm, les_models, profiles, dt_gcm, factor=1, write=True, conservative=False):
    """Batched twin of the second ``for les`` loop of splib.step (splib/splib.py:330-332).  With an LES ensemble
    and a GCM offering ``set_profile_tendencies`` this is ONE launch, ONE download and seven setter calls."""
    if not len(les_models):
        return
    if _is_ensemble(les_models):
        batch = _batch_of(les_models)
        b = batch.buf
        if profiles is None or profiles.get("_buffers") is not b:
            raise RuntimeError("set_gcm_tendencies_batched: pass what get_les_profiles_batched() returned")
        if (write and writer is not None or conservative) and "Rhobf" not in profiles:
            raise RuntimeError("conservative coarsening / spifs output need get_les_profiles_batched(diagnostics=True)")
        batch.bwd = backward_batched(batch, b.les_in.d, dt_gcm, factor, conservative, wait=False)
        if not b.piecewise:
            b.bwd_out.ready()
        if hasattr(gcm, "set_profile_tendencies"):
            for var in _TEND_VARS:                                               # spcpl.py:535-542, each as it lands
                b.bwd_out.ready("f_" + var)
                gcm.set_profile_tendencies(var, les_models.grid_indices, _wrap("f_" + var, batch.bwd["f_" + var]))
        b.bwd_out.ready()
        if write and writer is not None:
            _write_backward(batch, b.les_in.d)
        if not hasattr(gcm, "set_profile_tendencies"):
            for i, gi in enumerate(les_models.grid_indices):
                for var in ("U", "V", "T", "SH", "QL", "QI", "A"):
                    gcm.set_profile_tendency(var, gi, _wrap("f_" + var, batch.bwd["f_" + var][i].copy()))
        return
    batch = _batch_of(les_models[0])
    for les in les_models:
        batch.profiles[id(les)] = profiles[les]
    for les in les_models:
        set_gcm_tendencies(gcm, les, None, dt_gcm, factor, write, conservative)


# ---------------------------------------------------------------------------------------------
# tendencies replayed from spifs: splib/spcpl.py:558-570 ("not used - was thought to be necessary for restarts")
# ---------------------------------------------------------------------------------------------
_FILE_TENDENCIES = None      # (path, record, model time) -> what was read, so a per-les loop opens the file once


def set_gcm_tendencies_from_file(gcm, les, path=None):
    """splib/spcpl.py:558-570: the seven GCM tendencies of column ``les`` taken from the spifs record nearest to the GCM's
    model time (as stored: float32) instead of computed.  ``path``: a spifs file written by ``spio.SpifsWriter`` (default:
    the active ``spcpl.writer``, flushed first).  The file is read once per (path, model time), not once per column."""
    global _FILE_TENDENCIES
    from . import spio
    if path is None:
        if writer is None:
            raise RuntimeError("set_gcm_tendencies_from_file: no spifs file (pass path= or install spcpl.writer)")
        writer.sync()
        path = writer.path
    t = float(_num(gcm.get_model_time()))
    if _FILE_TENDENCIES is None or _FILE_TENDENCIES[0] != (path, t):
        ti, tv, gi, data = spio.read_record_nearest(path, t, ["f_" + v for v in _TEND_VARS])
        log.info("set_gcm_tendencies_from_file() %s %d %s", t, ti, tv)
        _FILE_TENDENCIES = ((path, t), {int(g): c for c, g in enumerate(gi)}, data)
    _, col_of, data = _FILE_TENDENCIES
    c = col_of[int(les.grid_index)]
    for var in _TEND_VARS:                                                       # spcpl.py:564-570
        gcm.set_profile_tendency(var, les.grid_index, _wrap("f_" + var, data["f_" + var][c].copy()))


# ---------------------------------------------------------------------------------------------
# spinup diagnostics: splib/spcpl.py:574-609
# ---------------------------------------------------------------------------------------------
def write_les_profiles_batched(les_models):
    """Fetch the LES slab means of every column and write them to spifs (used during spinup,
    splib/splib.py:390): u, v, presf, qt, ql, ql_ice, ql_water, thl, t (from K5 with the GCM pressures,
    spcpl.py:593-594), t_, qr.  One getter round per column (RPC, as in the reference), ONE kernel launch
    and one write per variable for all columns.  Returns the dict of host arrays it wrote."""
    if not len(les_models):
        return {}
    keys = ("U", "V", "presf", "THL", "QT", "QL", "QL_ice", "QR", "T")
    if _is_ensemble(les_models):
        batch = _batch_of(les_models)
        hn = batch.buf.les_in.hn
        for k in keys:
            batch.buf.les_in.writable(k)
        les_models.get_profiles_batched(keys, {k: hn[k] for k in keys})
        prof = batch.upload_profiles(keys)
    else:
        batch = _batch_of(les_models[0])
        src = lambda m: {"U": m.get_profile_U(), "V": m.get_profile_V(), "presf": m.get_presf(),      # noqa: E731
                         "THL": m.get_profile_THL(), "QT": m.get_profile_QT(), "QL": m.get_profile_QL(),
                         "QL_ice": m.get_profile_QL_ice(), "QR": m.get_profile_QR(), "T": m.get_profile_T()}
        prof = batch.stack_profiles(keys, src)
    _inputs_ready(batch)
    h = _to_host
    with batch.engine.on_stream():
        d = batch.engine.diagnostics(batch.gcm, batch.zf, prof)                              # K5: t, ql_water
        out = dict(u=h(prof["U"]), v=h(prof["V"]), presf=h(prof["presf"]), qt=h(prof["QT"]), ql=h(prof["QL"]),
                   ql_ice=h(prof["QL_ice"]), ql_water=h(d["ql_water"]), thl=h(prof["THL"]), t=h(d["t"]), t_=h(prof["T"]),
                   qr=h(prof["QR"]))
    if writer is not None:
        writer.write(**out)
    return out


def write_les_profiles(les):
    """splib/spcpl.py:574-609 for one column (computes the whole batch on the first call of a step)."""
    batch = _batch_of(les)
    key = ("wlp", batch.profile_generation, id(batch.fwd))
    if getattr(batch, "wlp_key", None) != key:
        batch.wlp = write_les_profiles_batched(batch.les_models)
        batch.wlp_key = key
    i = batch.index_of(les)
    return {k: v[i] for k, v in batch.wlp.items()}


# ---------------------------------------------------------------------------------------------
# variability nudge: splib/spcpl.py:613-744 (qt_forcing == 'variance')
# ---------------------------------------------------------------------------------------------
VN_ERR_SIGN, VN_ERR_CONV = 256, 512          # status bits of spc_variability_nudge_f64 (include/spc.h)


def variability_nudge(les, DT, constantT=False, write=True):
    """splib/spcpl.py:613-744 for one LES (the reference's signature): see variability_nudge_batched."""
    return variability_nudge_batched([les], DT, constantT, write)[0]


VN_MAX_COLS = 32767          # columns per launch of spc_variability_nudge_f64 (include/spc.h: the grid's y extent)


def _vnudge_chunk(eng, n, itot, jtot, ktot, constantT):
    """columns per launch on ONE engine: the kernel's limit, and what fits its device -- fields (qt, qsat; thl, ql with
    constantT), the transposed-plane workspace, R and the profiles, within 80 % of the memory that is available now: what the
    driver reports free PLUS what torch's caching allocator holds without using it (after a few coupled steps most of the
    free memory sits there).  The reference loops over any number of LES (splib/spcpl.py:377-382); columns are independent, so
    chunks give the bits of one launch."""
    chunk = min(int(VN_MAX_COLS), n)
    dev = getattr(eng, "device", None)
    if dev is not None and dev.type == "cuda":
        lib = getattr(eng, "lib", None)
        per_col = (4 if constantT else 2) * itot * jtot * ktot * 8 + itot * jtot * 8 + 8 * ktot * 8
        if lib is not None:
            per_col += max(0, int(lib.spc_vnudge_workspace_bytes(1, itot, jtot, ktot)))
        free, _ = torch.cuda.mem_get_info(dev)
        free += max(0, torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev))
        chunk = max(1, min(chunk, int(0.8 * free) // max(per_col, 1)))
        if chunk < n:
            log.info("sp_coupler_amd: variability nudge of %d LES on %s in launches of %d (%.1f GiB available, %.2f GiB per LES)",
                     n, dev, chunk, free / 2.0 ** 30, per_col / 2.0 ** 30)
    return chunk


def _vnudge_launch(F, Rs, constantT):
    """stacked host arrays -> as few launches of K6 as the kernel's column limit and the device memory allow -> host results;
    ``F``: dict of [n x ...] arrays.  With several engines (multi.MultiDeviceEngine) the LES are dealt out in contiguous row
    blocks, one per device (sharding.shard_bounds, as every other array of the batch), and each device nudges ITS LES: the 3-D
    fields -- the largest objects in the system -- never meet on one card.  Chunks are issued in rounds, one chunk per device
    and round: uploads and launches of a round go out device by device (asynchronous), then the results come back."""
    from .sharding import shard_bounds
    eng = get_engine()
    engines = list(getattr(eng, "engines", None) or [eng])
    n, itot, jtot, ktot = F["qt"].shape
    k = max(1, min(len(engines), n))
    bounds = shard_bounds(n, k)
    queues = []                                                # per device: its chunks (lo, hi) in row order
    for d in range(k):
        lo, hi = bounds[d], bounds[d + 1]
        chunk = _vnudge_chunk(engines[d], hi - lo, itot, jtot, ktot, constantT) if hi > lo else 1
        queues.append([(c, min(hi, c + chunk)) for c in range(lo, hi, chunk)])
    host = None
    qt_out = numpy.empty_like(F["qt"])
    thl_out = numpy.empty_like(F["thl"]) if constantT else None
    for rnd in range(max(len(q) for q in queues)):
        live = []
        for d, q in enumerate(queues):
            if rnd >= len(q):
                continue
            e, (lo, hi) = engines[d], q[rnd]
            up = lambda a, e=e: torch.from_numpy(numpy.ascontiguousarray(a)).to(e.device, e.dtype)      # noqa: E731
            with e.on_stream():
                T = {key: up(v[lo:hi]) for key, v in F.items() if v is not None}
                res = e.variability_nudge(T["qt"], T["qsat"], up(Rs[lo:hi]), T["ql_av"], T["qt_av"], T["ql_ref"], presf=T["presf"],
                                          thl=T.get("thl"), ql=T.get("ql"), constantT=constantT)
            live.append((e, lo, hi, T, res))
        for e, lo, hi, T, res in live:
            with e.on_stream():
                if host is None:
                    host = {key: numpy.empty((n,) + tuple(v.shape[1:]), dtype=torch.empty(0, dtype=v.dtype).numpy().dtype) for key, v in res.items()}
                for key, v in res.items():
                    host[key][lo:hi] = v.cpu().numpy()
                qt_out[lo:hi] = T["qt"].cpu().numpy()
                if constantT:
                    thl_out[lo:hi] = T["thl"].cpu().numpy()
        del live
    return host, qt_out, thl_out


def _vnudge_finish(host, rows, dtv, write):
    """alpha = log(beta) / DT (spcpl.py:739), the spifs rows (spcpl.py:742-744) and scipy's exceptions"""
    out = []
    for i in range(host["beta"].shape[0]):
        beta = host["beta"][i]
        out.append(dict(beta=beta, alpha=numpy.log(beta) / dtv, qt_std=host["qt_std"][i], a=host["a"][i], status=host["status"][i]))
    if write and writer is not None:                                          # spcpl.py:742-744
        writer.write(rows=rows, qt_alpha=numpy.stack([o["alpha"] for o in out]),
                     qt_beta=numpy.stack([o["beta"] for o in out]), qt_std=numpy.stack([o["qt_std"] for o in out]))
    st = host["status"]
    if (st & VN_ERR_SIGN).any():
        i, k = numpy.argwhere((st & VN_ERR_SIGN) != 0)[0]
        raise ValueError("f(a) and f(b) must have different signs (variability nudge, LES %d level %d)" % (i, k))
    if (st & VN_ERR_CONV).any():
        i, k = numpy.argwhere((st & VN_ERR_CONV) != 0)[0]
        raise RuntimeError("Failed to converge after 100 iterations. (variability nudge, LES %d level %d)" % (i, k))
    return out


def variability_nudge_batched(les_models, DT, constantT=False, write=True):
    """spcpl.variability_nudge (splib/spcpl.py:613-744) for every LES in ONE launch.  Per LES, as the reference:
    a zero-mean Gaussian field R from numpy's GLOBAL generator (spcpl.py:620-621, drawn in les order), the 3-D
    fields ``les.get_field("Qsat"/"QT")`` (+ "THL", "QL" with constantT), the slab means ``les.get_profile("QL"/"QT")``,
    ``les.get_presf()`` and ``les.ql_ref``; afterwards ``les.fields.QT`` (and ``.THL``) are set and qt_alpha, qt_beta,
    qt_std written to spifs.  Returns a list of dicts (beta, alpha, qt_std, a, status) per LES.  Where scipy's
    brentq would raise (no sign change for the additive noise / no convergence) this raises the same exception
    type AFTER the launch; levels that were fine have been applied."""
    les_models = list(les_models)
    if not les_models:
        return []
    dtv = float(_num(DT))
    Rs, F = [], {k: [] for k in ("qsat", "qt", "ql_av", "qt_av", "presf", "ql_ref", "thl", "ql")}
    for les in les_models:
        itot, jtot = int(les.get_itot()), int(les.get_jtot())
        R = numpy.random.normal(size=(itot, jtot))                            # spcpl.py:620
        R -= R.sum() / (itot * jtot)                                          # spcpl.py:621
        Rs.append(R)
        F["qsat"].append(_num(les.get_field("Qsat")))                         # spcpl.py:627-632
        F["qt"].append(_num(les.get_field("QT")))
        F["ql_av"].append(_num(les.get_profile("QL")))
        F["qt_av"].append(_num(les.get_profile("QT")))
        F["presf"].append(_num(les.get_presf()))
        F["ql_ref"].append(_num(les.ql_ref))
        if constantT:                                                         # spcpl.py:634-636
            F["thl"].append(_num(les.get_field("THL")))
            F["ql"].append(_num(les.get_field("QL")))
    shapes = {a.shape for a in F["qt"]} | {a.shape for a in F["qsat"]}
    if len(shapes) != 1:
        raise ValueError("variability_nudge_batched: one launch handles LES instances of ONE field shape, got %s; "
                         "call it per group of equal shapes" % sorted(shapes))
    host, qt_new, thl_new = _vnudge_launch({k: (numpy.stack(v) if v else None) for k, v in F.items()}, numpy.stack(Rs), constantT)
    for i, les in enumerate(les_models):
        target = les.fields if hasattr(les, "fields") else None
        if target is not None:
            target.QT = _wrap("qt", qt_new[i])                                # spcpl.py:735
            if constantT:
                target.THL = _wrap("thl", thl_new[i])                         # spcpl.py:736-737
        else:
            les.set_field("QT", _wrap("qt", qt_new[i]))
            if constantT:
                les.set_field("THL", _wrap("thl", thl_new[i]))
    rows = [_batch_of(les).index_of(les) for les in les_models] if (write and writer is not None) else None
    return _vnudge_finish(host, rows, dtv, write)


def variability_nudge_ensemble(ens, DT, constantT=False, write=True):
    """The same for an LES ENSEMBLE (batched model protocol, sp_coupler_amd.models): the 3-D fields of all columns come
    from ``ens.get_fields_batched(name) -> [n x itot x jtot x ktot]`` ("Qsat", "QT"; "THL", "QL" with constantT) and go
    back through ``ens.set_fields_batched(name, array)``; slab means from ``get_profiles_batched``; ``ens.ql_ref`` is
    what set_les_forcings_batched stored (spcpl.py:348).  R is drawn per column in column order like the per-LES loop."""
    dtv = float(_num(DT))
    n = len(ens)
    qt = _num(ens.get_fields_batched("QT"))
    _, itot, jtot, ktot = qt.shape
    Rs = numpy.empty((n, itot, jtot))
    for i in range(n):
        R = numpy.random.normal(size=(itot, jtot))                            # spcpl.py:620
        R -= R.sum() / (itot * jtot)                                          # spcpl.py:621
        Rs[i] = R
    av = {k: numpy.empty((n, ktot)) for k in ("QL", "QT", "presf")}
    ens.get_profiles_batched(("QL", "QT", "presf"), av)
    F = dict(qsat=_num(ens.get_fields_batched("Qsat")), qt=qt, ql_av=av["QL"], qt_av=av["QT"], presf=av["presf"],
             ql_ref=_num(ens.ql_ref), thl=_num(ens.get_fields_batched("THL")) if constantT else None,
             ql=_num(ens.get_fields_batched("QL")) if constantT else None)
    host, qt_new, thl_new = _vnudge_launch(F, Rs, constantT)
    ens.set_fields_batched("QT", _wrap("qt", qt_new))                         # spcpl.py:735
    if constantT:
        ens.set_fields_batched("THL", _wrap("thl", thl_new))                  # spcpl.py:736-737
    return _vnudge_finish(host, list(range(n)), dtv, write)


# ---------------------------------------------------------------------------------------------
# diagnostics for non-SP output columns: splib/spcpl.py:251-267
# ---------------------------------------------------------------------------------------------
def output_column_conversion(profile):
    """Adds Tv, Zh, Zf, Psurf, THL, QT to ``profile`` (dict of [n x nG] arrays, names as in the
    reference's netCDF mapping: T, SH, QL, QI, Pf, Ph, Zgfull, Zghalf). Works on one column ([nG]) or
    many ([n x nG]); computed by the diagnostics kernel K5."""
    eng = get_engine()
    one = numpy.asarray(_num(profile["T"])).ndim == 1
    g = {}
    with eng.on_stream():
        for src, dst in (("T", "T"), ("SH", "SH"), ("QL", "QL"), ("QI", "QI"), ("Pf", "Pfull"), ("Zgfull", "Zgfull"),
                         ("Zghalf", "Zghalf")):
            a = numpy.atleast_2d(_num(profile[src]))
            g[dst] = eng.to_devices(numpy.ascontiguousarray(a, dtype=numpy.float64), rows=a.shape[0])
        d = {k: _to_host(v) for k, v in eng.diagnostics(g).items()}
    sel = (lambda a: a[0]) if one else (lambda a: a)
    Ph = numpy.atleast_2d(_num(profile["Ph"]))
    profile["Tv"] = sel(d["Tv"])
    profile["Zh"] = sel(d["Zh"][:, 1:])                                      # spcpl.py:261
    profile["Zf"] = sel(d["Zf"])
    profile["Psurf"] = sel(Ph[:, -1])                                        # spcpl.py:263
    profile["Ph"] = sel(Ph[:, 1:])                                           # spcpl.py:264
    profile["THL"] = sel(d["THL"])
    profile["QT"] = sel(d["QT"])
    return profile
