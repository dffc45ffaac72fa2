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

from .engine import Engine

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
    global _engine
    if _engine is None:
        _engine = Engine()
    return _engine


def set_engine(engine):
    global _engine
    _engine = engine


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


def _result(x):
    """value of an async request (``.result()``) or the value itself"""
    return x.result() if hasattr(x, "result") and callable(x.result) else x


class ColumnBatch:
    """All SP columns of one GCM step, resident in HBM as [n_cols x n_lev] tensors."""

    def __init__(self, engine, les_models, gcm_host, surf_host, extra_cols):
        self.engine = engine
        self.les_models = list(les_models)
        self.n = len(self.les_models)
        self.row = {id(les): i for i, les in enumerate(self.les_models)}
        self.extra_cols = list(extra_cols)
        self.gcm_host = gcm_host          # dict var -> ndarray [n_total x n], rows n.. are the extra columns
        self.surf_host = surf_host
        dev = engine.device
        self.gcm = {k: torch.from_numpy(numpy.ascontiguousarray(v[:self.n])).to(dev, engine.dtype)
                    for k, v in gcm_host.items()} if self.n else {}
        for k, v in surf_host.items():
            self.gcm[k] = torch.from_numpy(numpy.ascontiguousarray(v[:self.n])).to(dev, engine.dtype)
        self.zf = self.zh = None
        self.zf_host = self.zh_host = None
        self.profiles = {}                # id(les) -> profile dict (values or async requests)
        self.profile_generation = 0
        self.fwd = None                   # host results of the last forward launch
        self.fwd_key = None
        self.bwd = None
        self.bwd_key = None
        if self.n:
            self._pack_les_grid()

    def _pack_les_grid(self):
        """les.zf_cache / les.zh_cache (splib/splib.py:152-153): one shared [nL] grid when all LES
        instances agree (the normal case), else [n x nL]."""
        zfs = [_num(les.zf_cache) for les in self.les_models]
        zhs = [_num(les.zh_cache) for les in self.les_models]
        shared = all(z.shape == zfs[0].shape and numpy.array_equal(z, zfs[0]) for z in zfs) and \
            all(numpy.array_equal(z, zhs[0]) for z in zhs)
        self.zf_host = zfs[0] if shared else numpy.stack(zfs)
        self.zh_host = zhs[0] if shared else numpy.stack(zhs)
        dev, dt = self.engine.device, self.engine.dtype
        self.zf = torch.from_numpy(numpy.ascontiguousarray(self.zf_host)).to(dev, dt)
        self.zh = torch.from_numpy(numpy.ascontiguousarray(self.zh_host)).to(dev, dt)

    def index_of(self, les):
        return self.row[id(les)]

    # ---- LES slab means -> device -------------------------------------------------------------
    def stack_profiles(self, keys, source):
        """source(les) -> dict; returns dict key -> device tensor ([n x nL], [n x nG] for A, [n] scalars)"""
        rows = [source(les) for les in self.les_models]
        out = {}
        for k in keys:
            arr = numpy.stack([_num(_result(r[k])) for r in rows])
            out[k] = torch.from_numpy(numpy.ascontiguousarray(arr)).to(self.engine.device, self.engine.dtype)
        return out


_current = None


def current_batch():
    return _current


def _batch_of(les):
    b = getattr(les, "_spc_batch", None)
    if b is None:
        raise RuntimeError("gather_gcm_data() must be called before the per-les coupling functions")
    return b


# ---------------------------------------------------------------------------------------------
# gather: splib/spcpl.py:55-132
# ---------------------------------------------------------------------------------------------
def gather_gcm_data(gcm, les_models, couple_surface, output_column_indices=None, write=True, attach_rows=False):
    """Pull all SP columns' profiles with ONE ``gcm.get_profile_fields`` call per variable (as the
    reference does, spcpl.py:66) and pack them into HBM.  The reference then scatters rows onto the
    ``les`` objects (spcpl.py:81-86); here each ``les`` gets a handle to the batch instead
    (``attach_rows=True`` also sets the per-variable row views for code that reads ``les.T`` etc.)."""
    global _current
    extra_cols = [] if output_column_indices is None else list(output_column_indices)
    cols = [les.grid_index for les in les_models] + extra_cols
    start = time.time()
    profile_data, surface_data = {}, {}
    empty = not any(cols)                                                    # quirk kept: spcpl.py:63,71
    for v in gcm_vars:
        profile_data[v] = [] if empty else _num(gcm.get_profile_fields(v, cols))
    if couple_surface:
        for v in surf_vars:
            surface_data[v] = [] if empty else _num(gcm.get_surface_field(v, cols))
    log.info("Fetching gcm data took %d s" % (time.time() - start))
    if empty:
        _current = None
        return None
    batch = ColumnBatch(get_engine(), les_models, profile_data, surface_data, extra_cols)
    if _current is not None:   # slab means fetched by get_les_profiles() after the previous step's LES run
        batch.profiles = {k: v for k, v in _current.profiles.items() if k in batch.row}
    for i, les in enumerate(les_models):
        les._spc_batch = batch
        if attach_rows:
            for v in gcm_vars:
                setattr(les, v, profile_data[v][i][:])
            for v in surface_data:
                setattr(les, v, surface_data[v][i])
    # extra output columns: spcpl.py:89-129
    if extra_cols and write and writer is not None:
        n0 = len(les_models)
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


def _first_step_profile(les):
    """splib/spcpl.py:302-308, 321: live getters on the first step"""
    return {"U": les.get_profile_U(), "V": les.get_profile_V(), "THL": les.get_profile_THL(),
            "QT": les.get_profile_QT(), "QL": les.get_profile_QL(), "PS": les.get_surface_pressure(),
            "Rain": les.get_rain()}


def forward_batched(batch, profiles, dt_gcm, factor, couple_surface=False, want_profiles=True):
    """K1 (+fused K2) for every column of ``batch``. ``profiles``: dict of device tensors U,V,THL,QT,QL
    [n x nL], PS [n] (+Rain, rain_last [n]). Returns dict of HOST arrays (one D2H per output)."""
    eng = batch.engine
    dt = float(_num(dt_gcm))
    res = eng.forward(batch.gcm, batch.zf, profiles, float(factor), dt, zh=batch.zh, want_profiles=want_profiles,
                      want_heights=True, couple_surface=couple_surface)
    batch.dev_fwd = res
    host = {k: v.cpu().numpy() for k, v in res.items()}     # .cpu() synchronises with the launch stream
    if "ql_ref" in host:
        host["ql"] = host["ql_ref"]
    return host


def _ensure_forward(batch, les, firststep, profile, dt_gcm, factor, couple_surface):
    key = (batch.profile_generation, bool(firststep), float(_num(dt_gcm)), float(factor), bool(couple_surface))
    if batch.fwd is not None and batch.fwd_key == key:
        return batch.fwd
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
    prof["rain_last"] = torch.from_numpy(rain_last).to(batch.engine.device, batch.engine.dtype)
    batch.fwd = forward_batched(batch, prof, dt_gcm, factor, couple_surface)
    batch.fwd["rain"] = prof["Rain"].cpu().numpy()
    batch.fwd_key = key
    batch.fwd_written = False
    return batch.fwd


def _write_forward(batch):
    """spifs rows of convert_profiles + set_les_forcings for ALL columns (spcpl.py:230-244, 352-376)."""
    f, g = batch.fwd, batch.gcm_host
    n = batch.n
    d = {k: v.cpu().numpy() for k, v in batch.engine.diagnostics(batch.gcm).items()}     # K5: Tv, THL, QT
    writer.write(U=g["U"][:n], V=g["V"][:n], T=g["T"][:n], SH=g["SH"][:n], QL=g["QL"][:n], QI=g["QI"][:n],
                 Pf=g["Pfull"][:n], Ph=g["Phalf"][:n, 1:], Zf=f["Zf"], Zh=f["Zh"][:, 1:], Psurf=g["Phalf"][:n, -1],
                 Tv=d["Tv"], THL=d["THL"], QT=d["QT"], f_u=f["f_u"], f_v=f["f_v"], f_thl=f["f_thl"], f_qt=f["f_qt"],
                 rain=f["rain"], rainrate=f["rainrate"] * 3600)                              # spcpl.py:358
    if "wthl" in f:
        s = batch.surf_host
        writer.write(z0m=f["z0m"], z0h=f["z0h"], wthl=f["wthl"], wqt=f["wqt"], TLflux=s["TLflux"][:n],
                     TSflux=s["TSflux"][:n], SHflux=s["SHflux"][:n], QLflux=s["QLflux"][:n], QIflux=s["QIflux"][:n])
    batch.fwd_written = True


def convert_profiles(les, write=True):
    """splib/spcpl.py:171-246: (u, v, thl, qt, ps, ql) for one column; caches les.gcm_Zf / gcm_Zh."""
    batch = _batch_of(les)
    i = batch.index_of(les)
    if getattr(batch, "conv", None) is None:
        eng = batch.engine
        nL = batch.zf.shape[-1]
        z = torch.zeros(batch.n, nL, device=eng.device, dtype=eng.dtype)
        dummy = {"U": z, "V": z, "THL": z, "QT": z, "QL": z, "PS": torch.zeros(batch.n, device=eng.device, dtype=eng.dtype)}
        res = eng.forward(batch.gcm, batch.zf, dummy, 0.0, 1.0, want_profiles=True, want_heights=True)
        batch.conv = {k: res[k].cpu().numpy() for k in ("u", "v", "thl", "qt", "ps", "ql_ref", "Zf", "Zh")}
    c = batch.conv
    les.gcm_Zf = _wrap("Zf", c["Zf"][i])                                     # spcpl.py:200
    les.gcm_Zh = _wrap("Zh", c["Zh"][i])                                     # spcpl.py:201
    return (_wrap("u", c["u"][i]), _wrap("v", c["v"][i]), _wrap("thl", c["thl"][i]), _wrap("qt", c["qt"][i]),
            _wrap("ps", c["ps"][i]), _wrap("ql", c["ql_ref"][i]))


def set_les_forcings(les, gcm, asynchronous, firststep, profile, dt_gcm, factor, couple_surface, qt_forcing='sp',
                     write=True, variability_nudge_constant_T=False):
    """splib/spcpl.py:299-385. The first call of a step computes the forcings of ALL columns in one
    launch; this call then pushes column ``les``'s rows to its setters and returns the request dict."""
    if qt_forcing == 'variance':
        raise NotImplementedError("variability_nudge (splib/spcpl.py:613-744) is outside the hot path (SURVEY 8(f4))")
    batch = _batch_of(les)
    f = _ensure_forward(batch, les, firststep, profile, dt_gcm, factor, couple_surface)
    i = batch.index_of(les)
    les.gcm_Zf, les.gcm_Zh = _wrap("Zf", f["Zf"][i]), _wrap("Zh", f["Zh"][i])   # spcpl.py:200-201
    les.rain = f["rain"][i]                                                  # spcpl.py:324
    req = {
        "U": les.set_tendency_U(_wrap("f_u", f["f_u"][i]), return_request=asynchronous),                 # :341
        "V": les.set_tendency_V(_wrap("f_v", f["f_v"][i]), return_request=asynchronous),                 # :342
        "THL": les.set_tendency_THL(_wrap("f_thl", f["f_thl"][i]), return_request=asynchronous),         # :343
        "QT": les.set_tendency_QT(_wrap("f_qt", f["f_qt"][i]), return_request=asynchronous),             # :344
        "SP": les.set_tendency_surface_pressure(_wrap("f_ps", f["f_ps"][i]), return_request=asynchronous),  # :345
        "QL": les.set_tendency_QL(_wrap("f_ql", f["f_ql"][i]), return_request=asynchronous),             # :346
        "QLp": les.set_ref_profile_QL(_wrap("ql_ref", f["ql_ref"][i]), return_request=asynchronous),     # :347
    }
    les.ql_ref = _wrap("ql_ref", f["ql_ref"][i])                             # spcpl.py:348
    if write and writer is not None and not batch.fwd_written:
        _write_forward(batch)                    # once per launch, for all columns
    if couple_surface:                                                       # spcpl.py:359-364
        req["Z0M_surf"] = les.set_z0m_surf(_wrap("z0m", f["z0m"][i]), return_request=asynchronous)
        req["Z0H_surf"] = les.set_z0h_surf(_wrap("z0h", f["z0h"][i]), return_request=asynchronous)
        req["WT_surf"] = les.set_wt_surf(_wrap("wthl", f["wthl"][i]), return_request=asynchronous)
        req["WQ_surf"] = les.set_wq_surf(_wrap("wqt", f["wqt"][i]), return_request=asynchronous)
    return req


def set_les_forcings_batched(les_models, gcm, asynchronous, firststep, profiles, dt_gcm, factor, couple_surface,
                             qt_forcing='sp', write=True):
    """Batched twin: the whole ``for les in les_models`` loop of splib.step (splib/splib.py:317-323).
    ``profiles``: dict les -> profile dict (ignored on the first step). Returns list of request dicts."""
    if not les_models:
        return []
    batch = _batch_of(les_models[0])
    if not firststep:
        for les in les_models:
            batch.profiles[id(les)] = profiles[les]
    return [set_les_forcings(les, gcm, asynchronous, firststep, None if firststep else profiles[les], dt_gcm, factor,
                             couple_surface, qt_forcing, write) for les in les_models]


def convert_surface_fluxes(les):
    """splib/spcpl.py:136-167: (z0m, z0h, wthl, wqt). ``les`` is an LES object of the current batch (values
    come from the fused forward launch) or, as in the reference, a dict of GCM data for columns without
    an LES (keys Z0M, Z0H, QLflux, QIflux, SHflux, TSflux, Ph, T; one column or [n x ...])."""
    if isinstance(les, dict):
        eng = get_engine()
        Ph, T = numpy.atleast_2d(_num(les["Ph"])), numpy.atleast_2d(_num(les["T"]))   # KeyError if missing, spcpl.py:146
        one = numpy.ndim(_num(les["T"])) == 1
        dev = lambda a: torch.from_numpy(numpy.ascontiguousarray(numpy.atleast_1d(_num(a)))).to(eng.device, eng.dtype)  # noqa: E731
        wthl, wqt = eng.surface_fluxes(dev(Ph[:, -1]), dev(T[:, -1]), dev(les["QLflux"]), dev(les["QIflux"]),
                                       dev(les["SHflux"]), dev(les["TSflux"]))
        wthl, wqt = wthl.cpu().numpy(), wqt.cpu().numpy()
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
def cloud_fraction_indices(les):
    """indices = searchsorted(zh, Zh, side='right')[:-1][::-1]  (splib/spcpl.py:26 / 764), from K2"""
    batch = _batch_of(les)
    i = batch.index_of(les)
    if batch.fwd is not None and "idx" in batch.fwd:
        return batch.fwd["idx"][i]
    if getattr(batch, "idx_host", None) is None:
        if getattr(batch, "conv", None) is None:
            convert_profiles(les, write=False)
        Zh = torch.from_numpy(batch.conv["Zh"]).to(batch.engine.device, batch.engine.dtype)
        batch.idx_host = batch.engine.cloud_indices(batch.zh, Zh).cpu().numpy()
    return batch.idx_host[i]


def get_cloud_fraction(les):
    """splib/spcpl.py:22-29"""
    indices = cloud_fraction_indices(les)
    return les.get_cloudfraction(indices)[::-1]


def get_les_profiles(les, asynchronous):
    """splib/spcpl.py:747-767: 14 getters; the index map comes from the GPU (K2). The returned dict is
    also remembered so that the next forward / backward launch can batch all columns."""
    indices = cloud_fraction_indices(les)
    prof = {"U": les.get_profile_U(return_request=asynchronous), "V": les.get_profile_V(return_request=asynchronous),
            "presf": les.get_presf(return_request=asynchronous), "Rhof": les.get_rhof(return_request=asynchronous),
            "Rhobf": les.get_rhobf(return_request=asynchronous), "THL": les.get_profile_THL(return_request=asynchronous),
            "QT": les.get_profile_QT(return_request=asynchronous), "QL": les.get_profile_QL(return_request=asynchronous),
            "QL_ice": les.get_profile_QL_ice(return_request=asynchronous),
            "QR": les.get_profile_QR(return_request=asynchronous),
            "PS": les.get_surface_pressure(return_request=asynchronous), "T": les.get_profile_T(return_request=asynchronous),
            "A": les.get_cloudfraction(indices, return_request=asynchronous),
            "Rain": les.get_rain(return_request=asynchronous)}
    batch = _batch_of(les)
    if id(les) in batch.profiles and len(batch.profiles) >= batch.n:
        batch.profiles = {}
    if not batch.profiles:
        batch.profile_generation += 1
    batch.profiles[id(les)] = prof
    return prof


# ---------------------------------------------------------------------------------------------
# backward: set_gcm_tendencies
# ---------------------------------------------------------------------------------------------
_BWD_KEYS = ("T", "QT", "QL", "QL_ice", "U", "V", "A")


def backward_batched(batch, profiles, dt_gcm, factor=1, conservative=False):
    """K3 for every column of ``batch``; ``profiles``: dict of device tensors T,QT,QL,QL_ice,U,V [n x nL],
    A [n x nG] (+ Rhobf for conservative). Returns dict of HOST arrays."""
    eng = batch.engine
    Zf = batch.dev_fwd["Zf"] if getattr(batch, "dev_fwd", None) is not None and "Zf" in batch.dev_fwd else None
    res = eng.backward(batch.gcm, batch.zf, profiles, float(factor), float(_num(dt_gcm)), Zf=Zf,
                       conservative=conservative, zh=batch.zh)
    return {k: v.cpu().numpy() for k, v in res.items()}


def set_gcm_tendencies(gcm, les, profile, dt_gcm, factor=1, write=True, conservative=False):
    """splib/spcpl.py:388-555. First call of a step: ONE launch for all columns; every call: the seven
    ``gcm.set_profile_tendency`` setters for column ``les`` (spcpl.py:535-542)."""
    batch = _batch_of(les)
    key = (batch.profile_generation, float(_num(dt_gcm)), float(factor), bool(conservative))
    if batch.bwd is None or batch.bwd_key != key:
        if profile is not None:
            batch.profiles[id(les)] = profile
        missing = [m for m in batch.les_models if id(m) not in batch.profiles]
        if missing:
            raise RuntimeError("set_gcm_tendencies: LES profiles of %d columns are unknown; call get_les_profiles() "
                               "for every LES first (as splib.step_les_models does)" % len(missing))
        keys = _BWD_KEYS + (("Rhobf",) if conservative else ())
        prof = batch.stack_profiles(keys, lambda m: batch.profiles[id(m)])
        batch.bwd = backward_batched(batch, prof, dt_gcm, factor, conservative)
        batch.bwd_key = key
        if write and writer is not None:
            _write_backward(batch, prof)         # once per launch, for all columns
    b = batch.bwd
    i = batch.index_of(les)
    gcm.set_profile_tendency("U", les.grid_index, _wrap("f_U", b["f_U"][i]))     # spcpl.py:535
    gcm.set_profile_tendency("V", les.grid_index, _wrap("f_V", b["f_V"][i]))
    gcm.set_profile_tendency("T", les.grid_index, _wrap("f_T", b["f_T"][i]))
    gcm.set_profile_tendency("SH", les.grid_index, _wrap("f_SH", b["f_SH"][i]))
    gcm.set_profile_tendency("QL", les.grid_index, _wrap("f_QL", b["f_QL"][i]))
    gcm.set_profile_tendency("QI", les.grid_index, _wrap("f_QI", b["f_QI"][i]))
    gcm.set_profile_tendency("A", les.grid_index, _wrap("f_A", b["f_A"][i]))     # spcpl.py:542


def _write_backward(batch, prof):
    """spifs rows of set_gcm_tendencies for ALL columns (spcpl.py:412-425, 545-555)."""
    b, n = batch.bwd, batch.n
    src = lambda m: batch.profiles[id(m)]        # noqa: E731
    extra = batch.stack_profiles(("THL", "presf", "Rhof", "Rhobf", "QR"), src)
    dprof = dict(prof, THL=extra["THL"])
    d = batch.engine.diagnostics(batch.gcm, batch.zf, dprof)                                 # K5: t, ql_water
    h = lambda t: t.cpu().numpy()                # noqa: E731
    writer.write(u=h(prof["U"]), v=h(prof["V"]), presf=h(extra["presf"]), rhof=h(extra["Rhof"]),
                 rhobf=h(extra["Rhobf"]), qt=h(prof["QT"]), ql=h(prof["QL"]), ql_ice=h(prof["QL_ice"]),
                 ql_water=h(d["ql_water"]), thl=h(extra["THL"]), t=h(d["t"]), t_=h(prof["T"]), qr=h(extra["QR"]),
                 f_U=b["f_U"], f_V=b["f_V"], f_T=b["f_T"], f_SH=b["f_SH"], f_QL=b["f_QL"], f_QI=b["f_QI"], f_A=b["f_A"],
                 A=batch.gcm_host["A"][:n], A_d=h(prof["A"])[:, ::-1])                       # spcpl.py:404,550-551


def set_gcm_tendencies_batched(gcm, les_models, profiles, dt_gcm, factor=1, write=True, conservative=False):
    """Batched twin of the second ``for les`` loop of splib.step (splib/splib.py:330-332)."""
    if not les_models:
        return
    batch = _batch_of(les_models[0])
    for les in les_models:
        batch.profiles[id(les)] = profiles[les]
    for les in les_models:
        set_gcm_tendencies(gcm, les, None, dt_gcm, factor, write, conservative)


# ---------------------------------------------------------------------------------------------
# spinup diagnostics: splib/spcpl.py:574-609
# ---------------------------------------------------------------------------------------------
def write_les_profiles_batched(les_models):
    """Fetch the LES slab means of every column and write them to spifs (used during spinup,
    splib/splib.py:390): u, v, presf, qt, ql, ql_ice, ql_water, thl, t (from K5 with the GCM pressures,
    spcpl.py:593-594), t_, qr.  One getter round per column (RPC, as in the reference), ONE kernel launch
    and one write per variable for all columns.  Returns the dict of host arrays it wrote."""
    if not les_models:
        return {}
    batch = _batch_of(les_models[0])
    src = lambda m: {"U": m.get_profile_U(), "V": m.get_profile_V(), "presf": m.get_presf(),      # noqa: E731
                     "THL": m.get_profile_THL(), "QT": m.get_profile_QT(), "QL": m.get_profile_QL(),
                     "QL_ice": m.get_profile_QL_ice(), "QR": m.get_profile_QR(), "T": m.get_profile_T()}
    prof = batch.stack_profiles(("U", "V", "presf", "THL", "QT", "QL", "QL_ice", "QR", "T"), src)
    d = batch.engine.diagnostics(batch.gcm, batch.zf, prof)                                  # K5: t, ql_water
    h = lambda t: t.cpu().numpy()                # noqa: E731
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
# diagnostics for non-SP output columns: splib/spcpl.py:251-267
# ---------------------------------------------------------------------------------------------
def output_column_conversion(profile):
    """Adds Tv, Zh, Zf, Psurf, THL, QT to ``profile`` (dict of [n x nG] arrays, names as in the
    reference's netCDF mapping: T, SH, QL, QI, Pf, Ph, Zgfull, Zghalf). Works on one column ([nG]) or
    many ([n x nG]); computed by the diagnostics kernel K5."""
    eng = get_engine()
    one = numpy.asarray(_num(profile["T"])).ndim == 1
    g = {}
    for src, dst in (("T", "T"), ("SH", "SH"), ("QL", "QL"), ("QI", "QI"), ("Pf", "Pfull"), ("Zgfull", "Zgfull"),
                     ("Zghalf", "Zghalf")):
        a = numpy.atleast_2d(_num(profile[src]))
        g[dst] = torch.from_numpy(numpy.ascontiguousarray(a)).to(eng.device, eng.dtype)
    d = {k: v.cpu().numpy() for k, v in eng.diagnostics(g).items()}
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
