"""In-process synthetic GCM / LES pair implementing the duck-typed model contract of the reference
(method names of ``omuse.community.{oifs,dales}.interface`` as called from
``splib/spcpl.py`` and ``splib/splib.py``).  Replaces the stale ``splib/spdummy.py`` stand-ins (which lack
``return_request=``, ``get_rain``, ``get_rhof``, ``Zgfull`` ...) for closed-loop tests and demos: the LES
relaxes toward the forcings it is given, the GCM applies the tendencies it is given.
Pure NumPy on the host: these are the *external models* either side of the hot path, not the path.
"""
import numpy

from . import synthetic


class ImmediateRequest:
    """Stand-in for an AMUSE async request: already complete."""

    def __init__(self, value=None):
        self._v = value

    def result(self):
        return self._v

    def wait(self):
        return None

    def is_result_available(self):
        return True


class RequestsPool:
    """Minimal ``AsyncRequestsPool`` (amuse.rfi.async_request; used at splib/splib.py:316-324)."""

    def __init__(self):
        self.requests = []

    def add_request(self, r):
        self.requests.append(r)

    def waitall(self):
        for r in self.requests:
            if type(r) is not ImmediateRequest and hasattr(r, "wait"):
                r.wait()
        self.requests = []


def _ret(value, return_request):
    return ImmediateRequest(value) if return_request else value


class SyntheticGCM:
    """OpenIFS stand-in: ``npoints`` grid columns of ``nG`` levels."""

    support_async = True

    def __init__(self, npoints, nG=91, seed=1, dt=900.0):
        self.state = synthetic.make_gcm_columns(npoints, nG, seed, couple_surface=True)
        self.npoints, self.ktot, self.dt = npoints, nG, dt
        rng = numpy.random.default_rng(seed + 5)
        self.longitudes = rng.uniform(0, 360, npoints)
        self.latitudes = rng.uniform(-90, 90, npoints)
        self.model_time = 0.0
        self.step = 0
        self.first_half_step_done = False
        self.mask = set()
        self.tendencies = {}
        self.calls = []

    # -- getters used by spcpl.gather_gcm_data (splib/spcpl.py:66,74)
    def get_profile_fields(self, var, cols):
        return self.state[var][numpy.asarray(cols, dtype=numpy.int64)]

    def get_surface_field(self, var, cols):
        return self.state[var][numpy.asarray(cols, dtype=numpy.int64)]

    # -- setter used by spcpl.set_gcm_tendencies (splib/spcpl.py:535-542)
    def set_profile_tendency(self, var, grid_index, values):
        v = numpy.array(getattr(values, "number", values), dtype=numpy.float64)
        assert v.shape == (self.ktot,), (var, v.shape)
        self.tendencies.setdefault(var, {})[int(grid_index)] = v

    def get_timestep(self):
        return self.dt

    def get_model_time(self):
        return self.model_time

    def set_mask(self, i):
        self.mask.add(int(i))

    def set_vdf_in_sp_mask(self, b):
        self.vdf_in_sp = bool(b)

    def evolve_model_until_cloud_scheme(self):
        self.calls.append("until_cloud_scheme")

    def evolve_model_cloud_scheme(self):
        self.calls.append("cloud_scheme")
        self.tendencies = {}            # "note: overwrites set tendencies" (splib/splib.py:299)

    def evolve_model_from_cloud_scheme(self):
        """apply the SP tendencies at the masked columns and advance time"""
        self.calls.append("from_cloud_scheme")
        for var, per_col in self.tendencies.items():
            for gi, f in per_col.items():
                f = numpy.where(numpy.isfinite(f), f, 0.0)
                self.state[var][gi] = self.state[var][gi] + self.dt * f
        for k in ("SH", "QL", "QI", "A"):
            numpy.clip(self.state[k], 0.0, None, out=self.state[k])
        self.model_time += self.dt


class SyntheticLES:
    """DALES stand-in: slab-mean profiles on ``nL`` levels that relax under the forcings."""

    support_async = True

    def __init__(self, gcm, grid_index, nL=160, seed=0):
        self.grid_index = int(grid_index)
        self.zf, self.zh = synthetic.les_grid(nL)
        one = {k: v[grid_index:grid_index + 1] for k, v in gcm.state.items()}
        _, _, prof = synthetic.make_les_profiles(one, nL, seed + 13 * grid_index)
        self.p = {k: (v[0].copy() if v.ndim == 2 else float(v[0])) for k, v in prof.items()}
        self.A_lev = numpy.clip(self.p["QL"] * 2e3, 0.0, 1.0)
        self.nL = nL
        self.model_time = 0.0
        self.tend = {}
        self.surf = {}
        self.received = []
        self.lat = self.lon = 0.0

    # -- grid
    def get_zf(self):
        return self.zf

    def get_zh(self):
        return self.zh

    def get_itot(self):
        return 8

    def get_jtot(self):
        return 8

    def get_ktot(self):
        return self.nL

    def get_model_time(self):
        return self.model_time

    # -- slab-mean getters (splib/spcpl.py:303-308, 748-766)
    def _g(self, key, return_request):
        v = self.p[key]
        return _ret(v.copy() if isinstance(v, numpy.ndarray) else v, return_request)

    def get_profile_U(self, return_request=False):
        return self._g("U", return_request)

    def get_profile_V(self, return_request=False):
        return self._g("V", return_request)

    def get_profile_THL(self, return_request=False):
        return self._g("THL", return_request)

    def get_profile_QT(self, return_request=False):
        return self._g("QT", return_request)

    def get_profile_QL(self, return_request=False):
        return self._g("QL", return_request)

    def get_profile_QL_ice(self, return_request=False):
        return self._g("QL_ice", return_request)

    def get_profile_QR(self, return_request=False):
        return self._g("QR", return_request)

    def get_profile_T(self, return_request=False):
        return self._g("T", return_request)

    def get_presf(self, return_request=False):
        return self._g("presf", return_request)

    def get_rhof(self, return_request=False):
        return self._g("Rhof", return_request)

    def get_rhobf(self, return_request=False):
        return self._g("Rhobf", return_request)

    def get_surface_pressure(self, return_request=False):
        return self._g("PS", return_request)

    def get_rain(self, return_request=False):
        return self._g("Rain", return_request)

    def get_cloudfraction(self, indices, return_request=False):
        idx = numpy.clip(numpy.asarray(indices), 0, self.nL - 1)       # like splib/spdummy.py:319-321
        return _ret(self.A_lev[idx], return_request)

    # -- setters (splib/spcpl.py:341-347, 361-364)
    def _s(self, name, values, return_request):
        v = numpy.array(getattr(values, "number", values), dtype=numpy.float64)
        self.tend[name] = v
        self.received.append(name)
        return _ret(None, return_request)

    def set_tendency_U(self, v, return_request=False):
        return self._s("U", v, return_request)

    def set_tendency_V(self, v, return_request=False):
        return self._s("V", v, return_request)

    def set_tendency_THL(self, v, return_request=False):
        return self._s("THL", v, return_request)

    def set_tendency_QT(self, v, return_request=False):
        return self._s("QT", v, return_request)

    def set_tendency_QL(self, v, return_request=False):
        return self._s("QL", v, return_request)

    def set_tendency_surface_pressure(self, v, return_request=False):
        return self._s("PS", v, return_request)

    def set_ref_profile_QL(self, v, return_request=False):
        return self._s("QL_ref", v, return_request)

    def set_z0m_surf(self, v, return_request=False):
        return self._s("z0m", v, return_request)

    def set_z0h_surf(self, v, return_request=False):
        return self._s("z0h", v, return_request)

    def set_wt_surf(self, v, return_request=False):
        return self._s("wt", v, return_request)

    def set_wq_surf(self, v, return_request=False):
        return self._s("wq", v, return_request)

    def set_field(self, name, values):
        self.p[name] = numpy.asarray(getattr(values, "number", values)).mean(axis=(0, 1))

    def set_surface_pressure(self, ps):
        self.p["PS"] = float(getattr(ps, "number", ps))

    # -- time stepping: the slab means follow the nudging forcings (that is what nudging does)
    def evolve_model(self, t, exactEnd=True):
        dt = float(t) - self.model_time
        if dt > 0:
            for key in ("U", "V", "THL", "QT", "QL"):
                if key in self.tend:
                    self.p[key] = self.p[key] + dt * self.tend[key]
            if "PS" in self.tend:
                self.p["PS"] = self.p["PS"] + dt * float(self.tend["PS"])
            self.p["QL"] = numpy.clip(self.p["QL"], 0.0, None)
            self.p["QL_ice"] = numpy.minimum(self.p["QL_ice"], self.p["QL"])
            self.p["T"] = self.p["THL"] * (self.p["presf"] / 1e5) ** (287.04 / 1004.) + 2.53e6 * self.p["QL"] / 1004.
            self.p["Rain"] = self.p["Rain"] + 1e-6 * dt
            self.A_lev = numpy.clip(self.p["QL"] * 2e3, 0.0, 1.0)
            self.model_time = float(t)
        return ImmediateRequest(0.0)

    def write_restart(self):
        pass


def make_models(n_les, npoints=None, nG=91, nL=160, seed=1):
    """A GCM with ``npoints`` columns and LES instances in the first ``n_les`` NON-ZERO grid indices
    (grid index 0 alone would trip the reference's ``any(cols)`` quirk, splib/spcpl.py:63)."""
    npoints = npoints or (n_les + 4)
    gcm = SyntheticGCM(npoints, nG, seed)
    les_models = []
    for i in range(1, n_les + 1):
        les = SyntheticLES(gcm, i, nL, seed)
        les.zf_cache, les.zh_cache = les.get_zf(), les.get_zh()          # splib/splib.py:152-153
        les.lat, les.lon = gcm.latitudes[i], gcm.longitudes[i]
        gcm.set_mask(i)
        les_models.append(les)
    return gcm, les_models


# ---------------------------------------------------------------------------------------------
# Batched model protocol (optional).  The reference talks to every LES through
# ~20 RPCs per column per step (splib/spcpl.py:341-347, 748-766) and to the GCM through 7 per column
# (spcpl.py:535-542).  A model object MAY additionally offer the batched calls below; sp_coupler_amd.spcpl uses
# them when present (one call per variable for ALL columns, writing straight into / reading straight from the
# coupler's pinned transfer buffers) and falls back to the reference's per-column calls otherwise.
#   GCM   supports_out = True           get_profile_fields(var, cols, out=ndarray), get_surface_field(..., out=)
#         set_profile_tendencies(var, grid_indices, values[n x nG])
#   LES   an ENSEMBLE object passed as `les_models` with batched = True, list-like over per-column LES objects:
#         grid_indices, zf_cache, zh_cache, get_profiles_batched(keys, out), get_cloudfraction_batched(indices, out),
#         set_forcings_batched(**arrays), evolve_model_batched(t); for qt_forcing='variance' additionally
#         get_fields_batched(name) -> [n x itot x jtot x ktot], set_fields_batched(name, array), model_time
# ---------------------------------------------------------------------------------------------
import time as _time

#: seconds spent INSIDE model methods of the batched stand-ins (bench.py subtracts it: `dropin` rate excludes model time)
model_seconds = 0.0


def _timed(fn):
    def wrapper(*a, **kw):
        global model_seconds
        t0 = _time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            model_seconds += _time.perf_counter() - t0
    wrapper.__name__ = fn.__name__
    wrapper.__doc__ = fn.__doc__
    return wrapper


class BatchedSyntheticGCM(SyntheticGCM):
    """SyntheticGCM + the optional batched protocol (out= getters, one tendency setter per variable)."""

    supports_out = True

    @_timed
    def get_profile_fields(self, var, cols, out=None):
        idx = numpy.asarray(cols, dtype=numpy.int64)
        if out is None:
            return self.state[var][idx]
        return numpy.take(self.state[var], idx, axis=0, out=out)

    get_surface_field = get_profile_fields

    @_timed
    def set_profile_tendencies(self, var, grid_indices, values):
        """all SP columns at once: values [n x nG] (a view into the coupler's transfer buffer: copied here)"""
        self.tendencies[var] = (numpy.asarray(grid_indices, dtype=numpy.int64), numpy.array(values, dtype=numpy.float64))

    @_timed
    def evolve_model_from_cloud_scheme(self):
        self.calls.append("from_cloud_scheme")
        for var, t in self.tendencies.items():
            if isinstance(t, tuple):
                gi, f = t
                f = numpy.where(numpy.isfinite(f), f, 0.0)
                self.state[var][gi] = self.state[var][gi] + self.dt * f
            else:
                for g, f in t.items():
                    f = numpy.where(numpy.isfinite(f), f, 0.0)
                    self.state[var][g] = self.state[var][g] + self.dt * f
        for k in ("SH", "QL", "QI", "A"):
            numpy.clip(self.state[k], 0.0, None, out=self.state[k])
        self.model_time += self.dt


class TimedSyntheticGCM(SyntheticGCM):
    """SyntheticGCM with the reference's per-column protocol only, its methods counted in ``model_seconds``"""

    get_profile_fields = _timed(SyntheticGCM.get_profile_fields)
    get_surface_field = _timed(SyntheticGCM.get_surface_field)
    set_profile_tendency = _timed(SyntheticGCM.set_profile_tendency)
    evolve_model_from_cloud_scheme = _timed(SyntheticGCM.evolve_model_from_cloud_scheme)


class _LESRow:
    """Per-column face of an ensemble row: the reference's per-LES method names on row i of the ensemble arrays."""

    support_async = True

    def __init__(self, ens, i):
        self._e, self._i = ens, i
        self.grid_index = int(ens.grid_indices[i])
        self.zf_cache, self.zh_cache = ens.zf_cache, ens.zh_cache
        self.lat = self.lon = 0.0
        self.received = []

    def get_zf(self):
        return self._e.zf_cache

    def get_zh(self):
        return self._e.zh_cache

    def get_itot(self):
        return 8

    get_jtot = get_itot

    def get_ktot(self):
        return self._e.nL

    def get_model_time(self):
        return self._e.model_time

    @property
    def model_time(self):
        return self._e.model_time

    @property
    def p(self):
        return {k: v[self._i] for k, v in self._e.p.items()}

    def get_cloudfraction(self, indices, return_request=False):
        v = self._e.A_lev[self._i].take(indices, mode="clip")          # like splib/spdummy.py:319-321
        return ImmediateRequest(v) if return_request else v

    def evolve_model(self, t, exactEnd=True):
        self._e.evolve_model_batched(t)         # the ensemble advances as one; later rows find it already there
        return ImmediateRequest(0.0)

    def write_restart(self):
        pass


def _row_getter(key):
    def get(self, return_request=False):
        v = self._e.p[key][self._i]
        v = v.copy() if v.ndim else float(v)
        return ImmediateRequest(v) if return_request else v
    return get


def _row_setter(name):
    def set_(self, values, return_request=False):
        e = self._e
        t = e.tend.get(name)
        if t is None:
            t = e.tend[name] = e._zeros_like_tend(name)
        t[self._i] = getattr(values, "number", values)
        self.received.append(name)
        return ImmediateRequest(None) if return_request else None
    return set_


for _m, _k in (("get_profile_U", "U"), ("get_profile_V", "V"), ("get_profile_THL", "THL"), ("get_profile_QT", "QT"),
               ("get_profile_QL", "QL"), ("get_profile_QL_ice", "QL_ice"), ("get_profile_QR", "QR"), ("get_profile_T", "T"),
               ("get_presf", "presf"), ("get_rhof", "Rhof"), ("get_rhobf", "Rhobf"), ("get_surface_pressure", "PS"),
               ("get_rain", "Rain")):
    setattr(_LESRow, _m, _row_getter(_k))
for _m, _k in (("set_tendency_U", "U"), ("set_tendency_V", "V"), ("set_tendency_THL", "THL"), ("set_tendency_QT", "QT"),
               ("set_tendency_QL", "QL"), ("set_tendency_surface_pressure", "PS"), ("set_ref_profile_QL", "QL_ref"),
               ("set_z0m_surf", "z0m"), ("set_z0h_surf", "z0h"), ("set_wt_surf", "wt"), ("set_wq_surf", "wq")):
    setattr(_LESRow, _m, _row_setter(_k))

# request-dict keys of set_les_forcings (splib/spcpl.py:383-385) -> tendency slots of the stand-in
_FORCING_SLOT = {"U": "U", "V": "V", "THL": "THL", "QT": "QT", "SP": "PS", "QL": "QL", "QLp": "QL_ref",
                 "Z0M_surf": "z0m", "Z0H_surf": "z0h", "WT_surf": "wt", "WQ_surf": "wq"}


class SyntheticLESEnsemble:
    """All DALES stand-ins of a run as [n x nL] arrays: same arithmetic as n ``SyntheticLES`` objects, with the
    optional batched protocol.  List-like over per-column faces (``_LESRow``), so the reference's per-LES loops
    work on it unchanged."""

    batched = True

    def __init__(self, grid_indices, zf, zh, prof):
        self.grid_indices = numpy.asarray(grid_indices, dtype=numpy.int64)
        self.n = len(self.grid_indices)
        self.zf_cache, self.zh_cache = zf, zh
        self.nL = zf.shape[-1]
        self.p = {k: numpy.array(v, dtype=numpy.float64) for k, v in prof.items() if k not in ("A", "rain_last")}
        self.A_lev = numpy.clip(self.p["QL"] * 2e3, 0.0, 1.0)
        self.model_time = 0.0
        self.tend = {}
        self._rows = [None] * self.n

    @classmethod
    def from_models(cls, les_list):
        """stack existing SyntheticLES objects (same state, same evolution)"""
        keys = les_list[0].p.keys()
        prof = {k: numpy.stack([numpy.asarray(m.p[k]) for m in les_list]) for k in keys}
        return cls([m.grid_index for m in les_list], les_list[0].zf, les_list[0].zh, prof)

    @classmethod
    def for_gcm(cls, gcm, grid_indices, nL=160, seed=0):
        """fresh ensemble for the given GCM columns (one vectorised generator call)"""
        gi = numpy.asarray(grid_indices, dtype=numpy.int64)
        sub = {k: v[gi] for k, v in gcm.state.items()}
        zf, zh, prof = synthetic.make_les_profiles(sub, nL, seed)
        for i in gi:
            gcm.set_mask(i)
        return cls(gi, zf, zh, prof)

    # -- list-like ---------------------------------------------------------------------------------
    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self.n))]
        if self._rows[i] is None:
            self._rows[i] = _LESRow(self, i)
        return self._rows[i]

    def __iter__(self):
        return (self[i] for i in range(self.n))

    def _zeros_like_tend(self, name):
        return numpy.zeros((self.n, self.nL)) if name in ("U", "V", "THL", "QT", "QL", "QL_ref") else numpy.zeros(self.n)

    # -- batched protocol --------------------------------------------------------------------------
    @_timed
    def get_profiles_batched(self, keys, out):
        """out[key][...] = slab means of ALL columns ([n x nL], or [n] for PS / Rain); `out` arrays are views
        into the coupler's pinned upload buffer"""
        for k in keys:
            numpy.copyto(out[k], self.p[k])

    @_timed
    def get_cloudfraction_batched(self, indices, out):
        """les.get_cloudfraction(indices) for every column: indices [n x nG] -> out [n x nG]"""
        idx = numpy.clip(indices, 0, self.nL - 1)
        numpy.copyto(out, numpy.take_along_axis(self.A_lev, idx, axis=1))

    @_timed
    def set_forcings_batched(self, **arrays):
        """the setters of spcpl.py:341-347 / 361-364 for all columns, keyed like the request dict of
        set_les_forcings (U, V, THL, QT, SP, QL, QLp, Z0M_surf, Z0H_surf, WT_surf, WQ_surf); values are views
        into the coupler's download buffer and are copied here"""
        for k, v in arrays.items():
            self.tend[_FORCING_SLOT[k]] = numpy.array(v, dtype=numpy.float64)

    # -- optional: 3-D fields for qt_forcing='variance' (spcpl.variability_nudge_ensemble) ------------------------------
    fields3d = None

    def attach_fields(self, fields):
        """``fields``: dict "Qsat", "QT", "THL", "QL" -> [n x itot x jtot x ktot] (what les.get_field returns, stacked)"""
        self.fields3d = {k: numpy.array(v, dtype=numpy.float64) for k, v in fields.items()}

    def get_fields_batched(self, name):
        if self.fields3d is None:
            raise NotImplementedError("this ensemble carries no 3-D fields (attach_fields)")
        return self.fields3d[name].copy()

    def set_fields_batched(self, name, values):
        self.fields3d[name] = numpy.array(getattr(values, "number", values), dtype=numpy.float64)

    @_timed
    def evolve_model_batched(self, t):
        dt = float(t) - self.model_time
        if dt <= 0:
            return
        p = self.p
        for key in ("U", "V", "THL", "QT", "QL"):
            if key in self.tend:
                p[key] = p[key] + dt * self.tend[key]
        if "PS" in self.tend:
            p["PS"] = p["PS"] + dt * self.tend["PS"]
        p["QL"] = numpy.clip(p["QL"], 0.0, None)
        p["QL_ice"] = numpy.minimum(p["QL_ice"], p["QL"])
        p["T"] = p["THL"] * (p["presf"] / 1e5) ** (287.04 / 1004.) + 2.53e6 * p["QL"] / 1004.
        p["Rain"] = p["Rain"] + 1e-6 * dt
        self.A_lev = numpy.clip(p["QL"] * 2e3, 0.0, 1.0)
        self.model_time = float(t)


class NullLES:
    """An LES face whose methods cost (almost) nothing -- getters hand out one preallocated array / request, setters
    drop their argument: what a step through it costs is the COUPLER's and the driver's own per-column work (bench.py
    `dropin.per_les_protocol.value_null_models`; subtracting measured model time instead leaks the timing wrappers'
    own overhead, ~35 calls per column and step, into the coupler's share)."""

    def __init__(self, grid_index, zf, zh, nG):
        nL = zf.shape[-1]
        self.grid_index, self.zf_cache, self.zh_cache = int(grid_index), zf, zh
        self.lat = self.lon = 0.0
        self._a, self._s = numpy.zeros(nL), 1.0e5
        self._ra, self._rs, self._rn = ImmediateRequest(self._a), ImmediateRequest(self._s), ImmediateRequest(None)
        self._c = numpy.zeros(nG)
        self._rc = ImmediateRequest(self._c)

    def get_zf(self):
        return self.zf_cache

    def get_zh(self):
        return self.zh_cache

    def get_model_time(self):
        return 0.0

    def evolve_model(self, t, exactEnd=True):
        return self._rn

    def get_cloudfraction(self, indices, return_request=False):
        return self._rc if return_request else self._c


for _m in ("get_profile_U", "get_profile_V", "get_profile_THL", "get_profile_QT", "get_profile_QL", "get_profile_QL_ice",
           "get_profile_QR", "get_profile_T", "get_presf", "get_rhof", "get_rhobf"):
    setattr(NullLES, _m, lambda self, return_request=False: self._ra if return_request else self._a)
for _m in ("get_surface_pressure", "get_rain"):
    setattr(NullLES, _m, lambda self, return_request=False: self._rs if return_request else self._s)
for _m in ("set_tendency_U", "set_tendency_V", "set_tendency_THL", "set_tendency_QT", "set_tendency_QL",
           "set_tendency_surface_pressure", "set_ref_profile_QL", "set_z0m_surf", "set_z0h_surf", "set_wt_surf", "set_wq_surf"):
    setattr(NullLES, _m, lambda self, v, return_request=False: self._rn if return_request else None)


class NullTendencyGCM(SyntheticGCM):
    """SyntheticGCM whose per-column tendency setter and second half step cost nothing (pairs with NullLES)"""

    def set_profile_tendency(self, var, grid_index, values):
        pass

    def evolve_model_from_cloud_scheme(self):
        self.model_time += self.dt


def make_batched_models(n_les, npoints=None, nG=91, nL=160, seed=1):
    """(BatchedSyntheticGCM, SyntheticLESEnsemble) with LES in grid columns 1..n_les -- the fast construction for
    large column counts (one vectorised generator call instead of n_les)."""
    npoints = npoints or (n_les + 4)
    gcm = BatchedSyntheticGCM(npoints, nG, seed)
    ens = SyntheticLESEnsemble.for_gcm(gcm, numpy.arange(1, n_les + 1), nL, seed)
    return gcm, ens
