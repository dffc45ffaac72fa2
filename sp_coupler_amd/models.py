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
            if hasattr(r, "wait"):
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
