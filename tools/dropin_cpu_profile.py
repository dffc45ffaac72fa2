#!/usr/bin/env python3
"""Host cost of the per-LES drop-in protocol WITHOUT a GPU: the product's spcpl + driver on a no-op engine (outputs are
zeros), so that only the Python / NumPy work of the coupler is timed.  `python tools/dropin_cpu_profile.py [n] [prof]`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from sp_coupler_amd import models, spcpl
from sp_coupler_amd.driver import Coupler


class _Plan:
    class _A:
        factor = dt = 0.0
    def __init__(self, outputs):
        self.outputs, self.args = outputs, _Plan._A()
    def set_scalars(self, f, d):
        pass
    def launch(self, stream=None):
        return self.outputs


class NullEngine:
    device, dtype = torch.device("cpu"), torch.float64
    def arena(self, specs, rows=None):
        from sp_coupler_amd.transfer import Arena
        return Arena(self.device, specs)
    def to_devices(self, host_array, rows=None, n_cols=None):
        return torch.from_numpy(numpy.ascontiguousarray(host_array))
    def plan_forward(self, g, zf, p, factor, dt, zh=None, out=None, **kw):
        return _Plan(dict(out))
    def plan_backward(self, g, zf, p, factor, dt, out=None, **kw):
        return _Plan(dict(out))
    def plan_diagnostics(self, g, zf=None, prof=None, out=None, **kw):
        return _Plan(dict(out or {}))
    def plan_cloud_indices(self, zh, Zh, out=None, **kw):
        return _Plan({"idx": out})


class NullLES:
    """model methods that cost (almost) nothing: what is left is the coupler's and the driver's own Python"""
    def __init__(self, gi, zf, zh, nL):
        self.grid_index, self.zf_cache, self.zh_cache = gi, zf, zh
        self._a, self._r, self._req = numpy.zeros(nL), models.ImmediateRequest(numpy.zeros(nL)), models.ImmediateRequest(None)
        self._s, self._sr = 1.0e5, models.ImmediateRequest(1.0e5)
        self._c = models.ImmediateRequest(numpy.zeros(91))
    def get_model_time(self): return 0.0
    def evolve_model(self, t, exactEnd=True): return self._req
    def get_cloudfraction(self, idx, return_request=False): return self._c if return_request else self._c.result()
for _m in ("get_profile_U", "get_profile_V", "get_profile_THL", "get_profile_QT", "get_profile_QL", "get_profile_QL_ice",
           "get_profile_QR", "get_profile_T", "get_presf", "get_rhof", "get_rhobf"):
    setattr(NullLES, _m, lambda self, return_request=False: self._r if return_request else self._a)
for _m in ("get_surface_pressure", "get_rain"):
    setattr(NullLES, _m, lambda self, return_request=False: self._sr if return_request else self._s)
for _m in ("set_tendency_U", "set_tendency_V", "set_tendency_THL", "set_tendency_QT", "set_tendency_QL",
           "set_tendency_surface_pressure", "set_ref_profile_QL"):
    setattr(NullLES, _m, lambda self, v, return_request=False: self._req if return_request else None)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spcpl.set_engine(NullEngine())
gcm, ens = models.make_batched_models(n, nG=91, nL=160, seed=3)
if len(sys.argv) > 2 and sys.argv[2] == "batched":
    cpl = Coupler(gcm, ens)
elif len(sys.argv) > 2 and sys.argv[2] == "null":
    gcm.__class__ = models.SyntheticGCM
    gcm.set_profile_tendency = lambda var, gi, v: None
    gcm.evolve_model_from_cloud_scheme = lambda: None
    cpl = Coupler(gcm, [NullLES(i + 1, ens.zf_cache, ens.zh_cache, 160) for i in range(n)])
else:
    gcm.__class__ = models.SyntheticGCM
    cpl = Coupler(gcm, [ens[i] for i in range(n)])
cpl.step(); cpl.step()
steps = 5
models.model_seconds = 0.0
if len(sys.argv) > 3:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(steps):
    cpl.step()
wall = time.perf_counter() - t0
if len(sys.argv) > 3:
    pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(25)
print("n=%d: %.2f ms/step wall, %.2f ms in model methods, %.2f ms coupler -> %.3g col-exch/s wall, %.3g minus models" % (
    n, wall / steps * 1e3, models.model_seconds / steps * 1e3, (wall - models.model_seconds) / steps * 1e3,
    n * steps / wall, n * steps / (wall - models.model_seconds)))
rows = cpl.timing_rows[-steps:]
names = ("gcm1", "gather", "forcings", "tendencies", "gcm2")
for j, nm in enumerate(names):
    print("  %-10s %.2f ms" % (nm, sum(r[1 + j] for r in rows) / steps * 1e3))
