#!/usr/bin/env python3
"""Where the drop-in step's coupler time goes (batched model protocol, 1024 columns): wall time of each spcpl call of
driver.Coupler.step minus the time spent inside model methods."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sp_coupler_amd import models, spcpl
from sp_coupler_amd.driver import Coupler

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
gcm, ens = models.make_batched_models(n, nG=91, nL=160, seed=3)
if len(sys.argv) > 2 and sys.argv[2] == "per-les":      # the reference's transport: a plain list of LES objects, plain GCM
    gcm.__class__ = models.SyntheticGCM
    ens = [ens[i] for i in range(n)]
cpl = Coupler(gcm, ens)
for _ in range(3):
    cpl.step()
acc = {}

def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        m0, t0 = models.model_seconds, time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) - (models.model_seconds - m0)
        return r
    return w
for name in ("gather_gcm_data", "set_les_forcings_batched", "get_les_profiles_batched", "get_les_profiles", "set_gcm_tendencies_batched"):
    setattr(spcpl, name, timed(name, getattr(spcpl, name)))
steps = 30
models.model_seconds = 0.0
t0 = time.perf_counter()
for _ in range(steps):
    cpl.step()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("n=%d: step %.3f ms wall, %.3f ms in model methods, %.3f ms coupler" % (n, wall / steps * 1e3, models.model_seconds / steps * 1e3,
                                                                       (wall - models.model_seconds) / steps * 1e3))
for k, v in acc.items():
    print("  %-28s %.3f ms" % (k, v / steps * 1e3))
