"""`python -m sp_coupler_amd [--les N] [--steps K] [--levels nG,nL] [--cplsurf] [--conservative] [--out spifs.nc]`

Closed-loop demo on the in-process synthetic GCM/LES pair (sp_coupler_amd.models): K GCM steps of the
batched coupling (gather -> K1(+K2) -> LES step -> K3 -> GCM step) and one timing.txt-style row per step
(gcm1, gather, set_les_forcings, set_gcm_tendencies, gcm2 -- the columns of splib/splib.py:340-343)."""
import argparse
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m sp_coupler_amd")
    ap.add_argument("--les", type=int, default=64, help="number of superparameterized columns")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--levels", default="91,160")
    ap.add_argument("--cplsurf", action="store_true")
    ap.add_argument("--conservative", action="store_true")
    ap.add_argument("--out", default=None, help="write a spifs file (netCDF-3) here")
    a = ap.parse_args(argv)
    from . import models, spcpl, spio
    from .driver import Coupler
    nG, nL = (int(x) for x in a.levels.split(","))
    gcm, les_models = models.make_models(a.les, nG=nG, nL=nL, seed=1)
    if a.out:
        spcpl.writer = spio.SpifsWriter(a.out, [m.grid_index for m in les_models], [m.lat for m in les_models],
                                        [m.lon for m in les_models], les_models[0].zf_cache, nG)
    cpl = Coupler(gcm, les_models, cplsurf=a.cplsurf, conservative_coarsening=a.conservative, write=bool(a.out))
    t0 = time.time()
    cpl.run(a.steps)
    wall = time.time() - t0
    if a.out:
        spcpl.writer.close()
        spcpl.writer = None
    print("# %d SP columns, %d<->%d levels, %d steps in %.3f s" % (a.les, nG, nL, a.steps, wall))
    print("# gcm1 gather set_les_forcings set_gcm_tendencies gcm2   [s]")
    for row in cpl.timing_rows:
        print(" ".join("%9.5f" % x for x in row[1:6]))
    return cpl


if __name__ == "__main__":
    main()
