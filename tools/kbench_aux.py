#!/usr/bin/env python3
"""Timing of the kernels off the default path: K4 (conservative coarsening, spc_backward_* with conservative=1),
K5 (spifs diagnostics) and K6 (variability nudge).  HIP events around back-to-back launches; bytes = algorithmic.
usage: python tools/kbench_aux.py [--sizes 1024,35718] [--levels 91,160] [--vn-cols 2,64]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy  # noqa: E402
import torch  # noqa: E402

from sp_coupler_amd import synthetic  # noqa: E402
from sp_coupler_amd.engine import Engine  # noqa: E402


def timed(fn, iters, heat_ms=40.0):
    import time
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < heat_ms:       # pre-heat past the clock ramp
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1024,35718")
    ap.add_argument("--levels", default="91,160")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--vn-cols", default="2,64")
    ap.add_argument("--sputils", action="store_true", help="also time the K7 operators (sputils helpers on their own) at every size")
    ap.add_argument("--k4-cbs", default="0", help="cols_per_block settings for K4 (0 = the library's choice)")
    ap.add_argument("--vn-shapes", default="64x64x160", help="LES field extents itot x jtot x ktot, comma separated")
    ap.add_argument("--vn-modes", default="default", help="default (LDS / streamed planes), sweep (SPC_VN_LDS=0: the sweeping kernel)")
    ap.add_argument("--vn-iters", type=int, default=20)
    a = ap.parse_args()
    nG, nL = (int(x) for x in a.levels.split(","))
    eng = Engine("cuda:0")
    sptr = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in (int(x) for x in a.sizes.split(",") if x):
        gcm, zf, zh, prof = synthetic.make_batch_tiled(n, nG, nL, seed=77, couple_surface=False)
        g = {k: torch.from_numpy(v).cuda() for k, v in gcm.items()}
        p = {k: torch.from_numpy(v).cuda() for k, v in prof.items()}
        zf_d, zh_d = torch.from_numpy(zf).cuda(), torch.from_numpy(zh).cuda()
        k3 = eng.plan_backward(g, zf_d, p, 1.0, 900.0, Zf=None, want_start_index=False)
        t3 = timed(lambda: k3.launch_raw(sptr), a.iters)
        for cb in (int(x) for x in a.k4_cbs.split(",")):
            k4 = eng.plan_backward(g, zf_d, p, 1.0, 900.0, Zf=None, want_start_index=False, conservative=True, zh=zh_d, cols_per_block=cb)
            t4 = timed(lambda: k4.launch_raw(sptr), a.iters)
            if cb:
                print("n=%d K4 cols_per_block=%d: %.1f us (%.2fx K3)" % (n, cb, t4, t4 / t3), flush=True)
        b3 = n * ((9 * nG + 6 * nL) + 7 * nG) * 8
        b4 = b3 + n * (nL + nG + 1) * 8                      # + Rhobf [nL], Zghalf [nG+1] per column
        k5a, k5b = eng.plan_diagnostics(g), eng.plan_diagnostics(g, zf_d, p)      # outputs bound once, as the step path's cached plans
        t5a = timed(lambda: k5a.launch_raw(sptr), a.iters)
        t5b = timed(lambda: k5b.launch_raw(sptr), a.iters)
        print("n=%d K5 launches: %s | %s" % (n, k5a.describe(), k5b.describe()), flush=True)
        b5a = n * ((6 * nG + nG + 1) + (4 * nG + nG + 1)) * 8          # reads T,SH,QL,QI,Pf,Zgfull,Zghalf; writes Tv,THL,QT,Zf,Zh
        b5b = b5a + n * (3 * nL + 3 * nL) * 8                          # + reads THL,QL,QL_ice; writes pf,t,ql_water
        print("n=%d %d<->%d | K3 %.1f us %.0f GB/s | K4 (conservative) %.1f us %.0f GB/s (%.2fx K3) | K5 gcm-level %.1f us %.0f GB/s | "
              "K5 +les-level %.1f us %.0f GB/s" % (n, nG, nL, t3, b3 / t3 / 1e3, t4, b4 / t4 / 1e3, t4 / t3, t5a, b5a / t5a / 1e3,
                                                 t5b, b5b / t5b / 1e3), flush=True)
    for n in (int(x) for x in a.sizes.split(",") if x and a.sputils):
        # K7: the helpers of splib/sputils.py as standalone operators, on the arrays K1 / K3 / K4 work on
        gcm, zf, zh, prof = synthetic.make_batch_tiled(n, nG, nL, seed=78, couple_surface=False)
        dev = lambda x: torch.from_numpy(numpy.ascontiguousarray(x)).cuda()      # noqa: E731
        Zf = dev(((gcm["Zgfull"] - gcm["Zghalf"][:, -1:]) / 9.81)[:, ::-1])       # ascending, as spcpl.py:224 passes it
        Zh = dev((gcm["Zghalf"] - gcm["Zghalf"][:, -1:]) / 9.81)
        T_, Pf = dev(gcm["T"][:, ::-1]), dev(gcm["Pfull"])
        zf_d, zh_d, qt, rho = dev(zf), dev(zh), dev(prof["QT"]), dev(prof["Rhobf"])
        rows = [("interp GCM->LES (x shared [nL], xp/fp per row [nG])", lambda: eng.interp(zf_d, Zf, T_), eng.plan_interp(zf_d, Zf, T_), n * (2 * nG + nL) * 8),
                ("interp LES->GCM (x per row [nG], xp shared [nL])", lambda: eng.interp(Zf, zf_d, qt), eng.plan_interp(Zf, zf_d, qt), n * (nG + nL + nG) * 8),
                ("searchsorted(zh, Zh, right)", lambda: eng.searchsorted(zh_d, Zh, side="right"), eng.plan_searchsorted(zh_d, Zh, side="right"), n * (2 * (nG + 1)) * 8),
                ("iexner(Pfull)", lambda: eng.exner(Pf, inverse=True), eng.plan_exner(Pf, inverse=True), n * 2 * nG * 8),
                ("interp_c(Zh, zh, qt, rhobf)", lambda: eng.interp_c(Zh, zh_d, qt, rho), eng.plan_interp_c(Zh, zh_d, qt, rho), n * (nG + 1 + 2 * nL + nG) * 8),
                ("rms rows [n x nL]", lambda: eng.rms(qt), eng.plan_rms(qt), n * (nL + 1) * 8)]
        import time
        for name, fn, plan, nbytes in rows:
            t = timed(fn, a.iters)                      # convenience call: checks + argument block + output allocation per call
            tp = timed(lambda: plan.launch_raw(sptr), a.iters)     # the plan: one foreign call, output bound once (out=)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                plan.run()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 200 * 1e6         # host + device per call, back to back
            print("n=%d K7 %-52s call %7.1f us | plan %7.1f us %6.0f GB/s (%.3f of 8 TB/s) | plan.run() wall %6.1f us" % (
                n, name, t, tp, nbytes / tp / 1e3, nbytes / tp / 1e3 / 8000.0, wall), flush=True)
    from tests.test_vnudge import make_les_fields
    for shape in (x for x in a.vn_shapes.split(",") if x):
        it, jt, kt = (int(v) for v in shape.split("x"))
        f = make_les_fields(it, jt, kt, seed=5)
        for ncol in (int(x) for x in a.vn_cols.split(",") if x):
            if ncol * it * jt * kt * 8 * 5 > 60e9:
                continue
            rep = lambda x: torch.from_numpy(numpy.ascontiguousarray(numpy.broadcast_to(x, (ncol,) + x.shape))).cuda()     # noqa: E731
            qt0, qsat = rep(f["qt"]), rep(f["qsat"])
            R = torch.from_numpy(numpy.random.default_rng(1).normal(size=(ncol, it, jt))).cuda()
            prof = {k: rep(f[k]) for k in ("ql_av", "qt_av", "ql_ref", "presf")}
            qt = qt0.clone()

            def run():
                qt.copy_(qt0)
                eng.variability_nudge(qt, qsat, R, prof["ql_av"], prof["qt_av"], prof["ql_ref"])
            for mode in a.vn_modes.split(","):
                os.environ["SPC_VN_LDS"] = "0" if mode == "sweep" else "1"
                t = timed(run, a.vn_iters)
                tc = timed(lambda: qt.copy_(qt0), a.vn_iters)
                print("K6 variability nudge [%s]: %d LES of %dx%dx%d: %.0f us per launch (%.0f us of it the qt reset copy)"
                      % (mode, ncol, it, jt, kt, t, tc), flush=True)
            os.environ.pop("SPC_VN_LDS", None)
            del qt0, qsat, qt, R, prof
            eng._vn_work = None
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
