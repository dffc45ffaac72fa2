#!/usr/bin/env python3
"""Kernel micro-benchmark: average duration of K1 / K3 alone over rotating batches, for several
batch sizes and cols_per_block settings (HIP events around M back-to-back launches).
Usage: python tools/kbench.py [--sizes 1024,35718] [--cbs 0,1,2,4,8] [--levels 91,160]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sp_coupler_amd import synthetic  # noqa: E402
from sp_coupler_amd.engine import Engine  # noqa: E402


def bytes_model(nG, nL, esize=8, shared_grid=True):
    """bench.py's model: SURVEY 8(d), the shared LES grid subtracted when the batch shares it (honest per-column traffic)"""
    from bench import algorithmic_bytes
    ab = algorithmic_bytes(nG, nL, esize, shared_grid)
    return ab["k1_launch"], ab["k3_launch"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1024,35718")
    ap.add_argument("--cbs", default="0")
    ap.add_argument("--levels", default="91,160")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--heat-ms", type=float, default=40.0, help="run the kernel this long before timing it: the first ~20 ms "
                    "after an idle gap run 10-13 %% slower (clock ramp, profiles/r02_shortrun_clock_ramp.log)")
    ap.add_argument("--min-ms", type=float, default=10.0, help="timed region at least this long (iters is raised to fit)")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--tag", default="")
    ap.add_argument("--per-column-grid", action="store_true", help="LES grid packed per column [n x nL] (north_star's literal "
                    "layout, spcpl.py:222) instead of one shared [nL] vector; bytes counted accordingly (44 196 B per exchange)")
    a = ap.parse_args()
    nG, nL = (int(x) for x in a.levels.split(","))
    dtype = torch.float64 if a.dtype == "f64" else torch.float32
    eng = Engine("cuda:0", dtype=dtype)
    stream = torch.cuda.current_stream()
    sptr = ctypes.c_void_p(stream.cuda_stream)
    fb, bb = bytes_model(nG, nL, 8 if dtype == torch.float64 else 4, not a.per_column_grid)
    for n in (int(x) for x in a.sizes.split(",")):
        live = n * (fb + bb)
        rot = max(2, min(16, int(600e6 // live) + 1))
        data = []
        for r in range(rot):
            g, zf, zh, p, _ = synthetic.make_batch_tiled_device(eng.device, n, nG, nL, seed=100 + r, couple_surface=False, dtype=dtype,
                                                                per_column_grid=a.per_column_grid)
            data.append((g, zf, zh, p))
        for cb in (int(x) for x in a.cbs.split(",")):
            pl = [eng.plan_exchange(g, zf, zh, p, 1.0, 1.0, 900.0, cols_per_block=cb) for g, zf, zh, p in data]   # as bench.py
            fpl, bpl = [x[0] for x in pl], [x[1] for x in pl]
            res = {}
            for name, plans in (("K1", fpl), ("K3", bpl)):
                import time
                t0, done = time.perf_counter(), 0
                while True:                                   # pre-heat: steady clocks before the timed loop
                    for i in range(50):
                        plans[i % rot].launch_raw(sptr)
                    torch.cuda.synchronize()
                    done += 50
                    if (time.perf_counter() - t0) * 1e3 >= a.heat_ms:
                        break
                per = (time.perf_counter() - t0) / done
                iters = max(a.iters, int(a.min_ms * 1e-3 / per))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for i in range(iters):
                    plans[i % rot].launch_raw(sptr)
                e1.record(stream)
                torch.cuda.synchronize()
                res[name] = e0.elapsed_time(e1) * 1e3 / iters
            print("%s %-36s n=%d %d<->%d %s%s cb=%d rot=%d | K1 %.2f us %.0f GB/s (%.1f%% of 8TB/s) | K3 %.2f us %.0f GB/s (%.1f%%)" % (
                a.tag, fpl[0].describe().split()[0].replace("k_forward", ""), n, nG, nL, a.dtype, " grid-per-column" if a.per_column_grid else "", cb, rot, res["K1"], n * fb / res["K1"] / 1e3, n * fb / res["K1"] / 1e3 / 80,
                res["K3"], n * bb / res["K3"] / 1e3, n * bb / res["K3"] / 1e3 / 80), flush=True)


if __name__ == "__main__":
    main()
