"""Where does a SHORT timed region (the driver's `--steps 20 --warmup 5`) lose time against the 1000-step rate?
Repeats the bench.py protocol (W warm-up steps, sync, K steps, sync) several times in one process and stamps
every step with HIP events, so the first launches after an idle gap can be told from steady state.
usage: python tools/shortrun.py [n_cols] [K] [W] [trials]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
from sp_coupler_amd.engine import Engine


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 35718
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    trials = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    eng = Engine("cuda:0")
    stream = torch.cuda.current_stream(eng.device)
    sptr = ctypes.c_void_p(stream.cuda_stream)
    wl = bench.Workload(eng, n, 91, 160, 7, 2, 1.0, 900.0)
    for idle in (0.0, 0.0, 0.05, 0.5, 0.0):
        for t in range(trials if idle == 0.0 else 2):
            if idle:
                torch.cuda.synchronize()
                time.sleep(idle)
            for i in range(W):
                wl.step(i, sptr)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
            t0 = time.perf_counter()
            ev[0].record(stream)
            for i in range(K):
                wl.step(i, sptr)
                ev[i + 1].record(stream)
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            per = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
            print("idle %.2fs trial %d: wall %.1f us/step (issue %.1f us/step), events total %.1f us/step; first 4 steps %s; "
                  "median %.1f" % (idle, t, wall / K * 1e6, t_issue / K * 1e6, sum(per) / K,
                                   " ".join("%.0f" % p for p in per[:4]), sorted(per)[K // 2]), flush=True)
    # the same without per-step events (exactly bench.py's region)
    for t in range(4):
        for i in range(W):
            wl.step(i, sptr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            wl.step(i, sptr)
        torch.cuda.synchronize()
        print("plain trial %d: wall %.1f us/step" % (t, (time.perf_counter() - t0) / K * 1e6), flush=True)


if __name__ == "__main__":
    main()
