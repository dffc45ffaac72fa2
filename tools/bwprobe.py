#!/usr/bin/env python3
"""What this box's HBM delivers: pure read, pure write, copy, and the many-stream read:write mixes of the coupling
kernels (K1: 14 read + 7 write streams, K3: 16 + 7), for several grid sizes.  GB/s = all bytes moved / time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import spc_tools
tl = spc_tools.load()
s = torch.cuda.current_stream(); sp = ctypes.c_void_p(s.cuda_stream)
src = torch.empty(3 << 30, dtype=torch.uint8, device="cuda").random_(0, 255)
dst = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
MODES = os.environ.get("BWPROBE", "all")
HEAT_MS = float(os.environ.get("BWPROBE_HEAT_MS", "40"))


def heat(run, k):
    """run the case itself for HEAT_MS before timing it: the first ~20 ms after an idle gap are 10-13 % slower
    (clock ramp, profiles/r02_shortrun_clock_ramp.log) -- the round-2 logs taken without this read low"""
    import time
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < HEAT_MS:
        run(k)
        torch.cuda.synchronize()


cases = [(14, 7, 64 << 20), (114, 7, 64 << 20), (214, 7, 64 << 20)] if MODES == "burst" else None
if MODES == "stride":     # does the placement of the 21 streams relative to each other matter (channel / bank aliasing)?
    cases = [(14, 7, (64 << 20) + off) for off in (0, 256, 4096 + 256, (1 << 20) + 4096 + 256, 45719040 - (64 << 20), 26002704 - (64 << 20))]
if MODES == "small":
    # the config-2 launch shape: ONE launch moving 14 + 7 streams of 1 MiB (22 MB, what K1 moves at 1024 columns),
    # rotating through the 3 GiB / 2 GiB buffers so that every launch is cold
    per = 1 << 20
    for nr, nw in ((14, 7), (16, 7), (1, 1)):
        for grid in (256, 512, 1024, 2048):
            tot = (nr + nw) * per if (nr, nw) != (1, 1) else 2 * 11 * per
            pp = per if (nr, nw) != (1, 1) else 11 * per
            slots = min((3 << 30) // (max(nr, 1) * pp), (2 << 30) // (max(nw, 1) * pp))
            def run(k, nr=nr, nw=nw, pp=pp, grid=grid, slots=slots):
                for i in range(k):
                    o = i % slots
                    rc = tl.spc_stream_probe(nr, nw, dst.data_ptr() + o * nw * pp, src.data_ptr() + o * nr * pp, pp, grid, sp)
                    assert rc == 0, tl.spc_tools_last_error()
            heat(run, 50)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); run(2000); b.record(s); torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / 2000
            print("one launch, %2d read + %d write streams, %.1f MB in all, grid %4d: %6.2f us  %5.0f GB/s (%.1f%% of 8 TB/s)" % (
                nr, nw, tot / 1e6, grid, us, tot / us / 1e3, tot / us / 1e3 / 80), flush=True)
    sys.exit(0)
for nr, nw, per in cases or ((1, 1, 1 << 30), (1, 0, 1 << 30), (8, 0, 128 << 20), (0, 1, 1 << 30), (0, 7, 128 << 20), (2, 1, 512 << 20),
                    (4, 2, 256 << 20), (14, 7, 64 << 20), (16, 7, 64 << 20)):
    for grid in ((2048,) if MODES == "stride" else (256, 512, 1024, 2048, 4096) if cases else (1024, 2048, 4096, 16384)):
        def run(k):
            for _ in range(k):
                rc = tl.spc_stream_probe(nr, nw, dst.data_ptr(), src.data_ptr(), per, grid, sp)
                assert rc == 0, tl.spc_tools_last_error()
        heat(run, 5)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); run(40); b.record(s); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 40
        print("read streams %2d write streams %d  %10d B each  grid %5d: %8.1f us  %6.0f GB/s" % (
            nr % 100, nw, per, grid, us, (nr % 100 + nw) * per / us / 1e3) + ("  [%d-thread workgroups]" % {0: 256, 1: 512, 2: 1024}[nr // 100]), flush=True)
