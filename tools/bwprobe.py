#!/usr/bin/env python3
"""What this box's HBM delivers: pure read, pure write, copy, and the many-stream read:write mixes of the coupling
kernels (K1: 14 read + 7 write streams, K3: 16 + 7), for several grid sizes.  GB/s = all bytes moved / time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sp_coupler_amd.engine import Engine
eng = Engine("cuda:0")
s = torch.cuda.current_stream(); sp = ctypes.c_void_p(s.cuda_stream)
src = torch.empty(3 << 30, dtype=torch.uint8, device="cuda").random_(0, 255)
dst = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
MODES = os.environ.get("BWPROBE", "all")
cases = [(14, 7, 64 << 20), (114, 7, 64 << 20), (214, 7, 64 << 20)] if MODES == "burst" else None
for nr, nw, per in cases or ((1, 1, 1 << 30), (1, 0, 1 << 30), (8, 0, 128 << 20), (0, 1, 1 << 30), (0, 7, 128 << 20), (2, 1, 512 << 20),
                    (4, 2, 256 << 20), (14, 7, 64 << 20), (16, 7, 64 << 20)):
    for grid in ((256, 512, 1024, 2048, 4096) if cases else (1024, 2048, 4096, 16384)):
        def run(k):
            for _ in range(k):
                rc = eng.lib.spc_stream_probe(nr, nw, dst.data_ptr(), src.data_ptr(), per, grid, sp)
                assert rc == 0, eng.lib.spc_last_error()
        run(2); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); run(5); b.record(s); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 5
        print("read streams %2d write streams %d  %4d MiB each  grid %5d: %8.1f us  %6.0f GB/s" % (
            nr % 100, nw, per >> 20, grid, us, (nr % 100 + nw) * per / us / 1e3) + ("  [%d-thread workgroups]" % {0: 256, 1: 512, 2: 1024}[nr // 100]), flush=True)
