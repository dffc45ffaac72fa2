#!/usr/bin/env python3
"""Floor for ONE launch that moves as many bytes as K1 does at 1024 columns (14.3 MB read + 8.3 MB written):
a pure streaming copy of 11.3 MB (22.6 MB read+write) per launch, rotating through a 1 GiB buffer."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import spc_tools
tl = spc_tools.load()
nb = 11_300_000 // 16 * 16
big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda").random_(0, 255)
out = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream(); sp = ctypes.c_void_p(s.cuda_stream)
slots = (1 << 30) // nb
def run(iters):
    for i in range(iters):
        o = (i % slots) * nb
        tl.spc_stream_copy(out.data_ptr() + o, big.data_ptr() + o, nb, sp)
run(50); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(s); run(400); b.record(s); torch.cuda.synchronize()
us = a.elapsed_time(b) * 1e3 / 400
print("grid %s: %.2f us per 22.6 MB launch -> %.0f GB/s" % (os.environ.get("SPC_COPY_GRID", "2048"), us, 2 * nb / us / 1e3))
