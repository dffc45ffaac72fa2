#!/usr/bin/env python3
"""Does the ACCESS WIDTH cost bandwidth?  A 256 MiB device-to-device copy with 4, 8 and 16 bytes per lane (grid-stride, 2 048 and
8 192 workgroups of 256 threads): what the fp32 kernels (4 B per lane) can expect next to the fp64 ones (8 B)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import spc_tools
tl = spc_tools.load()
s = torch.cuda.current_stream(); sp = ctypes.c_void_p(s.cuda_stream)
n = 1 << 28
src = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
dst = torch.empty_like(src)
def timed(fn, it=20):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(it): fn()
    b.record(s); torch.cuda.synchronize()
    return 2.0 * n * it / (a.elapsed_time(b) * 1e-3) / 1e9
for rnd in range(2):
    for grid in (2048, 8192, 32768):
        print("4 B/lane, %5d workgroups: %6.0f GB/s" % (grid, timed(lambda: tl.spc_stream_copy_f32(dst.data_ptr(), src.data_ptr(), n, grid, sp))), flush=True)
    print("8 B/lane,  2048 workgroups: %6.0f GB/s" % timed(lambda: tl.spc_stream_copy_f64(dst.data_ptr(), src.data_ptr(), n, sp)), flush=True)
    print("16 B/lane, 2048 workgroups: %6.0f GB/s" % timed(lambda: tl.spc_stream_copy(dst.data_ptr(), src.data_ptr(), n, sp)), flush=True)
