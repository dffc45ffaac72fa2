#!/usr/bin/env python3
"""How long is ONE dependent fp64 add on this GPU?  k_vnudge_std (numpy's sequential qt.std sums: 2 x itot*jtot dependent adds
per level) cannot run faster than that chain.  One wave, n dependent adds, HIP events."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import spc_tools
tl = spc_tools.load()
s = torch.cuda.current_stream(); sp = ctypes.c_void_p(s.cuda_stream)
inp = torch.rand(64, dtype=torch.float64, device="cuda") * 1e-3
out = torch.empty(64, dtype=torch.float64, device="cuda")
for n in (8192, 65536, 1 << 20):
    for _ in range(3):
        assert tl.spc_probe_add_chain(out.data_ptr(), inp.data_ptr(), n, sp) == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(20):
        tl.spc_probe_add_chain(out.data_ptr(), inp.data_ptr(), n, sp)
    b.record(s); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 20
    print("%8d dependent v_add_f64: %9.2f us per launch -> %.2f ns per add" % (n, us, us * 1e3 / n), flush=True)
