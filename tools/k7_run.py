#!/usr/bin/env python3
"""Workload for rocprofv3 passes over the K7 operators (the helpers of splib/sputils.py on their own): every operator
ITERS times on the arrays K1 / K3 / K4 work on (n rows, 91 <-> 160 levels), plus calibration copies of a known byte count.
usage: [rocprofv3 ... --] python3 tools/k7_run.py [n_rows] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy  # noqa: E402
import torch  # noqa: E402

from sp_coupler_amd import synthetic  # noqa: E402
from sp_coupler_amd.engine import Engine  # noqa: E402
from tools import spc_tools  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 35718
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
nG, nL = 91, 160
eng = Engine("cuda:0")
gcm, zf, zh, prof = synthetic.make_batch_tiled(n, nG, nL, seed=78, couple_surface=False)
dev = lambda x: torch.from_numpy(numpy.ascontiguousarray(x)).cuda()      # noqa: E731
Zf = dev(((gcm["Zgfull"] - gcm["Zghalf"][:, -1:]) / 9.81)[:, ::-1])
Zh = dev((gcm["Zghalf"] - gcm["Zghalf"][:, -1:]) / 9.81)
T_, Pf = dev(gcm["T"][:, ::-1]), dev(gcm["Pfull"])
zf_d, zh_d, qt, rho = dev(zf), dev(zh), dev(prof["QT"]), dev(prof["Rhobf"])
src = torch.empty(1 << 28, dtype=torch.uint8, device="cuda").random_(0, 255)      # 256 MiB
dst = torch.empty_like(src)
big = torch.empty(1 << 27, dtype=torch.float64, device="cuda")                   # 1 GiB: flushes the Infinity Cache between operators
torch.cuda.synchronize()
for it in range(3):
    spc_tools.stream_copy(dst, src)
    spc_tools.stream_copy(dst, src, f64=True)
ops = [lambda: eng.interp(zf_d, Zf, T_), lambda: eng.interp(Zf, zf_d, qt), lambda: eng.searchsorted(zh_d, Zh, side="right"),
       lambda: eng.exner(Pf, inverse=True), lambda: eng.interp_c(Zh, zh_d, qt, rho), lambda: eng.rms(qt)]
for op in ops:
    for i in range(iters):
        if i % 4 == 0:
            big.fill_(1.0)
        op()
torch.cuda.synchronize()
print("k7 workload done: n=%d iters=%d copy_bytes=%d" % (n, iters, src.numel()))
