#!/usr/bin/env python3
"""Workload for the PMC passes (run under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`):
a few K1/K3 launches on rotating batches far larger than the Infinity Cache, plus calibration copies of a
KNOWN byte count with 16 B/lane and 8 B/lane accesses (MI355X_MICROARCH.md, HBM: calibrate FETCH_SIZE in
your own access pattern)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sp_coupler_amd import synthetic  # noqa: E402
from sp_coupler_amd.engine import Engine  # noqa: E402
from tools import spc_tools  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rot = int(sys.argv[2]) if len(sys.argv) > 2 else 8
with_k4 = len(sys.argv) > 3 and sys.argv[3] == "k4"        # also the conservative backward (K4) on the same batches
eng = Engine("cuda:0")
sptr = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fpl, bpl, cpl = [], [], []
for r in range(rot):
    gcm, zf, zh, prof = synthetic.make_batch_tiled(n, 91, 160, seed=500 + r, couple_surface=False)
    g = {k: torch.from_numpy(v).cuda() for k, v in gcm.items()}
    p = {k: torch.from_numpy(v).cuda() for k, v in prof.items()}
    zf_d, zh_d = torch.from_numpy(zf).cuda(), torch.from_numpy(zh).cuda()
    fp, bp = eng.plan_exchange(g, zf_d, zh_d, p, 1.0, 1.0, 900.0)      # exactly what bench.py times
    fpl.append(fp)
    bpl.append(bp)
    if with_k4:
        cpl.append(eng.plan_backward(g, zf_d, p, 1.0, 900.0, Zf=None, want_start_index=False, conservative=True, zh=zh_d))
src = torch.empty(1 << 28, dtype=torch.uint8, device="cuda").random_(0, 255)      # 256 MiB
dst = torch.empty_like(src)
torch.cuda.synchronize()
for it in range(3):
    spc_tools.stream_copy(dst, src)
    spc_tools.stream_copy(dst, src, f64=True)
for i in range(3 * rot):
    fpl[i % rot].launch_raw(sptr)
    bpl[i % rot].launch_raw(sptr)
    if with_k4:
        cpl[i % rot].launch_raw(sptr)
torch.cuda.synchronize()
print("pmc workload done: n=%d rot=%d copy_bytes=%d" % (n, rot, src.numel()))
