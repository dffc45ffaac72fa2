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
mode = sys.argv[3] if len(sys.argv) > 3 else ""
with_k4 = mode == "k4"                                      # also the conservative backward (K4) on the same batches
percol = mode == "percol"                                   # LES grid packed per column [n x nL] (north_star's literal layout)
nG, nL = (137, 512) if mode.startswith("cfg5") else (91, 160)
dtype = torch.float32 if mode.endswith("f32") else torch.float64
eng = Engine("cuda:0", dtype=dtype)
sptr = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fpl, bpl, cpl = [], [], []
for r in range(rot):
    g, zf_d, zh_d, p, _ = synthetic.make_batch_tiled_device(eng.device, n, nG, nL, seed=500 + r, couple_surface=False, dtype=dtype,
                                                            per_column_grid=percol)
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
print("pmc workload done: n=%d rot=%d mode=%s %d<->%d %s copy_bytes=%d" % (n, rot, mode or "-", nG, nL, dtype, src.numel()))
