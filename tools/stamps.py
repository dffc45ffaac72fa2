#!/usr/bin/env python3
"""Diagnostic: where K1's / K3's time goes at small batches. Builds a -DSPC_STAMPS variant of the library
(never shipped), runs the kernel once warm, prints per-phase medians from in-kernel 100 MHz wall-clock stamps.
usage: tools/stamps.py [n_cols] [cols_per_block] [k1|k3|k4]      (STAMP_MODE=2: entry / end only, no intermediate drains)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

lib = "/tmp/libspc_stamps.so"
mode = os.environ.get("STAMP_MODE", "1")   # 1: every phase (drains memory at each stamp); 2: entry/end only
subprocess.run([ge.HIPCC] + ge.HIP_FLAGS + ["-DSPC_STAMPS=" + mode, ge.HIP_SRC, "-o", lib], check=True)
os.environ["SPC_LIB"] = lib
import numpy  # noqa: E402
import torch  # noqa: E402

from sp_coupler_amd import synthetic  # noqa: E402
from sp_coupler_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
which = sys.argv[3] if len(sys.argv) > 3 else "k1"
eng = Engine("cuda:0")
eng.lib.spc_debug_set_stamps.argtypes = [ctypes.c_void_p]
plans = []
ROT = 8 if n <= 4096 else 2
for r in range(ROT):
    gcm, zf, zh, prof = synthetic.make_batch_tiled(n, 91, 160, seed=r, couple_surface=False)
    prof = {k: v for k, v in prof.items() if k not in ("Rain", "rain_last")}      # lean hot path, as bench.py
    g = {k: torch.from_numpy(v).cuda() for k, v in gcm.items()}
    p = {k: torch.from_numpy(v).cuda() for k, v in prof.items()}
    if which in ("k3", "k4"):
        plans.append(eng.plan_backward(g, torch.from_numpy(zf).cuda(), p, 1.0, 900.0, Zf=None, want_start_index=False, cols_per_block=cb,
                                       conservative=(which == "k4"), zh=torch.from_numpy(zh).cuda()))
    else:
        plans.append(eng.plan_forward(g, torch.from_numpy(zf).cuda(), p, 1.0, 900.0, zh=torch.from_numpy(zh).cuda(),
                                      want_heights=False, cols_per_block=cb))
nblk = (n + max(cb, 1) - 1) // max(cb, 1) if cb else n
stamps = torch.zeros(n * 8, dtype=torch.int64, device="cuda")
import time  # noqa: E402
t_heat = time.perf_counter()
while time.perf_counter() - t_heat < 0.05:      # pre-heat past the clock ramp
    for i in range(16):
        plans[i % ROT].launch()
    torch.cuda.synchronize()
assert eng.lib.spc_debug_set_stamps(stamps.data_ptr()) == 0
plans[3 % ROT].launch()
torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(n, 8)
st = st[st[:, 0] > 0][:, :6].astype(numpy.float64) * 10.0   # ns
t0 = st[:, 0].min()
names = ["entry", "prologue loads landed", "phase1 done (pow, LDS)", "barrier passed",
         "phase2+idx done (stores landed)", "end"]
if which == "k3":
    names = ["entry", "prologue loads landed (LES slab, Zf, GCM inputs)", "staged in LDS", "barrier passed",
             "searches, 7 interpolations, tendencies; stores landed", "end"]
if which == "k4":
    names = ["entry", "LES slab, heights staged; barrier", "cell ranges (two scans per level); barrier", "layer sums (8 lanes per level); barrier",
             "tendencies; stores landed", "end"]
print(plans[0].describe())
print("blocks stamped:", len(st), " kernel span (first entry -> last end): %.2f us" % ((st[:, 5].max() - t0) / 1e3))
print("entry spread: %.2f us" % ((st[:, 0].max() - t0) / 1e3))
if mode == "2":
    life = st[:, 5] - st[:, 0]
    print("per-workgroup lifetime (entry -> end, no intermediate drains): median %.2f us p10 %.2f p90 %.2f max %.2f" % (
        numpy.median(life) / 1e3, numpy.percentile(life, 10) / 1e3, numpy.percentile(life, 90) / 1e3, life.max() / 1e3))
    print("end times rel. first entry: median %.2f p90 %.2f max %.2f us" % (
        numpy.median(st[:, 5] - t0) / 1e3, numpy.percentile(st[:, 5] - t0, 90) / 1e3, (st[:, 5] - t0).max() / 1e3))
    # timeline of a multi-round launch: resident workgroups and completions over time (start / steady state / drain)
    ent, end = (st[:, 0] - t0) / 1e3, (st[:, 5] - t0) / 1e3
    span = end.max()
    print("last workgroup enters at %.1f us of %.1f (drain = %.1f us)" % (ent.max(), span, span - ent.max()))
    nb = 20
    edges = numpy.linspace(0.0, span, nb + 1)
    print("  window [us]      resident(avg)  completed  lifetime of those completed (median us)")
    for a, b in zip(edges[:-1], edges[1:]):
        mid = numpy.linspace(a, b, 9)[1:-1]
        res = numpy.mean([((ent <= t) & (end > t)).sum() for t in mid])
        done = (end > a) & (end <= b)
        print("  %6.1f-%6.1f   %8.0f     %6d      %s" % (a, b, res, done.sum(), "%.1f" % numpy.median(life[done] / 1e3) if done.any() else "-"))
    sys.exit(0)
for i, nm in enumerate(names):
    d = st[:, i] - (st[:, i - 1] if i else t0)
    print("%-30s median +%.2f us  (p10 %.2f, p90 %.2f)   abs median %.2f us" % (
        nm, numpy.median(d) / 1e3, numpy.percentile(d, 10) / 1e3, numpy.percentile(d, 90) / 1e3,
        numpy.median(st[:, i] - t0) / 1e3))
