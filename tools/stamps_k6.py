#!/usr/bin/env python3
"""Diagnostic: where one evaluation round of K6's root finder goes (round-4 verdict, next 8).  Builds a -DSPC_STAMPS variant of
the library (never shipped): thread 0 of every k_vnudge_solve workgroup sums, over all its evaluation rounds, the shader-clock
time spent in (0) the hand-over of x / mode + barrier, (1) its wave's leaf sums, (2) the barrier behind them, (3) the tree
combine, (4) f and one step of the brentq state machine -- counters drained at every stamp, so the parts add up to a round that
is a little LONGER than an unstamped one.   usage: tools/stamps_k6.py [n_les] [itot]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

lib = "/tmp/libspc_stamps_k6.so"
if not os.path.exists(os.path.join(ROOT, "build", "variants", "libspc_stamps.so")):
    subprocess.run([ge.HIPCC] + ge.HIP_FLAGS + ["-DSPC_STAMPS=1", ge.HIP_SRC, "-o", lib], check=True)
else:
    lib = os.path.join(ROOT, "build", "variants", "libspc_stamps.so")
import numpy  # noqa: E402
import torch  # noqa: E402

from sp_coupler_amd.engine import Engine  # noqa: E402
from tests.test_vnudge import make_les_fields  # noqa: E402

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 2
it = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kt = 160
eng = Engine("cuda:0", lib_path=lib)
eng.lib.spc_debug_set_stamps.argtypes = [ctypes.c_void_p]
f = make_les_fields(it, it, kt, seed=5)
rep = lambda x: torch.from_numpy(numpy.ascontiguousarray(numpy.broadcast_to(x, (ncol,) + x.shape))).cuda()     # noqa: E731
qt0, qsat = rep(f["qt"]), rep(f["qsat"])
R = torch.from_numpy(numpy.random.default_rng(1).normal(size=(ncol, it, it))).cuda()
prof = {k: rep(f[k]) for k in ("ql_av", "qt_av", "ql_ref", "presf")}
qt = qt0.clone()


def run():
    qt.copy_(qt0)
    return eng.variability_nudge(qt, qsat, R, prof["ql_av"], prof["qt_av"], prof["ql_ref"])


for _ in range(20):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print("K6, %d LES of %dx%dx%d, stamped build with the stamps OFF: %.1f us per launch (incl. the qt reset copy)" % (ncol, it, it, kt, e0.elapsed_time(e1) * 1e3 / 20))
nblk = 1 << 16
stamps = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
assert eng.lib.spc_debug_set_stamps(stamps.data_ptr()) == 0
res = run()
torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(nblk, 8)
st_raw = st[st[:, 5] > 0]
st = st_raw.astype(numpy.float64)
rounds, wall_ns = st[:, 5], st[:, 6] * 10.0
raw7 = st_raw[:, 7]
m2_cyc, m2_rounds = (raw7 & ((1 << 48) - 1)).astype(numpy.float64), (raw7 >> 48).astype(numpy.float64)
cyc = st[:, :5]
tot_cyc = cyc.sum(axis=1) + m2_cyc
# calibrate the shader clock against the 100 MHz wall clock on the workgroups with the most rounds (their life is almost all rounds)
busy = rounds >= numpy.percentile(rounds, 90)
ns_per_cyc = numpy.median(wall_ns[busy] / tot_cyc[busy])
print("workgroups stamped: %d; evaluation rounds per workgroup: median %d, max %d; life of the busiest workgroups %.1f us; 1 counter tick ~ %.3f ns"
      % (len(st), numpy.median(rounds), rounds.max(), numpy.median(wall_ns[busy]) / 1e3, ns_per_cyc))
names = ["hand-over of x / mode + barrier", "leaf sums of thread 0's wave", "barrier behind the leaves", "tree combine (dependency rounds, one wave)",
         "f + one brentq step (one lane)"]
per_round = cyc[busy] / rounds[busy][:, None] * ns_per_cyc
for i, nm in enumerate(names):
    print("  %-46s %7.1f ns per round (p10 %.1f, p90 %.1f)" % (nm, numpy.median(per_round[:, i]), numpy.percentile(per_round[:, i], 10), numpy.percentile(per_round[:, i], 90)))
print("  %-46s %7.1f ns per round" % ("sum (without the mode-2 leaf sums below)", numpy.median(per_round.sum(axis=1))))
b1 = busy & (rounds - m2_rounds > 0)
b2 = busy & (m2_rounds > 0)
if b1.any():
    print("  leaf sums per MODE-1 round (multiplicative search: planes in LDS only)  %7.1f ns  (thread 0's level: %d such rounds in the median busy workgroup)"
          % (numpy.median(cyc[b1, 1] / (rounds[b1] - m2_rounds[b1]) * ns_per_cyc), numpy.median(rounds[b1] - m2_rounds[b1])))
if b2.any():
    print("  leaf sums per MODE-2 round (additive search: + a R)                   %7.1f ns  (%d such rounds)"
          % (numpy.median(m2_cyc[b2] / m2_rounds[b2] * ns_per_cyc), numpy.median(m2_rounds[b2])))
st_flags = res["status"].cpu().numpy()
print("levels with a multiplicative / additive root search: %d / %d of %d" % (int(((st_flags & 3) == 1).sum()), int(((st_flags & 3) == 2).sum()), st_flags.size))
