#!/usr/bin/env python3
"""Would a HIP graph pay for launch-bound sequences?  The six K7 operators (and K5 + K1 + K3) at 1 024 and 4 096 rows,
launched one by one through the plans (Python + ctypes + one dispatch each) against ONE replay of a torch.cuda.CUDAGraph
that captured the same launches on a side stream.  `python tools/graph_probe.py`."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from sp_coupler_amd import synthetic
from sp_coupler_amd.engine import Engine

eng = Engine("cuda:0")
torch.cuda.set_device(0)
nG, nL = 91, 160
for n in (1024, 4096):
    gcm, zf, zh, prof = synthetic.make_batch_tiled(n, nG, nL, seed=78, couple_surface=False)
    dev = lambda x: torch.from_numpy(numpy.ascontiguousarray(x)).cuda()      # noqa: E731
    Zf = dev(((gcm["Zgfull"] - gcm["Zghalf"][:, -1:]) / 9.81)[:, ::-1])
    Zh = dev((gcm["Zghalf"] - gcm["Zghalf"][:, -1:]) / 9.81)
    T_, Pf = dev(gcm["T"][:, ::-1]), dev(gcm["Pfull"])
    zf_d, zh_d, qt, rho = dev(zf), dev(zh), dev(prof["QT"]), dev(prof["Rhobf"])
    plans = [eng.plan_interp(zf_d, Zf, T_), eng.plan_interp(Zf, zf_d, qt), eng.plan_searchsorted(zh_d, Zh, side="right"),
             eng.plan_exner(Pf, inverse=True), eng.plan_interp_c(Zh, zh_d, qt, rho), eng.plan_rms(qt)]
    side = torch.cuda.Stream()
    sp = ctypes.c_void_p(side.cuda_stream)
    def seq():
        for p in plans:
            p.launch_raw(sp)
    for _ in range(20):
        seq()
    side.synchronize()
    ref = [p.result.clone() for p in plans]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        seq()
    for p in plans:
        p.result.zero_()
    g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(p.result, r) or bool(((p.result == r) | ((p.result != p.result) & (r != r))).all()) for p, r in zip(plans, ref))
    def wall(fn, it=300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(it):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / it * 1e6
    for _ in range(3):
        print("n=%d six K7 operators: one by one %.1f us | graph replay %.1f us | same bits: %s" % (n, wall(seq), wall(g.replay), same), flush=True)
