"""K6 timing probe: 2 LES of 64x64x160, (a) the synthetic case, (b) nothing to do (ql_ref = ql_av = 0: no evaluation round),
(c) every level multiplicative.  Run under rocprofv3 --kernel-trace; kernels appear in this order, 5 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from sp_coupler_amd.engine import Engine
from tests.test_vnudge import make_les_fields
eng = Engine("cuda:0")
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 2
IT, JT, KT = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "64x64x160").split("x"))
f = make_les_fields(IT, JT, KT, seed=5)
rep = lambda x: torch.from_numpy(numpy.ascontiguousarray(numpy.broadcast_to(x, (ncol,) + x.shape))).cuda()
qt0, qsat = rep(f["qt"]), rep(f["qsat"])
R = torch.from_numpy(numpy.random.default_rng(1).normal(size=(ncol, IT, JT))).cuda()
for name, ql_ref, ql_av in (("synthetic", f["ql_ref"], f["ql_av"]), ("idle", f["ql_ref"] * 0, f["ql_av"] * 0)):
    prof = {"ql_av": rep(ql_av), "qt_av": rep(f["qt_av"]), "ql_ref": rep(ql_ref)}
    for i in range(5):
        qt = qt0.clone()
        r = eng.variability_nudge(qt, qsat, R, prof["ql_av"], prof["qt_av"], prof["ql_ref"])
        torch.cuda.synchronize()
    st = r["status"].cpu().numpy()
    print(name, "status counts:", {int(k): int((st == k).sum()) for k in numpy.unique(st)}, flush=True)
