#!/usr/bin/env python3
"""PCIe-inclusive rate: host NumPy arrays in -> upload -> K1(+K2) -> K3 -> download of every result the
model setters receive.  Reported in DESIGN.md next to (never instead of) bench.py's HBM-resident value."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy  # noqa: E402
import torch  # noqa: E402

from sp_coupler_amd import synthetic  # noqa: E402
from sp_coupler_amd.engine import Engine  # noqa: E402

eng = Engine("cuda:0")
FWD_IN = ("U", "V", "T", "SH", "QL", "QI", "Pfull", "Phalf", "Zgfull", "Zghalf", "A")
for n in (1024, 35718):
    gcm, zf, zh, prof = synthetic.make_batch(n, 91, 160, seed=3, couple_surface=False)
    gcm = {k: gcm[k] for k in FWD_IN}
    prof = {k: prof[k] for k in ("U", "V", "THL", "QT", "QL", "PS", "T", "QL_ice", "A")}
    pin = lambda d: {k: torch.from_numpy(v).pin_memory() for k, v in d.items()}   # noqa: E731
    for pinned in (False, True):
        gh, ph = (pin(gcm), pin(prof)) if pinned else ({k: torch.from_numpy(v) for k, v in gcm.items()},
                                                       {k: torch.from_numpy(v) for k, v in prof.items()})
        zf_d, zh_d = torch.from_numpy(zf).cuda(), torch.from_numpy(zh).cuda()

        def once():
            g = {k: v.to(eng.device, non_blocking=True) for k, v in gh.items()}
            p = {k: v.to(eng.device, non_blocking=True) for k, v in ph.items()}
            f = eng.forward(g, zf_d, p, 1.0, 900.0, zh=zh_d, want_heights=False)
            b = eng.backward(g, zf_d, p, 1.0, 900.0, want_start_index=False)
            out = [v.cpu() for v in list(f.values()) + list(b.values())]
            return out
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        reps = 20 if n <= 2048 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            once()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        nbytes = sum(v.numel() * 8 for v in list(gh.values()) + list(ph.values()))
        print("n=%d pinned=%s: %.3f ms per exchange incl. H2D+D2H -> %.3e column-exchanges/s (H2D %.1f MB)" % (
            n, pinned, dt * 1e3, n / dt, nbytes / 1e6), flush=True)
