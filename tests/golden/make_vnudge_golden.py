#!/usr/bin/env python3
"""Generates tests/golden/vnudge_small.npz: inputs and expected outputs of the variability nudge
(splib/spcpl.py:613-744) for one small synthetic LES (12 x 10 x 24), with and without constantT.

PROVENANCE: the expected outputs come from THIS repo's oracle (oracle/vnudge_oracle.py: NumPy + scipy.optimize.brentq,
the routines the reference itself calls), not from the reference -- it cannot be imported here (omuse/amuse absent) and
its tests hold no fixture for this function.  They pin the oracle (and the installed numpy / scipy) against drift and
give the HIP path a fixed vector; they do NOT pin parity with the reference ("parity unpinned").  float64, stored exactly.
usage: python tests/golden/make_vnudge_golden.py"""
import os
import sys

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vnudge_oracle as vo  # noqa: E402
from tests.test_vnudge import make_les_fields  # noqa: E402

f = make_les_fields(12, 10, 24, seed=11)
numpy.random.seed(42)                       # splib.initialize seeds numpy's global generator with 42 (splib.py:181)
R = vo.make_R(12, 10)
out = {"in_" + k: v for k, v in f.items()}
out["in_R"] = R
for cT in (False, True):
    r = vo.variability_nudge(f["qt"], f["qsat"], f["ql_av"], f["qt_av"], f["presf"], f["ql_ref"], R, 900.0, cT, thl=f["thl"], ql=f["ql"])
    assert r["error"] is None
    tag = "cT%d_" % int(cT)
    for k in ("qt", "thl", "beta", "alpha", "qt_std", "a", "status"):
        out[tag + k] = r[k]
path = os.path.join(HERE, "vnudge_small.npz")
numpy.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes")
