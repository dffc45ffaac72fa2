#!/usr/bin/env python3
"""Generates tests/golden/config1_L19.npz and config1_L91.npz: inputs and expected outputs of one
column-exchange for BASELINE.json's config 1 (T21 bundled case: 2 SP columns; the bundled vertical grid is
L19, BASELINE.json states 91 -- both are kept), 160 LES levels, dt = 900 s, factor = 1.

PROVENANCE: the expected outputs come from THIS repo's NumPy oracle (oracle/spcpl_oracle.py), not from the
reference itself -- the reference cannot be imported here (omuse/amuse absent).  They pin the oracle against
drift and give the HIP path a fixed vector; they do NOT pin parity with the reference ("parity unpinned"
beyond tests/golden/reference_known_answers.json).  float64, stored exactly.
usage: python tests/golden/make_config1_golden.py"""
import os
import sys

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import spcpl_oracle as orc  # noqa: E402
from sp_coupler_amd import synthetic  # noqa: E402

for nG in (19, 91):
    gcm, zf, zh, prof = synthetic.make_batch(2, nG, 160, seed=synthetic.CONFIGS[1][3], couple_surface=True)
    f = orc.forward_batched(gcm, prof, zf, zh, 1.0, 900.0, couple_surface=True)
    b = orc.backward_batched(gcm, f["Zf"], prof, zf, 1.0, 900.0)
    bc = orc.backward_batched(gcm, f["Zf"], prof, zf, 1.0, 900.0, conservative=True, Zh=f["Zh"], zh=zh)
    out = {"in_gcm_" + k: v for k, v in gcm.items()}
    out.update({"in_les_" + k: v for k, v in prof.items()})
    out.update(in_zf=zf, in_zh=zh)
    out.update({"fwd_" + k: v for k, v in f.items()})
    out.update({"bwd_" + k: v for k, v in b.items() if k.startswith("f_") or k == "start_index"})
    out.update({"bwdc_" + k: v for k, v in bc.items() if k.startswith("f_")})
    path = os.path.join(HERE, "config1_L%d.npz" % nG)
    numpy.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")
