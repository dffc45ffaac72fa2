"""Config 1 golden vectors (tests/golden/config1_L{19,91}.npz; oracle-generated, see
tests/golden/make_config1_golden.py): CPU test = the oracle and the generator still reproduce them;
GPU test = the HIP path reproduces them."""
import os

import numpy
import pytest

from oracle import spcpl_oracle as orc
from sp_coupler_amd import synthetic
from tests import oracle_c

HERE = os.path.dirname(os.path.abspath(__file__))


def load(nG):
    z = numpy.load(os.path.join(HERE, "golden", "config1_L%d.npz" % nG))       # allow_pickle defaults to False
    gcm = {k[7:]: z[k] for k in z.files if k.startswith("in_gcm_")}
    prof = {k[7:]: z[k] for k in z.files if k.startswith("in_les_")}
    return z, gcm, z["in_zf"], z["in_zh"], prof


@pytest.mark.parametrize("nG", [19, 91])
def test_oracles_reproduce_config1_golden(nG):
    z, gcm, zf, zh, prof = load(nG)
    g2, zf2, zh2, p2 = synthetic.make_batch(2, nG, 160, seed=synthetic.CONFIGS[1][3], couple_surface=True)
    assert all(numpy.array_equal(gcm[k], g2[k]) for k in gcm) and all(numpy.array_equal(prof[k], p2[k]) for k in prof)
    f = orc.forward_batched(gcm, prof, zf, zh, 1.0, 900.0, couple_surface=True)
    b = orc.backward_batched(gcm, f["Zf"], prof, zf, 1.0, 900.0)
    for k in f:
        assert numpy.array_equal(f[k], z["fwd_" + k]), k
    for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A", "start_index"):
        assert numpy.array_equal(b[k], z["bwd_" + k]), k
    fc = oracle_c.forward(gcm, zf, zh, prof, 1.0, 900.0)                        # the C restatement too
    assert numpy.array_equal(fc["idx"], z["fwd_idx"]) and numpy.array_equal(fc["f_u"], z["fwd_f_u"])
    bc = oracle_c.backward(gcm, fc["Zf"], zf, prof, 1.0, 900.0, conservative=True, zh=zh, Zh=fc["Zh"])
    assert numpy.array_equal(bc["f_T"], z["bwdc_f_T"])


@pytest.mark.gpu
@pytest.mark.parametrize("nG", [19, 91])
def test_hip_reproduces_config1_golden(nG):
    import torch
    from sp_coupler_amd.engine import Engine
    from tests.gpu_util import EPS, assert_bits, host, to_dev
    eng = Engine("cuda:0")
    z, gcm, zf, zh, prof = load(nG)
    g, p = to_dev(gcm, eng.device), to_dev(prof, eng.device)
    zf_d, zh_d = torch.from_numpy(zf).to(eng.device), torch.from_numpy(zh).to(eng.device)
    f = eng.forward(g, zf_d, p, 1.0, 900.0, zh=zh_d, want_profiles=True, couple_surface=True)
    b = eng.backward(g, zf_d, p, 1.0, 900.0, Zf=f["Zf"])
    bc = eng.backward(g, zf_d, p, 1.0, 900.0, Zf=f["Zf"], conservative=True, zh=zh_d, Zh=f["Zh"])
    torch.cuda.synchronize()
    for k in ("idx", "Zf", "Zh", "u", "v", "qt", "ql_ref", "f_u", "f_v", "f_qt", "f_ql", "f_ps", "rainrate", "wqt", "z0m"):
        assert_bits(k, host(f[k]), z["fwd_" + k])
    assert numpy.abs(host(f["thl"]) - z["fwd_thl"]).max() <= 8 * EPS * numpy.abs(z["fwd_thl"]).max()
    assert numpy.abs(host(f["f_thl"]) - z["fwd_f_thl"]).max() <= 1e-10 * numpy.abs(z["fwd_f_thl"]).max()
    for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A", "start_index"):
        assert_bits(k, host(b[k]), z["bwd_" + k])
    for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A"):
        assert_bits(k, host(bc[k]), z["bwdc_" + k])
