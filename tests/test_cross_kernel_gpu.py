"""Closure between independently written kernels at BASELINE.json's full sizes (where no CPU oracle finishes in
seconds): what the fused step kernels K1 / K3 / K4 produce for EVERY column of config 4 (348 528 columns) must equal,
bit for bit, the same quantities composed from the standalone sputils operators (K7: interp, searchsorted, iexner,
interp_c -- different kernels, different search / interpolation code paths: `bracket` + `interp_at` and one thread per
layer instead of `bracket2` + `interp_fields` and eight lanes per layer) and IEEE-exact elementwise torch arithmetic in
the operation order of splib/spcpl.py:214-215, 224-228, 328-333, 471-489, 498, 518-533."""
import pytest
import torch

from sp_coupler_amd import synthetic

pytestmark = pytest.mark.gpu
DT, FACTOR = 900.0, 0.75
RLV, CP = 2.53e6, 1004.


@pytest.fixture(scope="module")
def case():
    from sp_coupler_amd.engine import Engine
    eng = Engine("cuda:0")
    n, nG, nL, seed = synthetic.CONFIGS[4]
    g, zf, zh, p, _ = synthetic.make_batch_tiled_device(eng.device, n, nG, nL, seed=seed, couple_surface=False)
    fwd = eng.forward(g, zf, p, FACTOR, DT, zh=zh, want_profiles=True)
    return eng, g, zf, zh, p, fwd


def div(t, scalar):
    """true IEEE division (torch divides a tensor by a Python scalar as a multiplication by the reciprocal)"""
    return t / torch.full((), scalar, dtype=t.dtype, device=t.device)


def tend(x, ref):
    return div(FACTOR * (x - ref), DT)


def test_forward_kernel_equals_the_composed_sputils_operators(case):
    eng, g, zf, zh, p, fwd = case
    Zf_up = fwd["Zf"].flip(1).contiguous()                                       # Zf[::-1], spcpl.py:224-228
    up = lambda t: t.flip(1).contiguous()                                        # noqa: E731
    thl_ = (g["T"] - div(RLV * (g["QL"] + g["QI"]), CP)) * eng.exner(g["Pfull"], inverse=True)      # spcpl.py:214
    qt_ = g["SH"] + g["QL"] + g["QI"]                                            # spcpl.py:215
    for name, src, slab in (("u", g["U"], "U"), ("v", g["V"], "V"), ("thl", thl_, "THL"), ("qt", qt_, "QT"), ("ql", g["QL"], "QL")):
        prof = eng.interp(zf, Zf_up, up(src))                                    # x = h shared, xp / fp per column
        assert torch.equal(prof, fwd["ql_ref" if name == "ql" else name]), name
        assert torch.equal(tend(prof, p[slab]), fwd["f_" + name]), "f_" + name   # spcpl.py:328-333
        del prof
    idx = eng.searchsorted(zh, fwd["Zh"], side="right")[:, :-1].flip(1)          # spcpl.py:26 / 764
    assert torch.equal(idx.to(torch.int32), fwd["idx"])


@pytest.mark.parametrize("conservative", [False, True])
def test_backward_kernels_equal_the_composed_sputils_operators(case, conservative):
    eng, g, zf, zh, p, fwd = case
    bwd = eng.backward(g, zf, p, FACTOR, DT, Zf=fwd["Zf"], conservative=conservative, zh=zh if conservative else None,
                       Zh=fwd["Zh"] if conservative else None)
    Zf = fwd["Zf"]
    nG = Zf.shape[1]
    start = eng.searchsorted(-Zf, -zf[-1:], side="left")                          # spcpl.py:498: searchsorted(-Zf, -h[-1])
    assert torch.equal(start[:, 0].to(torch.int32), bwd["start_index"])
    above = torch.arange(nG, device=Zf.device)[None, :] < start                  # f[0:start_index] *= 0, spcpl.py:527-533
    qlw = p["QL"] - p["QL_ice"]                                                  # spcpl.py:402
    if conservative:
        coarse = lambda q: eng.interp_c(fwd["Zh"], zh, q, p["Rhobf"])            # noqa: E731   spcpl.py:482-488
    else:
        coarse = lambda q: eng.interp(Zf, zf, q)                                 # noqa: E731   spcpl.py:471-477
    t, qt, ql = coarse(p["T"]), coarse(p["QT"]), coarse(p["QL"])
    want = {"f_T": tend(t, g["T"]), "f_SH": tend(qt - ql, g["SH"]), "f_QL": tend(coarse(qlw), g["QL"]),
            "f_QI": tend(coarse(p["QL_ice"]), g["QI"]), "f_U": tend(coarse(p["U"]), g["U"]), "f_V": tend(coarse(p["V"]), g["V"]),
            "f_A": tend(p["A"].flip(1), g["A"])}                                 # spcpl.py:404, 518-526
    for k, w in want.items():
        w = torch.where(above, w * 0.0, w)
        same = (w == bwd[k]) | (torch.isnan(w) & torch.isnan(bwd[k]))
        assert bool(same.all()), "%s: %d of %d elements differ" % (k, int((~same).sum()), same.numel())
        nz = ~torch.isnan(w)
        assert torch.equal(torch.signbit(w[nz]), torch.signbit(bwd[k][nz])), k + ": sign of zero"
