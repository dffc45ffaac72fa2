"""CPU tests: pin the oracle (oracle/) against every known answer the reference's own tests hold
(tests/golden/reference_known_answers.json) and cross-check its two restatements."""
import json
import os

import numpy
import pytest

from oracle import spcpl_oracle as orc
from sp_coupler_amd import synthetic
from tests import oracle_c

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "reference_known_answers.json")))
TOL = KA["tolerance"]


# ---- reference's sputils_test.py -----------------------------------------------------------------
def test_rms():
    a, b, c = KA["rms"]["numbers"]
    assert abs(orc.rms(numpy.array([a, b, c])) - numpy.sqrt((a * a + b * b + c * c) / 3)) < TOL
    assert abs(orc.rms(numpy.array([a, b, c])) - KA["rms"]["expected"]) < TOL


def test_rms_repeat():
    a, n = KA["rms_repeat"]["a"], KA["rms_repeat"]["n"]
    assert abs(orc.rms(numpy.array([a] * n)) - a) < TOL


def test_exner():
    a = KA["exner"]["a"]
    p = a * orc.pref0
    assert abs(numpy.log(orc.exner(p)) - numpy.log(a) * orc.rd / orc.cp) < TOL
    assert abs(orc.exner(p) - KA["exner"]["expected"]) < TOL


def test_exner_unity():
    assert abs(orc.exner(orc.pref0) - 1) < TOL


def test_iexner():
    p = KA["iexner_product"]["a"] * orc.pref0
    assert abs(orc.exner(p) * orc.iexner(p) - 1) < TOL


def test_constants_match_reference_values():
    assert (orc.pref0, orc.rd, orc.rv, orc.cp, orc.rlv, orc.grav) == (1e5, 287.04, 461.5, 1004., 2.53e6, 9.81)
    assert orc.rd / orc.cp == 0.2858964143426295


# ---- reference's spcpl_test.py (cloud-fraction level-index map) ------------------------------------
def test_cloud_fraction_index_map_known_answer():
    cf = KA["cloud_fraction"]
    zh, Zh, zf = numpy.array(cf["zh"], float), numpy.array(cf["gcm_Zh"]), numpy.array(cf["zf"], float)
    idx = orc.cloud_fraction_indices(zh, Zh)
    assert idx.tolist() == cf["indices"]
    # the dummy LES's get_cloudfraction clips to 0..k-1 and the caller reverses (spcpl.py:28)
    A = (0.5 + 0.2 * numpy.cos(6. * zf / 4000.))[numpy.clip(idx, 0, 19)][::-1]
    assert numpy.allclose(A, cf["A_after_clip_and_reverse"], atol=cf["A_tolerance"], rtol=0)
    k = 20
    assert abs(A[0] - (0.5 + 0.2 * numpy.cos(6. * (1. - k) / k))) < TOL      # spcpl_test.py:15
    assert abs(A[-1] - (0.5 + 0.2)) < TOL                                    # spcpl_test.py:16
    # the C restatement gives the same integers
    idx_c = oracle_c.cloud_indices(zh, Zh[None, :].copy())
    assert idx_c[0].tolist() == cf["indices"]


# ---- numpy.interp restatement is bit-exact -------------------------------------------------------
def test_interp_restated_bit_exact_vs_numpy():
    rng = numpy.random.default_rng(7)
    for trial in range(300):
        n = int(rng.integers(1, 40))
        xp = numpy.sort(rng.normal(size=n)) * 1000
        if trial % 5 == 0 and n > 3:
            xp[n // 2] = xp[n // 2 - 1]            # duplicate abscissa
        fp = rng.normal(size=n)
        if trial % 7 == 0:
            fp[rng.integers(0, n)] = numpy.inf     # exercises the NaN fallbacks
        if trial % 11 == 0:
            fp[rng.integers(0, n)] = numpy.nan
        x = numpy.concatenate([rng.normal(size=25) * 1500, xp[: min(n, 5)], [numpy.nan, -numpy.inf, numpy.inf]])
        with numpy.errstate(all="ignore"):
            want = numpy.interp(x, xp, fp)
        got = orc.interp_restated(x, xp, fp)
        assert numpy.array_equal(want, got, equal_nan=True), trial


# ---- the two restatements agree ------------------------------------------------------------------
@pytest.mark.parametrize("nG,nL,per_col", [(19, 160, False), (91, 160, False), (137, 512, False), (91, 160, True)])
def test_c_oracle_matches_numpy_oracle(nG, nL, per_col):
    gcm, zf, zh, prof = synthetic.make_batch(24, nG, nL, seed=99 + nG, per_column_grid=per_col)
    ref = orc.forward_batched(gcm, prof, zf, zh, 0.7, 900.0, couple_surface=True)
    got = oracle_c.forward(gcm, zf, zh, prof, 0.7, 900.0)
    assert numpy.array_equal(ref["idx"], got["idx"])
    for k in ("Zf", "Zh", "ps", "f_ps", "rainrate", "z0m", "z0h"):          # no pow involved: bit-exact
        assert numpy.array_equal(ref[k], got[k]), k
    for k in ("u", "v", "qt", "f_u", "f_v", "f_qt", "ql_ref", "f_ql", "wqt"):   # no pow(): bit-exact
        assert numpy.array_equal(ref[k], got[k]), k
    # thl passes through pow(): numpy's SIMD pow and glibc's may differ in the last ulp; the forcing
    # (thl - thl_d)/dt cancels ~6 digits, so its tolerance is set on the scale of thl, not of f_thl
    assert numpy.abs(ref["thl"] - got["thl"]).max() <= 4 * 2.3e-16 * numpy.abs(ref["thl"]).max()
    assert numpy.abs(ref["f_thl"] - got["f_thl"]).max() <= 4 * 2.3e-16 * numpy.abs(ref["thl"]).max() * 0.7 / 900.0
    assert numpy.abs(ref["wthl"] - got["wthl"]).max() <= 4 * 2.3e-16 * numpy.abs(ref["wthl"]).max()
    refb = orc.backward_batched(gcm, ref["Zf"], prof, zf, 1.3, 900.0)
    gotb = oracle_c.backward(gcm, ref["Zf"], zf, prof, 1.3, 900.0)
    assert numpy.array_equal(refb["start_index"], gotb["start_index"])
    for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A"):
        assert numpy.array_equal(refb[k], gotb[k]), k
        assert numpy.array_equal(numpy.signbit(refb[k]), numpy.signbit(gotb[k])), k  # -0.0 above the LES top


def test_backward_masks_above_les_top_with_signed_zero_and_nan():
    gcm, zf, zh, prof = synthetic.make_batch(3, 91, 160, seed=5)
    f = orc.forward_batched(gcm, prof, zf, zh)
    gcm["T"][1, 0] = numpy.nan                                               # NaN above the LES top stays NaN
    b = orc.backward_batched(gcm, f["Zf"], prof, zf)
    si = b["start_index"]
    assert (si > 0).all() and (si < 91).all()
    for c in range(3):
        masked = b["f_U"][c, :si[c]]
        assert (masked == 0).all() and numpy.signbit(masked).any() and not numpy.signbit(masked).all()
        assert (f["Zf"][c, :si[c]] > zf[-1]).all() and (f["Zf"][c, si[c]:] <= zf[-1]).all()
    assert numpy.isnan(b["f_T"][1, 0]) and b["f_T"][0, 0] == 0


def test_interp_c_conserves_weighted_integral():
    """sputils.interp_c (reference sputils.py:173-189): rho-weighted layer means; a constant profile
    is reproduced inside the LES domain and levels reaching above the LES top stay 0."""
    gcm, zf, zh, prof = synthetic.make_batch(2, 19, 160, seed=11)
    f = orc.forward_batched(gcm, prof, zf, zh)
    Zh = f["Zh"][0]
    Q = orc.interp_c(Zh, zh, numpy.full(160, 3.25), prof["Rhobf"][0])
    inside = Zh[:-1] < zh[-1]
    assert numpy.allclose(Q[inside], 3.25, rtol=1e-12) and (Q[~inside] == 0).all() and inside.any() and (~inside).any()


def test_synthetic_batch_is_physical_and_deterministic():
    g1, zf, zh, p1 = synthetic.make_config(2, n_cols=8)
    g2, _, _, p2 = synthetic.make_config(2, n_cols=8)
    assert all(numpy.array_equal(g1[k], g2[k]) for k in g1) and all(numpy.array_equal(p1[k], p2[k]) for k in p1)
    assert g1["T"].shape == (8, 91) and g1["Phalf"].shape == (8, 92) and p1["U"].shape == (8, 160)
    assert (numpy.diff(g1["Phalf"], axis=1) > 0).all() and (g1["Pfull"] > 0).all()
    Zf = (g1["Zgfull"] - g1["Zghalf"][:, -1:]) / 9.81
    assert (numpy.diff(Zf, axis=1) < 0).all() and (Zf[:, -1] > 0).all() and (Zf[:, -1] < 100).all()
    assert zf[0] == 12.5 and zf[-1] == 3987.5 and zh[0] == 0.0 and zh[1] == 25.0   # dales-input/prof.inp.001
    A19, B19 = synthetic.hybrid_coefficients(19)
    assert A19[1] == 2000 and B19[-1] == 1 and len(A19) == 20


@pytest.mark.parametrize("nG,nL", [(19, 160), (91, 160), (137, 512)])
def test_conservative_c_oracle_bit_exact_vs_numpy_oracle(nG, nL):
    """sputils.interp_c / integral incl. numpy's pairwise `.sum()` order, restated in C"""
    gcm, zf, zh, prof = synthetic.make_batch(12, nG, nL, seed=300 + nG)
    f = orc.forward_batched(gcm, prof, zf, zh)
    ref = orc.backward_batched(gcm, f["Zf"], prof, zf, 0.9, 900.0, conservative=True, Zh=f["Zh"], zh=zh)
    got = oracle_c.backward(gcm, f["Zf"], zf, prof, 0.9, 900.0, conservative=True, zh=zh, Zh=f["Zh"])
    lin = orc.backward_batched(gcm, f["Zf"], prof, zf, 0.9, 900.0)
    for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A"):
        assert numpy.array_equal(ref[k], got[k]), k
    assert not numpy.array_equal(ref["f_T"], lin["f_T"])            # it really is a different scheme
    assert numpy.array_equal(ref["f_A"], lin["f_A"])                # A is not interpolated
