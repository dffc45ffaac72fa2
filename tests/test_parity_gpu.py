"""-m gpu parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bars (BASELINE.json north_star): level / column indices bit-exact; fp64 forcings and
tendencies within 1e-10 relative of the oracle.  In fact everything that does not pass through
pow() is required to be BIT-exact here (interpolation, forcings of u, v, qt, ql, all tendencies);
values downstream of iexner()'s pow() are held to a few ulp of their own scale.
"""
import numpy
import pytest
import torch

from oracle import spcpl_oracle as orc
from sp_coupler_amd import synthetic
from tests import oracle_c
from tests.gpu_util import EPS, assert_bits, assert_close_scaled, host, to_dev

pytestmark = pytest.mark.gpu

REL_TOL = 1e-10          # north_star: fp64 forcings within 1e-10 relative of the reference
FACTOR, DT = 0.85, 900.0


@pytest.fixture(scope="module")
def eng():
    from sp_coupler_amd.engine import Engine
    return Engine("cuda:0")


def run_gpu(eng, gcm, zf, zh, prof, cb=0, Zf_from_forward=True):
    g, p = to_dev(gcm, eng.device), to_dev(prof, eng.device)
    zf_d, zh_d = torch.from_numpy(zf).to(eng.device), torch.from_numpy(zh).to(eng.device)
    fwd = eng.forward(g, zf_d, p, FACTOR, DT, zh=zh_d, want_profiles=True, couple_surface=True, cols_per_block=cb)
    bwd = eng.backward(g, zf_d, p, FACTOR, DT, Zf=fwd["Zf"] if Zf_from_forward else None, cols_per_block=cb)
    torch.cuda.synchronize()
    return {k: host(v) for k, v in fwd.items()}, {k: host(v) for k, v in bwd.items()}


def check_forward(got, ref, thl_scale):
    assert_bits("idx", got["idx"], ref["idx"])
    for k in ("Zf", "Zh", "ps", "f_ps", "rainrate", "z0m", "z0h", "u", "v", "qt", "ql_ref", "f_u", "f_v", "f_qt",
              "f_ql", "wqt"):
        assert_bits(k, got[k], ref[k])
    ulp = 8 * EPS                                           # device pow vs libm pow: a few ulp
    assert_close_scaled("thl", got["thl"], ref["thl"], ulp, thl_scale)
    assert_close_scaled("f_thl", got["f_thl"], ref["f_thl"], ulp, thl_scale * abs(FACTOR) / DT)
    assert_close_scaled("wthl", got["wthl"], ref["wthl"], ulp, numpy.abs(ref["wthl"]).max())
    # the north_star bar, stated on the forcing's own scale
    assert_close_scaled("f_thl(1e-10)", got["f_thl"], ref["f_thl"], REL_TOL, numpy.abs(ref["f_thl"]).max())


def check_backward(got, ref):
    assert_bits("start_index", got["start_index"], ref["start_index"])
    for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A"):
        assert_bits(k, got[k], ref[k])


@pytest.mark.parametrize("n,nG,nL,cb", [(2, 19, 160, 0), (2, 91, 160, 0), (37, 91, 160, 0), (37, 91, 160, 4),
                                         (37, 91, 160, 8), (1024, 91, 160, 0), (300, 137, 512, 0), (5, 1, 3, 0),
                                         (9, 2, 1, 2)])
def test_exchange_matches_numpy_oracle(eng, n, nG, nL, cb):
    """configs 1 (both 19 and 91 levels), 2 and the level geometry of config 5, vs the NumPy oracle
    (per-column loop calling numpy.interp / numpy.searchsorted exactly like the reference)."""
    gcm, zf, zh, prof = synthetic.make_batch(n, nG, nL, seed=4000 + n + nG)
    fwd, bwd = run_gpu(eng, gcm, zf, zh, prof, cb)
    ref_f = orc.forward_batched(gcm, prof, zf, zh, FACTOR, DT, couple_surface=True)
    ref_b = orc.backward_batched(gcm, ref_f["Zf"], prof, zf, FACTOR, DT)
    check_forward(fwd, ref_f, numpy.abs(ref_f["thl"]).max())
    check_backward(bwd, ref_b)


def test_per_column_les_grid_and_recomputed_Zf(eng):
    gcm, zf, zh, prof = synthetic.make_batch(61, 91, 160, seed=77, per_column_grid=True)
    fwd, bwd = run_gpu(eng, gcm, zf, zh, prof, Zf_from_forward=False)
    ref_f = orc.forward_batched(gcm, prof, zf, zh, FACTOR, DT, couple_surface=True)
    ref_b = orc.backward_batched(gcm, ref_f["Zf"], prof, zf, FACTOR, DT)
    check_forward(fwd, ref_f, numpy.abs(ref_f["thl"]).max())
    check_backward(bwd, ref_b)


def test_t159_sized_batch_vs_c_oracle(eng):
    """config 3 (T159 full-SP, 35 718 columns) against the plain-C oracle, which finishes in seconds."""
    gcm, zf, zh, prof = synthetic.make_config(3)
    fwd, bwd = run_gpu(eng, gcm, zf, zh, prof)
    ref_f = oracle_c.forward(gcm, zf, zh, prof, FACTOR, DT)
    ref_b = oracle_c.backward(gcm, ref_f["Zf"], zf, prof, FACTOR, DT)
    check_forward(fwd, ref_f, numpy.abs(ref_f["thl"]).max())
    check_backward(bwd, ref_b)


def _ulp_search(target, denom):
    """a double z with z / denom == target exactly (so Zf hits an LES level bit-for-bit)"""
    z = target * denom
    for _ in range(64):
        q = z / denom
        if q == target:
            return z
        z = numpy.nextafter(z, numpy.inf if q < target else -numpy.inf)
    raise AssertionError("no exact preimage")


def make_edge_batch():
    """8 columns exercising the quirks of SURVEY.md section 7 (see test_edge_columns)."""
    gcm, zf, zh, prof = synthetic.make_batch(8, 91, 160, seed=31)
    grav = 9.81
    # col 0: no condensate at all
    gcm["QL"][0] = 0.0
    gcm["QI"][0] = 0.0
    # col 1: a monotone column in which many GCM full levels sit EXACTLY on LES full levels and several
    # half levels exactly on LES half levels (x == xp[j] branch of numpy.interp, side='right' ties)
    gcm["Zghalf"][1, -1] = 0.0
    low = []
    for lev in range(150, -1, -10):
        low += [zf[lev], zf[lev] - 3.3]
    Zf1 = numpy.concatenate([numpy.linspace(60000.0, 4100.0, 58), [zf[159]], low])
    assert Zf1.shape == (91,) and (numpy.diff(Zf1) < 0).all()
    exact_full = [k for k in range(91) if Zf1[k] in zf]
    gcm["Zgfull"][1] = [_ulp_search(t, grav) if k in exact_full else t * grav for k, t in enumerate(Zf1)]
    Zh1 = numpy.concatenate([[61000.0], 0.5 * (Zf1[:-1] + Zf1[1:]), [0.0]])
    exact_half = []
    for k in range(1, 91):
        m = numpy.ceil(Zf1[k] / 25.0) * 25.0
        if Zf1[k] < m < Zf1[k - 1] and m <= zh[-1] and k % 3 == 0:
            Zh1[k] = m
            exact_half.append(k)
    gcm["Zghalf"][1] = [_ulp_search(t, grav) if k in exact_half else t * grav for k, t in enumerate(Zh1)]
    assert len(exact_full) >= 17 and len(exact_half) >= 3
    # col 2: whole GCM column squeezed below the lowest LES level (every LES level clamps to the top value)
    gcm["Zgfull"][2] = gcm["Zghalf"][2, -1] + numpy.linspace(10.0, 0.5, 91) * grav
    gcm["Zghalf"][2, :-1] = gcm["Zghalf"][2, -1] + numpy.linspace(10.5, 0.7, 91) * grav
    # col 3: whole GCM column above the LES top (every LES level clamps to the lowest GCM value)
    gcm["Zgfull"][3] = gcm["Zghalf"][3, -1] + numpy.linspace(80000.0, 5000.0, 91) * grav
    gcm["Zghalf"][3, :-1] = gcm["Zghalf"][3, -1] + numpy.linspace(81000.0, 5200.0, 91) * grav
    # col 4: NaN temperature at the model top (must stay NaN through `*= 0`), negative-zero producing tendency
    gcm["T"][4, 0] = numpy.nan
    gcm["U"][4, 1] = -1e30
    # col 5: infinities in a source profile exercise numpy.interp's NaN fallbacks
    prof["U"][5, 10] = numpy.inf
    prof["T"][5, 20] = -numpy.inf
    gcm["U"][5, 80] = numpy.inf
    for d in (gcm, prof):
        for k in d:
            d[k] = numpy.ascontiguousarray(d[k])
    return gcm, zf, zh, prof, Zf1, Zh1, exact_full, exact_half


def test_edge_columns(eng):
    """Quirks of SURVEY.md section 7: end clamping both ways, x == xp[j] exact hits, QL=QI=0 columns,
    LES top above every GCM level / below the lowest one, NaN and +-0 handling."""
    gcm, zf, zh, prof, Zf1, Zh1, exact_full, exact_half = make_edge_batch()
    with numpy.errstate(all="ignore"):
        ref_f = orc.forward_batched(gcm, prof, zf, zh, FACTOR, DT, couple_surface=True)
        ref_b = orc.backward_batched(gcm, ref_f["Zf"], prof, zf, FACTOR, DT)
    # the fixture really exercises what it claims
    assert all(ref_f["Zf"][1, k] == Zf1[k] for k in exact_full) and all(ref_f["Zh"][1, k] == Zh1[k] for k in exact_half)
    assert (numpy.diff(ref_f["Zf"], axis=1) < 0).all()       # numpy.interp needs monotone abscissae
    assert ref_b["start_index"][2] == 0 and ref_b["start_index"][3] == 91
    assert numpy.isnan(ref_b["f_T"][4, 0]) and numpy.signbit(ref_b["f_U"][4]).any()
    fwd, bwd = run_gpu(eng, gcm, zf, zh, prof)
    ok = numpy.isfinite(ref_f["thl"])
    assert_bits("idx", fwd["idx"], ref_f["idx"])
    for k in ("Zf", "Zh", "u", "v", "qt", "ql_ref", "f_u", "f_v", "f_qt", "f_ql", "f_ps"):
        assert_bits(k, fwd[k], ref_f[k])
    assert numpy.array_equal(numpy.isfinite(fwd["thl"]), ok)
    assert numpy.abs(fwd["thl"][ok] - ref_f["thl"][ok]).max() <= 8 * EPS * numpy.abs(ref_f["thl"][ok]).max()
    check_backward(bwd, ref_b)


def test_standalone_cloud_indices(eng):
    gcm, zf, zh, prof = synthetic.make_batch(50, 91, 160, seed=12)
    ref = orc.forward_batched(gcm, prof, zf, zh)
    idx = eng.cloud_indices(torch.from_numpy(zh).to(eng.device), torch.from_numpy(ref["Zh"]).to(eng.device))
    torch.cuda.synchronize()
    assert_bits("idx", host(idx), ref["idx"])
    assert host(idx).min() >= 0 and host(idx).max() <= 160
    # the reference test's known answer (splib/test/spcpl_test.py:10-16), through the HIP kernel
    zh20 = torch.arange(20, dtype=torch.float64, device=eng.device) * 200 + 100
    Zh = torch.tensor([[1e5, 1e3, 100., 10., 1., 0.]], dtype=torch.float64, device=eng.device)
    assert host(eng.cloud_indices(zh20, Zh))[0].tolist() == [0, 0, 1, 5, 20]


def test_diagnostics(eng):
    gcm, zf, zh, prof = synthetic.make_batch(33, 91, 160, seed=3)
    g, p = to_dev(gcm, eng.device), to_dev(prof, eng.device)
    d = eng.diagnostics(g, torch.from_numpy(zf).to(eng.device), p)
    torch.cuda.synchronize()
    for i in range(33):
        col = {k: gcm[k][i] for k in orc.gcm_vars}
        c = orc.convert_profiles(col, zf)
        assert_bits("Tv", host(d["Tv"])[i], c["Tv"])
        assert_bits("QT", host(d["QT"])[i], c["QT"])
        assert_bits("Zf", host(d["Zf"])[i], c["Zf"])
        assert_bits("Zh", host(d["Zh"])[i], c["Zh"])
        assert numpy.abs(host(d["THL"])[i] - c["THL"]).max() <= 8 * EPS * numpy.abs(c["THL"]).max()
        pf = orc.interp(zf, c["Zf"][::-1], col["Pfull"][::-1])
        assert_bits("pf", host(d["pf"])[i], pf)
        t = prof["THL"][i] * orc.exner(pf) + orc.rlv * prof["QL"][i] / orc.cp
        assert numpy.abs(host(d["t"])[i] - t).max() <= 8 * EPS * numpy.abs(t).max()
        assert_bits("ql_water", host(d["ql_water"])[i], prof["QL"][i] - prof["QL_ice"][i])


def test_f32_variant_tolerance(eng):
    """fp32 arithmetic (config 5's sweep) against the fp64 oracle: report-level tolerances."""
    from sp_coupler_amd.engine import Engine
    e32 = Engine("cuda:0", dtype=torch.float32)
    gcm, zf, zh, prof = synthetic.make_batch(64, 137, 512, seed=8)
    g, p = to_dev(gcm, e32.device, torch.float32), to_dev(prof, e32.device, torch.float32)
    zf_d, zh_d = torch.from_numpy(zf).to(e32.device, torch.float32), torch.from_numpy(zh).to(e32.device, torch.float32)
    fwd = e32.forward(g, zf_d, p, 1.0, DT, zh=zh_d, want_profiles=True)
    torch.cuda.synchronize()
    ref = orc.forward_batched(gcm, prof, zf, zh, 1.0, DT)
    for k, tol in (("u", 2e-5), ("thl", 2e-6), ("qt", 2e-5)):
        rel = numpy.abs(host(fwd[k]).astype(numpy.float64) - ref[k]).max() / numpy.abs(ref[k]).max()
        assert rel < tol, (k, rel)
    # indices computed from fp32 heights may differ from fp64 only where a half level sits within fp32
    # rounding of an LES level; on this grid that is < 1 % of the entries
    assert (host(fwd["idx"]) != ref["idx"]).mean() < 0.01


def test_layout_and_argument_errors(eng):
    gcm, zf, zh, prof = synthetic.make_batch(4, 91, 160, seed=2)
    g, p = to_dev(gcm, eng.device), to_dev(prof, eng.device)
    zf_d = torch.from_numpy(zf).to(eng.device)
    bad = dict(g)
    bad["T"] = g["T"][:, :90]
    with pytest.raises(ValueError):
        eng.forward(bad, zf_d, p, 1.0, DT)
    bad = dict(g)
    bad["U"] = g["U"].to(torch.float32)
    with pytest.raises(ValueError):
        eng.forward(bad, zf_d, p, 1.0, DT)
    bad = dict(p)
    bad["QT"] = p["QT"].cpu()
    with pytest.raises(ValueError):
        eng.forward(g, zf_d, bad, 1.0, DT)
    # padded column pitch is part of the ABI: same numbers
    pad = {k: (torch.zeros(4, v.shape[1] + 5, device=eng.device, dtype=v.dtype)[:, :v.shape[1]].copy_(v)
               if v.dim() == 2 else v) for k, v in g.items()}
    a = eng.forward(g, zf_d, p, 1.0, DT)
    b = eng.forward(pad, zf_d, p, 1.0, DT)
    torch.cuda.synchronize()
    for k in a:
        assert_bits(k, host(b[k]), host(a[k]))
    # empty batch: no launch, empty outputs
    e = {k: v[:0] for k, v in g.items()}
    pe = {k: v[:0] for k, v in p.items()}
    out = eng.forward(e, zf_d, pe, 1.0, DT)
    assert out["f_u"].shape == (0, 160)


@pytest.mark.parametrize("n,nG,nL,per_col", [(40, 91, 160, False), (9, 19, 160, False), (64, 137, 512, False),
                                             (33, 91, 160, True), (2500, 91, 160, False)])
def test_conservative_coarsening_bit_exact(eng, n, nG, nL, per_col):
    """K4 (sputils.interp_c / integral, flag conservative_coarsening) against the NumPy oracle, which
    evaluates the reference's expressions with ndarray.sum(): bit-exact incl. the pairwise sum order."""
    gcm, zf, zh, prof = synthetic.make_batch(n, nG, nL, seed=900 + n, per_column_grid=per_col)
    g, p = to_dev(gcm, eng.device), to_dev(prof, eng.device)
    zf_d, zh_d = torch.from_numpy(zf).to(eng.device), torch.from_numpy(zh).to(eng.device)
    fwd = eng.forward(g, zf_d, p, FACTOR, DT, zh=zh_d)
    got = eng.backward(g, zf_d, p, FACTOR, DT, Zf=fwd["Zf"], conservative=True, zh=zh_d, Zh=fwd["Zh"])
    got2 = eng.backward(g, zf_d, p, FACTOR, DT, Zf=None, conservative=True, zh=zh_d)     # heights recomputed in-kernel
    torch.cuda.synchronize()
    ref_f = oracle_c.forward(gcm, zf, zh, prof, FACTOR, DT)
    if n <= 64:
        ref = orc.backward_batched(gcm, ref_f["Zf"], prof, zf, FACTOR, DT, conservative=True, Zh=ref_f["Zh"], zh=zh)
    else:
        ref = oracle_c.backward(gcm, ref_f["Zf"], zf, prof, FACTOR, DT, conservative=True, zh=zh, Zh=ref_f["Zh"])
    check_backward({k: host(v) for k, v in got.items()}, ref)
    check_backward({k: host(v) for k, v in got2.items()}, ref)
    assert (host(got["f_T"]) != 0).any()


@pytest.mark.parametrize("nL,scale,pd", [(2000, 0.1, -1), (240, 0.01, 1), (480, 0.004, 2), (960, 0.002, 3)])
def test_conservative_coarsening_thick_layers_recursive_pairwise_sums(eng, nL, scale, pd):
    """GCM layers that span hundreds of LES cells (1 m / 0.1 m LES spacing): ndarray.sum() then recurses (blocks of 128,
    halves split at multiples of 8); the kernel's sums must follow.  Round 1 refused nL > 513; any nL that fits LDS works
    now.  The run-time-geometry K4 unrolls numpy's recursion to the depth nL needs (pd = 1, 2, 3: up to 248 / 488 / 968
    LES levels; taller grids walk it with an explicit stack, pd = -1): every one of the four instantiations is run."""
    from sp_coupler_amd import _abi
    gcm, zf, zh, prof = synthetic.make_batch(12, 91, nL, seed=52)
    zf, zh = numpy.ascontiguousarray(zf * scale), numpy.ascontiguousarray(zh * scale)          # 1 m cells, top at 2 km
    d = _abi.Dims(12, 91, nL, 91, 92, nL, 1, 0)
    assert _abi.describe_launch(eng.lib, d, 4, 0).startswith("k_backward_cons2<f64,0,0,pd=%d>" % pd)
    g, p = to_dev(gcm, eng.device), to_dev(prof, eng.device)
    zf_d, zh_d = torch.from_numpy(zf).to(eng.device), torch.from_numpy(zh).to(eng.device)
    got = eng.backward(g, zf_d, p, FACTOR, DT, Zf=None, conservative=True, zh=zh_d)
    torch.cuda.synchronize()
    ref_f = oracle_c.forward(gcm, zf, zh, prof, FACTOR, DT)
    ref = oracle_c.backward(gcm, ref_f["Zf"], zf, prof, FACTOR, DT, conservative=True, zh=zh, Zh=ref_f["Zh"])
    cells = numpy.diff(numpy.searchsorted(zh, ref_f["Zh"][0][::-1]))
    assert cells.max() > 128                                 # the fixture really has layers of > 128 cells
    check_backward({k: host(v) for k, v in got.items()}, ref)
    # and against NumPy itself (the reference's own evaluation) on a few columns
    sub = {k: v[:3] for k, v in gcm.items()}
    psub = {k: v[:3] for k, v in prof.items()}
    ref_np = orc.backward_batched(sub, ref_f["Zf"][:3], psub, zf, FACTOR, DT, conservative=True, Zh=ref_f["Zh"][:3], zh=zh)
    for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V"):
        assert_bits(k, host(got[k])[:3], ref_np[k])


def test_tall_les_columns_use_large_lds_and_too_tall_is_refused(eng):
    """nL = 2048 needs > 64 KiB of LDS per workgroup in K3 (opt-in up to gfx950's 160 KiB); nL = 4000 cannot
    be staged and must be refused with SPC_ERR_UNSUPPORTED, not launched."""
    from sp_coupler_amd import _abi
    gcm, zf, zh, prof = synthetic.make_batch(6, 91, 2048, seed=44)
    fwd, bwd = run_gpu(eng, gcm, zf, zh, prof)
    ref_f = oracle_c.forward(gcm, zf, zh, prof, FACTOR, DT)
    ref_b = oracle_c.backward(gcm, ref_f["Zf"], zf, prof, FACTOR, DT)
    check_forward(fwd, ref_f, numpy.abs(ref_f["thl"]).max())
    check_backward(bwd, ref_b)
    gcm, zf, zh, prof = synthetic.make_batch(2, 19, 4000, seed=45)
    g, p = to_dev(gcm, eng.device), to_dev(prof, eng.device)
    with pytest.raises(_abi.SpcError) as e:
        eng.backward(g, torch.from_numpy(zf).to(eng.device), p, 1.0, DT)
    assert e.value.code == _abi.SPC_ERR_UNSUPPORTED


def test_random_geometries_pitches_and_slab_sizes(eng):
    """Fuzz of the run-time-geometry kernels: random level counts, column counts, slab sizes, padded pitches,
    shared / per-column LES grids -- every launch bit-checked against the plain-C oracle."""
    import os
    rng = numpy.random.default_rng(int(os.environ.get("SPC_FUZZ_SEED", "2026")))
    for trial in range(int(os.environ.get("SPC_FUZZ_TRIALS", "40"))):     # soak: SPC_FUZZ_TRIALS=500 SPC_FUZZ_SEED=n
        nG = int(rng.integers(1, 200))
        nL = int(rng.integers(1, 600))
        u = rng.uniform()     # mostly small; 257..1024: the 512 / 1024-thread path; above: the multi-round (PRE = false) kernels
        n = int(rng.integers(1, 260)) if u < 0.6 else (int(rng.integers(257, 1100)) if u < 0.85 else int(rng.integers(1100, 5000)))
        cb = int(rng.choice([0, 1, 2, 3, 5, 8]))
        per_col = bool(rng.integers(0, 2)) and nL > 1
        pad = int(rng.choice([0, 0, 1, 7]))
        gcm, zf, zh, prof = synthetic.make_batch(n, nG, nL, seed=7000 + trial, per_column_grid=per_col)

        def dev(v):
            t = torch.from_numpy(numpy.ascontiguousarray(v)).to(eng.device)
            if pad and t.dim() == 2:
                buf = torch.full((t.shape[0], t.shape[1] + pad), float("nan"), device=eng.device, dtype=t.dtype)
                buf[:, :t.shape[1]] = t
                return buf[:, :t.shape[1]]
            return t
        g = {k: dev(v) for k, v in gcm.items()}
        p = {k: dev(v) for k, v in prof.items()}
        zf_d, zh_d = dev(zf), dev(zh)
        fwd = eng.forward(g, zf_d, p, FACTOR, DT, zh=zh_d, want_profiles=True, couple_surface=True, cols_per_block=cb)
        bwd = eng.backward(g, zf_d, p, FACTOR, DT, Zf=fwd["Zf"] if trial % 2 else None, cols_per_block=cb)
        torch.cuda.synchronize()
        ref_f = oracle_c.forward(gcm, zf, zh, prof, FACTOR, DT)
        ref_b = oracle_c.backward(gcm, ref_f["Zf"], zf, prof, FACTOR, DT)
        tag = "trial %d nG=%d nL=%d n=%d cb=%d per_col=%s pad=%d" % (trial, nG, nL, n, cb, per_col, pad)
        try:
            check_forward({k: host(v) for k, v in fwd.items()}, ref_f, numpy.abs(ref_f["thl"]).max())
            check_backward({k: host(v) for k, v in bwd.items()}, ref_b)
            if nL >= 2:             # K4 on the same geometry (where integral() has no value both sides give NaN)
                bc = eng.backward(g, zf_d, p, FACTOR, DT, Zf=fwd["Zf"], conservative=True, zh=zh_d,
                                  Zh=fwd["Zh"] if trial % 2 else None, cols_per_block=cb)
                torch.cuda.synchronize()
                ref_c = oracle_c.backward(gcm, ref_f["Zf"], zf, prof, FACTOR, DT, conservative=True, zh=zh, Zh=ref_f["Zh"])
                check_backward({k: host(v) for k, v in bc.items()}, ref_c)
        except AssertionError as e:
            raise AssertionError(tag + ": " + str(e))


def _pad_pitch(t, pad):
    if pad == 0 or t.dim() != 2:
        return t
    buf = torch.full((t.shape[0], t.shape[1] + pad), float("nan"), device=t.device, dtype=t.dtype)
    buf[:, :t.shape[1]] = t
    return buf[:, :t.shape[1]]


@pytest.mark.parametrize("cfg,pad", [(2, 0), (3, 0), (2, 3), (3, 5)])
def test_the_plans_bench_py_times_are_bit_checked(eng, cfg, pad):
    """Round-1 verdict: the kernels the benchmark times had no parity test.  This builds EXACTLY what bench.py,
    tools/kbench.py and tools/pmc_run.py launch -- Engine.plan_exchange (lean K1: no optional outputs, no rain
    rate, fused index map; K3 with Zf recomputed and no start_index) through launch_raw -- on config 2 (small
    launch: write-through stores) and config 3 (large launch: plain stores), with contiguous columns (compile-time
    geometry kernels) and with a padded pitch (run-time-geometry kernels), and bit-compares every output with the
    plain-C oracle: f_u, f_v, f_qt, f_ql, ql_ref, f_ps, idx and the seven tendencies (f_thl <= 8 ulp of thl / dt)."""
    import ctypes
    gcm, zf, zh, prof = synthetic.make_config(cfg)
    g = {k: _pad_pitch(v, pad) for k, v in to_dev(gcm, eng.device).items()}
    p = {k: _pad_pitch(v, pad) for k, v in to_dev(prof, eng.device).items()}
    zf_d, zh_d = torch.from_numpy(zf).to(eng.device), torch.from_numpy(zh).to(eng.device)
    fp, bp = eng.plan_exchange(g, zf_d, zh_d, p, FACTOR, FACTOR, DT)
    assert set(fp.outputs) == {"f_u", "f_v", "f_thl", "f_qt", "f_ql", "ql_ref", "f_ps", "idx"}
    assert set(bp.outputs) == {"f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A"}
    for t in list(fp.outputs.values()) + list(bp.outputs.values()):
        t.fill_(float("nan")) if t.is_floating_point() else t.fill_(-7)
    sptr = ctypes.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream)
    fp.launch_raw(sptr)
    bp.launch_raw(sptr)
    torch.cuda.synchronize()
    ref_f = oracle_c.forward(gcm, zf, zh, prof, FACTOR, DT, couple_surface=False)
    ref_b = oracle_c.backward(gcm, None, zf, prof, FACTOR, DT)
    F = {k: host(v) for k, v in fp.outputs.items()}
    assert_bits("idx", F["idx"], ref_f["idx"])
    for k in ("f_u", "f_v", "f_qt", "f_ql", "ql_ref", "f_ps"):
        assert_bits(k, F[k], ref_f[k])
    thl_scale = numpy.abs(ref_f["thl"]).max()
    assert_close_scaled("f_thl", F["f_thl"], ref_f["f_thl"], 8 * EPS, thl_scale * abs(FACTOR) / DT)
    assert_close_scaled("f_thl(1e-10)", F["f_thl"], ref_f["f_thl"], REL_TOL, numpy.abs(ref_f["f_thl"]).max())
    for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A"):
        assert_bits(k, host(bp.outputs[k]), ref_b[k])


def test_exner_power_accuracy_and_special_values(eng):
    """spc_pow, the specialised x**(-rd/cp) of iexner (sputils.py:33-34), through K5's THL = T * iexner(Pf) with T = 1:
    pressures from 1e-3 Pa to 2e5 Pa (the atmosphere needs 1 Pa .. 1.1e5 Pa), tiny / huge normal numbers, and the
    arguments that must take ocml's pow(): 0, subnormal, negative, inf, NaN.  Finite results within 2 ulp of numpy's
    (libm, < 1 ulp) power; special values of the same class."""
    rng = numpy.random.default_rng(5)
    n, nG = 64, 91
    pf = numpy.exp(rng.uniform(numpy.log(1e-3), numpy.log(2e5), size=(n, nG)))
    pf[0, :8] = [1e-300, 1e300, 2.2250738585072014e-308, 1.7976931348623157e308, 1e5, 1e5 * (1 + 2 ** -52), 99999.99999999999, 3.0]
    pf[1, :6] = [0.0, 5e-324, -1.0, numpy.inf, numpy.nan, -0.0]
    gcm = {k: numpy.zeros((n, nG)) for k in ("SH", "QL", "QI", "Zgfull")}
    gcm.update(T=numpy.ones((n, nG)), Pfull=pf, Zghalf=numpy.zeros((n, nG + 1)))
    d = eng.diagnostics(to_dev(gcm, eng.device))
    torch.cuda.synchronize()
    got = host(d["THL"])
    with numpy.errstate(all="ignore"):
        want = (pf / 1e5) ** ((-287.04) / 1004.)
    fin = numpy.isfinite(want)
    assert numpy.array_equal(numpy.isfinite(got), fin) and numpy.array_equal(numpy.isnan(got), numpy.isnan(want))
    assert numpy.array_equal(got[numpy.isinf(want)], want[numpy.isinf(want)])
    ulp = numpy.abs(got[fin] - want[fin]) / numpy.spacing(numpy.abs(want[fin]))
    assert ulp.max() <= 2.0, ulp.max()
    assert (ulp <= 1.0).mean() > 0.999


@pytest.mark.parametrize("nG,nL,per_col,pad", [(91, 160, False, 0), (19, 160, False, 0), (60, 100, False, 0),
                                              (91, 160, True, 0), (91, 160, False, 3), (120, 136, False, 0)])
def test_small_batch_large_workgroups(eng, nG, nL, per_col, pad, monkeypatch):
    """257..1024 columns run 2 / 4 columns per workgroup of 512 / 1024 threads (spc_hip.hip:small_block): every size
    class, ragged last workgroups, compile-time and run-time geometries, per-column grids, padded pitches -- bit-checked
    against the plain-C oracle, and bit-equal to the 256-thread path (SPC_SMALL_BLOCK=0)."""
    import ctypes
    sptr = ctypes.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream)
    for n in (257, 300, 512, 513, 777, 1023, 1024):
        gcm, zf, zh, prof = synthetic.make_batch(n, nG, nL, seed=8800 + n, per_column_grid=per_col)
        g = {k: _pad_pitch(v, pad) for k, v in to_dev(gcm, eng.device).items()}
        p = {k: _pad_pitch(v, pad) for k, v in to_dev(prof, eng.device).items()}
        zf_d = _pad_pitch(torch.from_numpy(zf).to(eng.device), pad)
        zh_d = _pad_pitch(torch.from_numpy(zh).to(eng.device), pad)
        ref_f = oracle_c.forward(gcm, zf, zh, prof, FACTOR, DT, couple_surface=False)
        ref_b = oracle_c.backward(gcm, None, zf, prof, FACTOR, DT)
        outs = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("SPC_SMALL_BLOCK", mode)
            fp, bp = eng.plan_exchange(g, zf_d, zh_d, p, FACTOR, FACTOR, DT)
            for t in list(fp.outputs.values()) + list(bp.outputs.values()):
                t.fill_(float("nan")) if t.is_floating_point() else t.fill_(-7)
            fp.launch_raw(sptr)
            bp.launch_raw(sptr)
            bs = eng.backward(g, zf_d, p, FACTOR, DT)              # with start_index
            torch.cuda.synchronize()
            outs[mode] = {**{k: host(v) for k, v in fp.outputs.items()}, **{k: host(v) for k, v in bp.outputs.items()},
                          "start_index": host(bs["start_index"]), "f_T2": host(bs["f_T"])}
        o = outs["1"]
        tag = "n=%d " % n
        assert_bits(tag + "idx", o["idx"], ref_f["idx"])
        for k in ("f_u", "f_v", "f_qt", "f_ql", "ql_ref", "f_ps"):
            assert_bits(tag + k, o[k], ref_f[k])
        assert_close_scaled(tag + "f_thl", o["f_thl"], ref_f["f_thl"], 8 * EPS, numpy.abs(ref_f["thl"]).max() * abs(FACTOR) / DT)
        for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A"):
            assert_bits(tag + k, o[k], ref_b[k])
        assert_bits(tag + "start_index", o["start_index"], ref_b["start_index"])
        assert_bits(tag + "f_T (start_index launch)", o["f_T2"], ref_b["f_T"])
        for k in o:                                                  # both workgroup shapes: the same bits (f_thl too)
            assert_bits(tag + k + " (large vs 256-thread workgroups)", o[k], outs["0"][k])


def test_small_batch_large_workgroups_f32(eng):
    from sp_coupler_amd.engine import Engine
    e32 = Engine("cuda:0", dtype=torch.float32)
    for n in (400, 1000):
        gcm, zf, zh, prof = synthetic.make_batch(n, 91, 160, seed=8900 + n)
        g32 = {k: torch.from_numpy(v).to(e32.device, torch.float32) for k, v in gcm.items()}
        p32 = {k: torch.from_numpy(v).to(e32.device, torch.float32) for k, v in prof.items()}
        zf32, zh32 = torch.from_numpy(zf).to(e32.device, torch.float32), torch.from_numpy(zh).to(e32.device, torch.float32)
        fp, bp = e32.plan_exchange(g32, zf32, zh32, p32, FACTOR, FACTOR, DT)
        fp.launch()
        bp.launch()
        torch.cuda.synchronize()
        ref_f = orc.forward_batched(gcm, prof, zf, zh, FACTOR, DT, couple_surface=False)
        ref_b = orc.backward_batched(gcm, ref_f["Zf"], prof, zf, FACTOR, DT)
        for k, ref in (("f_u", ref_f), ("f_qt", ref_f), ("f_thl", ref_f), ("f_T", ref_b), ("f_U", ref_b)):
            got = host((fp.outputs if k in fp.outputs else bp.outputs)[k]).astype(numpy.float64)
            scale = numpy.abs(ref[k]).max()
            assert numpy.abs(got - ref[k]).max() <= 2e-3 * scale, k


@pytest.mark.parametrize("nG,nL", [(91, 160), (137, 512), (19, 160)])
def test_conservative_coarsening_thick_layers_compile_time_geometries(eng, nG, nL):
    """The compile-time-geometry K4 kernels sum a layer with numpy's pairwise recursion unrolled to a fixed depth
    (spc_k4.hpp: vn_pw): an LES grid so fine that the lowest GCM layer covers ~90 % of its cells (> 128: the recursion
    really splits) -- bit-exact against the plain-C oracle and, on a few columns, against NumPy itself."""
    gcm, zf, zh, prof = synthetic.make_batch(16, nG, nL, seed=61 + nG)
    ref0 = oracle_c.forward(gcm, zf, zh, prof, FACTOR, DT)
    dz = float(numpy.median(ref0["Zh"][:, nG - 1])) / (0.9 * nL)          # lowest layer ~ 0.9 nL cells thick
    zh = numpy.arange(nL, dtype=numpy.float64) * dz
    zf = zh + 0.5 * dz
    g, p = to_dev(gcm, eng.device), to_dev(prof, eng.device)
    zf_d, zh_d = torch.from_numpy(zf).to(eng.device), torch.from_numpy(zh).to(eng.device)
    got = eng.backward(g, zf_d, p, FACTOR, DT, Zf=None, conservative=True, zh=zh_d)
    torch.cuda.synchronize()
    ref_f = oracle_c.forward(gcm, zf, zh, prof, FACTOR, DT)
    ref = oracle_c.backward(gcm, ref_f["Zf"], zf, prof, FACTOR, DT, conservative=True, zh=zh, Zh=ref_f["Zh"])
    inside = ref_f["Zh"][:, nG - 1] <= zh[-1]
    cells = numpy.searchsorted(zh, ref_f["Zh"][inside, nG - 1])
    assert inside.sum() >= 3 and cells.max() > 128, (inside.sum(), cells.max() if inside.any() else None)
    check_backward({k: host(v) for k, v in got.items()}, ref)
    cols = numpy.nonzero(inside)[0][:3]
    sub = {k: v[cols] for k, v in gcm.items()}
    psub = {k: v[cols] for k, v in prof.items()}
    ref_np = orc.backward_batched(sub, ref_f["Zf"][cols], psub, zf, FACTOR, DT, conservative=True, Zh=ref_f["Zh"][cols], zh=zh)
    for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V"):
        assert_bits(k, host(got[k])[cols], ref_np[k])


def test_fp32_variant_is_the_float32_evaluation_of_the_reference_lines():
    """The fp32 arithmetic variant (BASELINE config 5's sweep) forms its quotients through fp64 -- (float)((double)a * r), r = 1 /
    (double)b (csrc/spc_hip.hip: Divisor<float>, round 5) -- which is the CORRECTLY ROUNDED float quotient: the kernels' u, v,
    qt, ql and their forcings equal the reference's lines (spcpl.py:197-198, 215, 224-228, 328-333) evaluated by NumPy in
    float32 arithmetic, bit for bit.  (thl passes through pow: csrc/spc_powf.h, tests/test_sputils_gpu.py.)"""
    from sp_coupler_amd.engine import Engine
    e32 = Engine("cuda:0", dtype=torch.float32)
    f4 = numpy.float32
    for n, nG, nL in ((300, 91, 160), (1500, 137, 512), (64, 60, 100)):
        gcm, zf, zh, prof = synthetic.make_batch(n, nG, nL, seed=31 + nG, couple_surface=False)
        g32 = {k: v.astype(f4) for k, v in gcm.items()}
        p32 = {k: v.astype(f4) for k, v in prof.items()}
        zf32, zh32 = zf.astype(f4), zh.astype(f4)
        dev = lambda d: {k: torch.from_numpy(v).cuda() for k, v in d.items() if k not in ("Rain", "rain_last")}      # noqa: E731
        factor, dt = 0.75, 900.0
        r = e32.forward(dev(g32), torch.from_numpy(zf32).cuda(), dev(p32), factor, dt, zh=torch.from_numpy(zh32).cuda(), want_profiles=True)
        torch.cuda.synchronize()
        Zf = (g32["Zgfull"] - g32["Zghalf"][:, -1:]) / f4(9.81)                           # spcpl.py:198 in float32
        Zh = (g32["Zghalf"] - g32["Zghalf"][:, -1:]) / f4(9.81)                           # spcpl.py:197
        assert Zf.dtype == f4
        assert_bits("Zf", host(r["Zf"]), Zf)
        assert_bits("Zh", host(r["Zh"]), Zh)

        def interp32(x, xp, fp):                     # numpy's arr_interp, every operation in float32
            j = numpy.searchsorted(xp, x, side="right") - 1
            jc = numpy.clip(j, 0, len(xp) - 2)
            slope = (fp[jc + 1] - fp[jc]) / (xp[jc + 1] - xp[jc])
            out = slope * (x - xp[jc]) + fp[jc]
            out = numpy.where(xp[jc] == x, fp[jc], out)
            out = numpy.where(j < 0, fp[0], out)
            return numpy.where(j >= len(xp) - 1, fp[-1], out).astype(f4)
        qt_ = g32["SH"] + g32["QL"] + g32["QI"]                                           # spcpl.py:215
        for name, src, les in (("u", g32["U"], "U"), ("v", g32["V"], "V"), ("qt", qt_, "QT"), ("ql_ref", g32["QL"], "QL")):
            want = numpy.stack([interp32(zf32, Zf[c, ::-1], src[c, ::-1]) for c in range(n)])       # spcpl.py:224-228
            assert want.dtype == f4
            assert_bits(name, host(r[name]), want)
            fname = {"u": "f_u", "v": "f_v", "qt": "f_qt", "ql_ref": "f_ql"}[name]
            assert_bits(fname, host(r[fname]), f4(factor) * (want - p32[les]) / f4(dt))                 # spcpl.py:328-333
        assert_bits("f_ps", host(r["f_ps"]), f4(factor) * (g32["Phalf"][:, -1] - p32["PS"]) / f4(dt))


def test_fp32_eight_byte_access_kernels_equal_the_scalar_float_kernels(monkeypatch):
    """csrc/spc_f32v.hpp (round 5): K1 of the float variant with float2 accesses -- two LES levels of a column, two adjacent
    elements of the flat [ncol x nG] slab (a pair may straddle two columns: nG is odd) -- against the scalar float kernel
    (SPC_F32_VEC=0): same bits.  Odd column counts (the scalar tail of the last slab), write-through and plain stores, a grid
    per column, the three compile-time geometries, and array bases that are only 4-byte aligned (the launcher must fall back)."""
    from sp_coupler_amd import _abi
    from sp_coupler_amd.engine import Engine
    e32 = Engine("cuda:0", dtype=torch.float32)
    f4 = torch.float32
    seen = set()
    for n, nG, nL, percol in ((5001, 91, 160, False), (6145, 91, 160, True), (40001, 91, 160, False), (5003, 137, 512, False),
                              (9001, 19, 160, False)):
        gcm, zf, zh, prof = synthetic.make_batch_tiled(n, nG, nL, seed=400 + n, base=2048, couple_surface=False) if not percol else \
            synthetic.make_batch(n, nG, nL, seed=400 + n, couple_surface=False, per_column_grid=True)
        g = {k: torch.from_numpy(v).to(e32.device, f4) for k, v in gcm.items()}
        p = {k: torch.from_numpy(v).to(e32.device, f4) for k, v in prof.items()}
        zf_d, zh_d = torch.from_numpy(zf).to(e32.device, f4), torch.from_numpy(zh).to(e32.device, f4)
        monkeypatch.setenv("SPC_F32_VEC", "1")
        fp, bp = e32.plan_exchange(g, zf_d, zh_d, p, 0.8, 0.8, 900.0)
        names = (fp.describe().split()[0], bp.describe().split()[0])         # (batches of one round keep one column per workgroup: scalar)
        assert names[0].startswith("k_forward_f32v<%d,%d" % (nG, nL)), names
        seen.update(nm.split("<")[0] + nm[nm.index(",wt"):] for nm in names if "f32v" in nm)
        fp.launch(), bp.launch()
        torch.cuda.synchronize()
        vec = {k: host(v).copy() for k, v in list(fp.outputs.items()) + list(bp.outputs.items())}
        for t in list(fp.outputs.values()) + list(bp.outputs.values()):
            t.fill_(7)
        monkeypatch.setenv("SPC_F32_VEC", "0")
        assert "f32v" not in fp.describe() and "f32v" not in bp.describe()
        fp.launch(), bp.launch()
        torch.cuda.synchronize()
        for k, v in list(fp.outputs.items()) + list(bp.outputs.items()):
            assert_bits("%d %d<->%d %s" % (n, nG, nL, k), vec[k], host(v))
        if percol:
            continue
        # bases that are 4-byte aligned only: rows [1:] of arrays with an odd row length
        monkeypatch.setenv("SPC_F32_VEC", "1")
        g1 = {k: torch.cat([v[:1], v])[1:] for k, v in g.items()}
        assert g1["T"].data_ptr() % 8 == (4 if nG % 2 else 0)
        fp1, bp1 = e32.plan_exchange(g1, zf_d, zh_d, p, 0.8, 0.8, 900.0)
        fp1.launch(), bp1.launch()
        torch.cuda.synchronize()
        for k, v in list(fp1.outputs.items()) + list(bp1.outputs.items()):
            assert_bits("misaligned %d %s" % (n, k), host(v), vec[k])
    assert {"k_forward_f32v,wt=0>", "k_forward_f32v,wt=1>"} <= seen, seen


def test_fp32_random_batch_sizes_all_float_paths_agree(monkeypatch):
    """Fuzz of the float kernels (soak: SPC_FUZZ_TRIALS / SPC_FUZZ_SEED): random column counts (odd and even, every size class),
    the three compile-time geometries, explicit slab sizes, shared / per-column grids -- the 8-byte-access K1, the scalar
    compile-time kernels and the run-time-geometry kernels (padded pitch) must give the same bits."""
    import os
    from sp_coupler_amd.engine import Engine
    e32 = Engine("cuda:0", dtype=torch.float32)
    rng = numpy.random.default_rng(int(os.environ.get("SPC_FUZZ_SEED", "2027")))
    trials = max(6, int(os.environ.get("SPC_FUZZ_TRIALS", "40")) // 4)
    for trial in range(trials):
        nG, nL = [(91, 160), (137, 512), (19, 160)][int(rng.integers(0, 3))]
        u = rng.uniform()
        n = int(rng.integers(1, 1100)) if u < 0.3 else (int(rng.integers(1100, 9000)) if u < 0.8 else int(rng.integers(9000, 30000)))
        if nL == 512:
            n = min(n, 8000)
        cb = int(rng.choice([0, 0, 0, 2, 3, 4, 8]))
        per_col = bool(rng.integers(0, 4) == 0)
        gcm, zf, zh, prof = synthetic.make_batch_tiled(n, nG, nL, seed=8100 + trial, base=1024, couple_surface=False) if not per_col else \
            synthetic.make_batch(min(n, 3000), nG, nL, seed=8100 + trial, couple_surface=False, per_column_grid=True)
        n = gcm["T"].shape[0]
        f4 = lambda d: {k: torch.from_numpy(v).to(e32.device, torch.float32) for k, v in d.items()}       # noqa: E731
        g, p = f4(gcm), f4(prof)
        zf_d, zh_d = (torch.from_numpy(z).to(e32.device, torch.float32) for z in (zf, zh))
        out = {}
        for mode in ("vec", "scalar", "runtime"):
            monkeypatch.setenv("SPC_F32_VEC", "1" if mode == "vec" else "0")
            if mode == "runtime":          # one element of padding: the run-time-geometry kernels
                pad = lambda t: torch.cat([t, torch.full_like(t[:, :1], float("nan"))], dim=1)[:, :t.shape[1]] if t.dim() == 2 else t   # noqa: E731
                gg, pp = {k: pad(v) for k, v in g.items()}, {k: pad(v) for k, v in p.items()}
                zz = (pad(zf_d), pad(zh_d)) if per_col else (zf_d, zh_d)
            else:
                gg, pp, zz = g, p, (zf_d, zh_d)
            fp, bp = e32.plan_exchange(gg, zz[0], zz[1], pp, 0.9, 0.9, 900.0, cols_per_block=cb)
            fp.launch(), bp.launch()
            torch.cuda.synchronize()
            out[mode] = {k: host(v).copy() for k, v in list(fp.outputs.items()) + list(bp.outputs.items())}
        tag = "trial %d %d<->%d n=%d cb=%d per_col=%s" % (trial, nG, nL, n, cb, per_col)
        for k in out["vec"]:
            assert_bits(tag + " vec/scalar " + k, out["vec"][k], out["scalar"][k])
            assert_bits(tag + " scalar/runtime " + k, out["scalar"][k], out["runtime"][k])
