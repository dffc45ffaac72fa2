"""CPU: the checks behind bench.py's `config5` / `per_column_grid` legs and `--dtype f32` (round-4 verdict, items 1 and 3).
The fp32 arithmetic variant has no reference of its own (the reference computes in float64, AMUSE quantities over float64
arrays): its outputs are held against the fp64 oracle within stated bars (bench.F32_TOL), errors measured on the PROFILE's
scale because a forcing is a difference of nearly equal numbers (splib/spcpl.py:328-333, 518-526)."""
import numpy
import torch

import bench
from oracle import spcpl_oracle as orc
from sp_coupler_amd import synthetic
from tests.fake_engine import OracleEngine


def _outputs(n=64, nG=91, nL=160, per_column_grid=False, seed=9):
    gcm, zf, zh, prof = synthetic.make_batch(n, nG, nL, seed, couple_surface=False, per_column_grid=per_column_grid)
    f = orc.forward_batched(gcm, prof, zf, zh, 1.0, 900.0)
    b = orc.backward_batched(gcm, f["Zf"], prof, zf, 1.0, 900.0)
    return gcm, zf, zh, prof, f, b


def test_algorithmic_bytes_per_dtype_and_grid_layout():
    """44 196 B per column-exchange at 91 <-> 160 fp64 with the grid per column, 40 356 B with the
    shared grid subtracted; 109 900 B at 137 <-> 512; fp32 halves every array but the int32 index map"""
    assert bench.algorithmic_bytes(91, 160, 8, shared_grid=False)["exchange"] == 44196
    assert bench.algorithmic_bytes(91, 160, 8, shared_grid=True)["exchange"] == 40356
    assert bench.algorithmic_bytes(137, 512, 8, shared_grid=False)["exchange"] == 109900
    a8, a4 = bench.algorithmic_bytes(137, 512, 8), bench.algorithmic_bytes(137, 512, 4)
    assert a4["k3_launch"] * 2 == a8["k3_launch"] and (a4["k1_launch"] - 137 * 4) * 2 == a8["k1_launch"] - 137 * 4


def test_fp32_comparison_accepts_rounded_outputs_and_rejects_wrong_ones():
    gcm, zf, zh, prof, f, b = _outputs()
    F = {k: v.astype(numpy.float32) if v.dtype == numpy.float64 else v for k, v in f.items()}
    B = {k: v.astype(numpy.float32) if v.dtype == numpy.float64 else v for k, v in b.items()}
    scales = bench.f32_scales(gcm, prof, f)
    bad, det = bench.compare_with_oracle_f32(F, B, f, b, scales, 1.0, 900.0)
    assert not bad, bad
    assert det["idx_mismatch_fraction"] == 0.0 and det["f_thl"]["max_err_over_profile_scale"] < 1e-7
    # an error of 1e-3 of the profile's scale in one forcing is far outside every bar
    F2 = dict(F, f_u=F["f_u"] + numpy.float32(1e-3 * scales["f_u"] / 900.0))
    bad, _ = bench.compare_with_oracle_f32(F2, B, f, b, scales, 1.0, 900.0)
    assert any(x.startswith("f_u") for x in bad)
    # a level unmasked in one arithmetic only is counted, not measured -- and refused when it is more than a stray level
    B3 = dict(B, f_T=numpy.where(b["f_T"] == 0, numpy.float32(1.0), B["f_T"]))
    bad, det = bench.compare_with_oracle_f32(F, B3, f, b, scales, 1.0, 900.0)
    assert det["f_T"]["mask_mismatch_fraction"] > 0.1 and any("masked" in x for x in bad)
    # half of the index map one cell off
    F4 = dict(F, idx=F["idx"] + (numpy.arange(F["idx"].size).reshape(F["idx"].shape) % 2).astype(F["idx"].dtype))
    bad, _ = bench.compare_with_oracle_f32(F4, B, f, b, scales, 1.0, 900.0)
    assert any(x.startswith("idx") for x in bad)


def test_sample_check_handles_a_grid_per_column_and_float32_tensors():
    """what config_leg runs after its timed region, here on the test engine (NumPy oracle) instead of the device: rows read
    back, plain-C oracle, grid [nL] or [n x nL]; float32 tensors take the tolerance comparison"""
    for per_col in (False, True):
        gcm, zf, zh, prof, f, b = _outputs(96, per_column_grid=per_col)
        t = lambda d: {k: torch.from_numpy(numpy.ascontiguousarray(v)) for k, v in d.items()}          # noqa: E731
        g, p, zft, zht = t(gcm), t(prof), torch.from_numpy(zf), torch.from_numpy(zh)
        fp, bp = OracleEngine().plan_exchange(g, zft, zht, p, 1.0, 1.0, 900.0)
        fp.launch(), bp.launch()
        ok, det = bench.sample_check(fp.outputs, bp.outputs, g, p, zft, zht, 1.0, 900.0, m=50)
        assert ok, det
        assert det["rows_checked"] == 50 and det["first_row"] == 0 and det["last_row"] == 95
        bp.outputs["f_T"][95, 90] += 1.0
        ok, det = bench.sample_check(fp.outputs, bp.outputs, g, p, zft, zht, 1.0, 900.0, m=50)
        assert not ok and any("f_T" in x for x in det["failures"])
        bp.outputs["f_T"][95, 90] -= 1.0
        # the same batch as float32 tensors: inputs rounded, outputs = the fp64 results of the ROUNDED inputs, rounded
        g32, p32 = {k: v.float() for k, v in g.items()}, {k: v.float() for k, v in p.items()}
        gr = {k: v.double() for k, v in g32.items()}
        pr = {k: v.double() for k, v in p32.items()}
        fp2, bp2 = OracleEngine().plan_exchange(gr, zft.float().double(), zht.float().double(), pr, 1.0, 1.0, 900.0)
        fp2.launch(), bp2.launch()
        fo = {k: (v.float() if v.dtype == torch.float64 else v) for k, v in fp2.outputs.items()}
        bo = {k: (v.float() if v.dtype == torch.float64 else v) for k, v in bp2.outputs.items()}
        ok, det = bench.sample_check(fo, bo, g32, p32, zft.float(), zht.float(), 1.0, 900.0, m=50)
        assert ok and "fp32_vs_fp64_oracle" in det, det


def test_live_traffic_measurement_falls_back_without_a_gpu(monkeypatch):
    """bench.live_pmc_traffic (round-4 verdict, weak 13: `roofline.traffic` measured inside the bench run by two rocprofv3 --pmc
    child processes): without a GPU -- or without rocprofv3 -- it returns (None, reason) and the line keeps the constant of
    profiles/traffic.json, labelled as such; it never raises."""
    res, why = bench.live_pmc_traffic(16, timeout_s=120)
    assert res is None and isinstance(why, str) and why
    monkeypatch.setattr("shutil.which", lambda name: None)
    monkeypatch.setattr(bench.os.path, "exists", lambda p: False)
    assert bench.live_pmc_traffic(16) == (None, "rocprofv3 not found")
