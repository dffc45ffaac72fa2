"""multi.MultiDeviceEngine: every GPU of the node behind the ONE master process the reference has
(splib/splib.py:146-154) -- the column batch split into contiguous row blocks, one per device, no collective.

CPU: two (three) oracle-backed test engines stand for the devices; the partitioning, the per-(device, array) copies of
``transfer.ShardedArena`` and the per-device plans must reproduce the single-engine run of the same closed loop bit
for bit (same arithmetic, different placement).  GPU (-m gpu): two real ``Engine`` objects on the one card of the
test box, through the HIP kernels, against the single-engine path."""
import numpy
import pytest

from sp_coupler_amd import models, spcpl
from sp_coupler_amd.multi import MultiDeviceEngine, describe_partition
from sp_coupler_amd.sharding import shard_bounds


def _closed_loop(engine, n_les, nsteps, batched, cplsurf=False, conservative=False, nG=19, nL=40, per_column_grid=False):
    from sp_coupler_amd.driver import Coupler
    spcpl.set_engine(engine)
    try:
        if batched:
            gcm, les = models.make_batched_models(n_les, nG=nG, nL=nL, seed=5)
        else:
            gcm, les = models.make_models(n_les, nG=nG, nL=nL, seed=5)
            if per_column_grid:          # every LES on a grid of its own: zf / zh become [n x nL] arrays, sharded by rows too
                for i, m in enumerate(les):
                    m.zf_cache, m.zh_cache = m.zf_cache * (1 + 0.01 * i), m.zh_cache * (1 + 0.01 * i)
        cpl = Coupler(gcm, les, cplsurf=cplsurf, conservative_coarsening=conservative, les_forcing_factor=0.9)
        cpl.run(nsteps)
        b = spcpl.current_batch()
        heights = None if batched else (numpy.asarray(les[-1].gcm_Zf), numpy.asarray(les[-1].gcm_Zh))
        idx = numpy.array(spcpl._index_map(b))
        kinds = (type(b.buf.gcm_in).__name__, sorted({type(p).__name__ for p in b.buf.plans.values()}))
        return {k: v.copy() for k, v in gcm.state.items()}, heights, idx, kinds
    finally:
        spcpl.set_engine(None)


@pytest.mark.parametrize("ndev,n_les,batched,cplsurf,conservative", [(2, 7, False, False, False), (3, 8, True, True, False),
                                                                     (2, 5, True, False, True), (3, 2, False, True, False)])
def test_row_blocks_over_several_engines_equal_one_engine(ndev, n_les, batched, cplsurf, conservative):
    from tests.fake_engine import OracleEngine
    ref = _closed_loop(OracleEngine(), n_les, 3, batched, cplsurf, conservative)
    multi = MultiDeviceEngine([OracleEngine() for _ in range(ndev)], min_cols_per_device=1)
    assert multi.devices_for(n_les) == min(ndev, n_les)
    assert multi.bounds_for(n_les)[:min(ndev, n_les) + 1] == shard_bounds(n_les, min(ndev, n_les))
    from sp_coupler_amd.transfer import Sharded
    before = Sharded.gather_calls
    got = _closed_loop(multi, n_les, 3, batched, cplsurf, conservative)
    assert Sharded.gather_calls == before            # round 5: nothing on the step path gathers the batch onto one device
    for var in ref[0]:
        assert numpy.array_equal(ref[0][var], got[0][var], equal_nan=True), var
    if ref[1] is not None:
        assert numpy.array_equal(ref[1][0], got[1][0]) and numpy.array_equal(ref[1][1], got[1][1])
    assert numpy.array_equal(ref[2], got[2])
    assert ref[3][0] == "Arena" and got[3] == ("ShardedArena", ["MultiPlan"]), (ref[3], got[3])


def test_per_column_les_grids_are_sharded_with_their_rows():
    from tests.fake_engine import OracleEngine
    ref = _closed_loop(OracleEngine(), 7, 2, False, per_column_grid=True)
    got = _closed_loop(MultiDeviceEngine([OracleEngine() for _ in range(3)], min_cols_per_device=1), 7, 2, False, per_column_grid=True)
    for var in ref[0]:
        assert numpy.array_equal(ref[0][var], got[0][var], equal_nan=True), var
    assert numpy.array_equal(ref[2], got[2]) and got[3][0] == "ShardedArena"


def test_sharded_arena_partial_copies_and_host_only_rows():
    """transfer.ShardedArena: one pinned host buffer with full-length views, rows [0, rows) mirrored block-wise per device
    (host rows beyond `rows` -- the GCM's extra output columns -- stay host-only), copies of a sub-range of the arrays"""
    import torch
    from sp_coupler_amd.transfer import ShardedArena
    a = ShardedArena(["cpu", "cpu", "cpu"], [0, 3, 5, 5], [("A", (7, 4), torch.float64), ("B", (5,), torch.int32),
                                                             ("C", (5, 2), torch.float64)], rows=5)
    a.hn["A"][:] = numpy.arange(28).reshape(7, 4)
    a.hn["B"][:] = numpy.arange(5)
    a.hn["C"][:] = -1.0
    a.upload(upto="B")
    assert [tuple(p.shape) for p in a.d["A"].parts] == [(3, 4), (2, 4), (0, 4)] and a.d["A"].shape == (5, 4)
    assert torch.equal(a.d["A"].gather(), torch.from_numpy(a.hn["A"][:5])) and a.d["B"].parts[1].tolist() == [3, 4]
    for p in a.d["C"].parts:
        p.fill_(7.0)
    a.hn["A"][:] = 0
    a.download(upto="C", start="C")                       # only C comes back
    assert (a.hn["C"] == 7.0).all() and (a.hn["A"] == 0).all()
    a.download(upto="A")
    assert numpy.array_equal(a.hn["A"][:5], numpy.arange(20).reshape(5, 4)) and (a.hn["A"][5:] == 0).all()
    with pytest.raises(IndexError):
        a.d["A"][1:3]
    with pytest.raises(ValueError):
        ShardedArena(["cpu"], [0, 5], [("x", (3,), torch.float64)], rows=5)


def test_small_batches_stay_on_one_device_and_the_partition_is_the_sharding_one():
    from tests.fake_engine import OracleEngine
    m = MultiDeviceEngine([OracleEngine() for _ in range(8)])
    assert m.devices_for(1024) == 1 and m.bounds_for(1024) == [0, 1024] + [1024] * 7
    assert m.devices_for(35718) == 8 and m.bounds_for(35718) == shard_bounds(35718, 8)
    assert m.devices_for(5000) == 2 and m.bounds_for(5000)[:3] == [0, 2500, 5000]
    assert "rows 0-2500" in describe_partition(m, 5000)


def test_get_engine_inside_a_rank_per_gpu_job_takes_that_ranks_device(monkeypatch):
    """one-rank-per-GPU jobs (bench.py --gpus N, sharding.ShardedExchange) must not have every rank grab every GPU"""
    import sp_coupler_amd.spcpl as sp
    made = []
    monkeypatch.setattr(sp, "Engine", lambda *a, **k: made.append(a) or object())
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("LOCAL_RANK", "3")
    sp.set_engine(None)
    try:
        sp.get_engine()
        assert made == [("cuda:3",)]
    finally:
        sp.set_engine(None)


def test_all_gpus_behind_one_process_is_opt_in(monkeypatch):
    """several visible GPUs no longer mean MultiDeviceEngine silently -- SPC_DEVICES asks for it"""
    import torch
    import sp_coupler_amd.multi as mm
    import sp_coupler_amd.spcpl as sp
    made = []
    class FakeEngine:
        device = "cuda:0"

        def __init__(self, *a, **k):
            made.append(("engine", a))
    monkeypatch.setattr(sp, "Engine", FakeEngine)

    class FakeMulti:
        engines, min_cols_per_device = [], 2048

        def __init__(self, *a, **k):
            made.append(("multi", a))
    monkeypatch.setattr(mm, "MultiDeviceEngine", FakeMulti)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("SPC_DEVICES", raising=False)
    try:
        sp.set_engine(None)
        sp.get_engine()
        assert made == [("engine", ())]
        monkeypatch.setenv("SPC_DEVICES", "all")
        sp.set_engine(None)
        sp.get_engine()
        assert made[-1][0] == "multi"
        monkeypatch.setenv("SPC_SINGLE_GPU", "1")
        sp.set_engine(None)
        sp.get_engine()
        assert made[-1][0] == "engine"
    finally:
        sp.set_engine(None)


@pytest.mark.gpu
@pytest.mark.parametrize("batched,cplsurf", [(True, False), (False, True)])
def test_two_engines_on_one_gpu_through_the_hip_kernels(batched, cplsurf):
    from sp_coupler_amd.engine import Engine
    ref = _closed_loop(Engine("cuda:0"), 700, 2, batched, cplsurf, nG=91, nL=160)
    multi = MultiDeviceEngine([Engine("cuda:0"), Engine("cuda:0")], min_cols_per_device=100)
    assert multi.devices_for(700) == 2
    got = _closed_loop(multi, 700, 2, batched, cplsurf, nG=91, nL=160)
    for var in ref[0]:
        assert numpy.array_equal(ref[0][var], got[0][var], equal_nan=True), var
    assert numpy.array_equal(ref[2], got[2])


@pytest.mark.gpu
def test_multi_device_exchange_plans_equal_the_single_engine_plans_bitwise():
    """Engine.plan_exchange (what bench.py times) vs MultiDeviceEngine.plan_exchange on the same 5000 columns"""
    import torch
    from sp_coupler_amd import synthetic
    from sp_coupler_amd.engine import Engine
    eng = Engine("cuda:0")
    gcm, zf, zh, prof = synthetic.make_batch(5000, 91, 160, seed=77, couple_surface=False)
    g = {k: torch.from_numpy(v).to(eng.device) for k, v in gcm.items()}
    p = {k: torch.from_numpy(v).to(eng.device) for k, v in prof.items()}
    zf_d, zh_d = torch.from_numpy(zf).to(eng.device), torch.from_numpy(zh).to(eng.device)
    fp, bp = eng.plan_exchange(g, zf_d, zh_d, p, 1.0, 1.0, 900.0)
    fp.launch(), bp.launch()
    multi = MultiDeviceEngine([eng, Engine("cuda:0"), Engine("cuda:0")], min_cols_per_device=1000)
    gs = {k: multi.to_devices(v, rows=5000) for k, v in gcm.items()}
    ps = {k: multi.to_devices(v, rows=5000) for k, v in prof.items()}
    mfp, mbp = multi.plan_exchange(gs, multi.to_devices(zf), multi.to_devices(zh), ps, 1.0, 1.0, 900.0)
    assert [d is not None for d in mfp.describe()] == [True, True, True]
    mfp.launch(), mbp.launch()
    multi.synchronize()
    for plan, mplan in ((fp, mfp), (bp, mbp)):
        for k, t in plan.outputs.items():
            assert numpy.array_equal(t.cpu().numpy(), mplan.outputs[k].gather().cpu().numpy(), equal_nan=True), k


@pytest.mark.gpu
def test_engines_with_streams_of_their_own_on_one_gpu():
    """each engine of a MultiDeviceEngine may carry its own stream (Engine(stream=...)): chunks of one batch then run on
    different streams of one GPU; same bits as the plain engine"""
    import torch
    from sp_coupler_amd.engine import Engine
    ref = _closed_loop(Engine("cuda:0"), 900, 3, True, True, nG=91, nL=160)
    dev = torch.device("cuda:0")
    st = MultiDeviceEngine([Engine(dev), Engine(dev, stream=torch.cuda.Stream(dev)), Engine(dev, stream=torch.cuda.Stream(dev))],
                           min_cols_per_device=200)
    got = _closed_loop(st, 900, 3, True, True, nG=91, nL=160)
    for var in ref[0]:
        assert numpy.array_equal(ref[0][var], got[0][var], equal_nan=True), var
    assert numpy.array_equal(ref[2], got[2]) and got[3] == ("ShardedArena", ["MultiPlan"])
    # a batch below the threshold stays in one piece on the first engine: plain arenas, plain plans
    small = _closed_loop(MultiDeviceEngine([Engine(dev), Engine(dev)]), 300, 2, True, True, nG=91, nL=160)
    assert small[3][0] == "Arena" and "MultiPlan" not in small[3][1]


@pytest.mark.gpu
def test_bench_in_process_extra_runs_on_two_engines():
    import argparse
    import bench
    r = bench.in_process_all_gpus([0, 0], 1.0, 900.0, argparse.Namespace(cols=6000, cols_per_block=0, steps=3, warmup=1))
    assert r["value"] > 0 and r["distinct_devices"] == 1 and "rows 0-3000" in r["partition"]
    # every device's row block proves itself against the plain-C oracle (first and last row of the block included)
    assert r["verified"] is True and [c["rows"] for c in r["verified_detail"]] == [[0, 3000], [3000, 6000]]
    assert all(c["first_row"] == 0 and c["last_row"] == 2999 and c["failures"] == [] for c in r["verified_detail"])


# ---- round 5 (round-4 verdict, weak 9 / next 4): the slow paths, the sputils helpers and the nudge run PER DEVICE on that device's
# rows; nothing gathers the batch onto one device any more (transfer.Sharded.gather_calls counts) -----------------------------------
class _CountingEngine:
    """a test engine that counts what reaches it (rows per call)"""

    def __init__(self, inner):
        self.inner, self.calls = inner, []

    def __getattr__(self, name):
        attr = getattr(self.inner, name)
        if name in ("forward", "backward", "diagnostics", "cloud_indices", "surface_fluxes", "variability_nudge", "exner", "interp",
                    "searchsorted", "interp_c", "rms"):
            def counted(*a, **k):
                flat = [t for x in a for t in (x.values() if isinstance(x, dict) else [x]) if hasattr(x, "shape") or isinstance(x, dict)]
                rows = next((int(t.shape[0]) for t in flat if t.dim() >= 2), int(flat[0].shape[0]))    # rows of the first matrix
                self.calls.append((name, rows))
                return attr(*a, **k)
            return counted
        return attr


def _three():
    from tests.fake_engine import OracleEngine
    return [_CountingEngine(OracleEngine()) for _ in range(3)]


def make_case(itot, jtot, ktot, seed):
    """one synthetic LES for the nudge: the fields of tests/test_vnudge.make_les_fields + the zero-mean random plane R"""
    from tests.test_vnudge import make_les_fields
    c = make_les_fields(itot, jtot, ktot, seed)
    R = numpy.random.default_rng(seed + 1000).normal(size=(itot, jtot))
    c["R"] = R - R.sum() / (itot * jtot)
    return c


def test_convenience_forms_run_per_device_without_a_gather():
    import torch
    from sp_coupler_amd import synthetic
    from sp_coupler_amd.transfer import Sharded
    from tests.fake_engine import OracleEngine
    n = 11
    gcm, zf, zh, prof = synthetic.make_batch(n, 19, 40, seed=3)
    one = OracleEngine()
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}                                 # noqa: E731
    ref_f = one.forward(t(gcm), torch.from_numpy(zf), t(prof), 1.0, 900.0, zh=torch.from_numpy(zh), want_profiles=True, couple_surface=True)
    ref_b = one.backward(t(gcm), torch.from_numpy(zf), t(prof), 1.0, 900.0, conservative=True, zh=torch.from_numpy(zh))
    ref_d = one.diagnostics(t(gcm), torch.from_numpy(zf), t(prof))
    engines = _three()
    multi = MultiDeviceEngine(engines, min_cols_per_device=2)
    before = Sharded.gather_calls
    g = {k: multi.to_devices(v, rows=n) for k, v in gcm.items()}
    p = {k: multi.to_devices(v, rows=n) for k, v in prof.items()}
    zf_s, zh_s = multi.to_devices(zf), multi.to_devices(zh)
    f = multi.forward(g, zf_s, p, 1.0, 900.0, zh=zh_s, want_profiles=True, couple_surface=True)
    b = multi.backward(g, zf_s, p, 1.0, 900.0, conservative=True, zh=zh_s)
    d = multi.diagnostics(g, zf_s, p)
    idx = multi.cloud_indices(zh_s, d["Zh"])
    for res, ref in ((f, ref_f), (b, ref_b), (d, ref_d)):
        for k, v in ref.items():
            assert isinstance(res[k], Sharded) and res[k].bounds == multi.bounds_for(n)
            assert numpy.array_equal(res[k].to_host(), v.numpy(), equal_nan=True), k
    assert numpy.array_equal(idx.to_host(), ref_f["idx"].numpy())
    assert Sharded.gather_calls == before
    # every engine saw ITS rows only: 4 + 4 + 3
    for e, rows in zip(engines, (4, 4, 3)):
        assert {c[1] for c in e.calls} == {rows}, e.calls
        assert [c[0] for c in e.calls] == ["forward", "backward", "diagnostics", "cloud_indices"]


def test_sputils_helpers_shard_their_rows_over_the_engines():
    """sp_coupler_amd.sputils (the reference's helper names, NumPy in / NumPy out) on a MultiDeviceEngine: a batch of rows is
    dealt out block by block; a call on one column, or on device tensors, stays on the first engine"""
    from oracle import spcpl_oracle as orc
    from sp_coupler_amd import sputils, synthetic
    from sp_coupler_amd.transfer import Sharded
    from tests.fake_engine import OracleEngine
    n = 10
    gcm, zf, zh, prof = synthetic.make_batch(n, 19, 40, seed=8)
    Zf = (gcm["Zgfull"] - gcm["Zghalf"][:, -1:]) / 9.81
    Zh = (gcm["Zghalf"] - gcm["Zghalf"][:, -1:]) / 9.81
    xp, fp = numpy.ascontiguousarray(Zf[:, ::-1]), numpy.ascontiguousarray(gcm["U"][:, ::-1])
    engines = _three()
    spcpl.set_engine(MultiDeviceEngine(engines, min_cols_per_device=2))
    before = Sharded.gather_calls
    try:
        got = {"exner": sputils.exner(gcm["Pfull"]), "iexner": sputils.iexner(gcm["Pfull"]),
               "interp": sputils.interp(zf, xp, fp), "interp_lr": sputils.interp(zf, xp, fp, left=-1.0, right=7.0),
               "searchsorted": sputils.searchsorted(zh, Zh, side="right"), "interp_c": sputils.interp_c(Zh, zh, prof["QT"], prof["Rhobf"]),
               "interp_rho": sputils.interp_rho(Zh, zh, prof["Rhobf"]), "rms_rows": sputils.rms(prof["U"], axis=-1),
               "rms_all": sputils.rms(prof["U"]), "one_column": sputils.interp(zf, xp[3], fp[3])}
    finally:
        spcpl.set_engine(None)
    assert Sharded.gather_calls == before
    assert numpy.array_equal(got["exner"], orc.exner(gcm["Pfull"])) and numpy.array_equal(got["iexner"], orc.iexner(gcm["Pfull"]))
    want = numpy.stack([numpy.interp(zf, xp[r], fp[r]) for r in range(n)])
    assert numpy.array_equal(got["interp"], want) and numpy.array_equal(got["one_column"], want[3])
    assert numpy.array_equal(got["interp_lr"], numpy.stack([numpy.interp(zf, xp[r], fp[r], left=-1.0, right=7.0) for r in range(n)]))
    assert numpy.array_equal(got["searchsorted"], numpy.stack([numpy.searchsorted(zh, Zh[r], side="right") for r in range(n)]))
    assert numpy.array_equal(got["interp_c"], numpy.stack([orc.interp_c(Zh[r], zh, prof["QT"][r], prof["Rhobf"][r]) for r in range(n)]))
    with numpy.errstate(all="ignore"):
        assert numpy.array_equal(got["interp_rho"], numpy.stack([orc.interp_rho(Zh[r], zh, prof["Rhobf"][r]) for r in range(n)]))
    assert numpy.array_equal(got["rms_rows"], numpy.array([orc.rms(r) for r in prof["U"]])) and got["rms_all"] == orc.rms(prof["U"].reshape(-1))
    # the batched calls reached every engine with its 4 / 4 / 2 rows (ceil(10 / 3) per block); the whole-array rms and the
    # one-column call only the first
    ops = ["exner", "exner", "interp", "interp", "searchsorted", "interp_c", "interp_c", "rms"]
    for e, rows in zip(engines[1:], (4, 2)):
        assert e.calls == [(op, rows) for op in ops], e.calls
    assert engines[0].calls[:8] == [(op, 4) for op in ops] and [c[0] for c in engines[0].calls[8:]] == ["rms", "interp"]


def test_variability_nudge_deals_the_les_out_over_the_engines(monkeypatch):
    """spcpl._vnudge_launch: each engine nudges ITS LES (contiguous blocks, sharding.shard_bounds), in chunks of what fits it;
    same bits as one engine, the 3-D fields never meet on one device"""
    from tests.fake_engine import OracleEngine
    n = 5
    cases = [make_case(6, 5, 7, seed=40 + i) for i in range(n)]
    F = {k: numpy.stack([c[k] for c in cases]) for k in ("qt", "qsat", "ql_av", "qt_av", "ql_ref", "presf", "thl", "ql")}
    Rs = numpy.stack([c["R"] for c in cases])
    spcpl.set_engine(OracleEngine())
    try:
        ref = spcpl._vnudge_launch({k: v.copy() for k, v in F.items()}, Rs, True)
    finally:
        spcpl.set_engine(None)
    engines = _three()
    spcpl.set_engine(MultiDeviceEngine(engines, min_cols_per_device=1))
    monkeypatch.setattr(spcpl, "_vnudge_chunk", lambda eng, n_, *a: 1)           # one LES per launch: two rounds on the first engines
    try:
        got = spcpl._vnudge_launch({k: v.copy() for k, v in F.items()}, Rs, True)
    finally:
        spcpl.set_engine(None)
    for k in ref[0]:
        assert numpy.array_equal(ref[0][k], got[0][k], equal_nan=True), k
    assert numpy.array_equal(ref[1], got[1]) and numpy.array_equal(ref[2], got[2])
    assert [len(e.calls) for e in engines] == [2, 2, 1] and all(c == ("variability_nudge", 1) for e in engines for c in e.calls)


@pytest.mark.gpu
def test_slow_paths_helpers_and_nudge_on_two_engines_of_one_gpu():
    """the same through the HIP kernels: two Engines on the one card stand for two devices; bit-equal to one Engine, no gather"""
    import torch
    from sp_coupler_amd import sputils, synthetic
    from sp_coupler_amd.engine import Engine
    from sp_coupler_amd.transfer import Sharded
    n = 2501
    gcm, zf, zh, prof = synthetic.make_batch(n, 91, 160, seed=12)
    one = Engine("cuda:0")
    up = lambda d: {k: torch.from_numpy(v).to(one.device) for k, v in d.items()}                  # noqa: E731
    zf_d, zh_d = torch.from_numpy(zf).to(one.device), torch.from_numpy(zh).to(one.device)
    ref_f = one.forward(up(gcm), zf_d, up(prof), 1.0, 900.0, zh=zh_d, want_profiles=True, couple_surface=True)
    ref_b = one.backward(up(gcm), zf_d, up(prof), 1.0, 900.0, conservative=True, zh=zh_d)
    ref_d = one.diagnostics(up(gcm), zf_d, up(prof))
    multi = MultiDeviceEngine([Engine("cuda:0"), Engine("cuda:0", stream=torch.cuda.Stream("cuda:0"))], min_cols_per_device=500)
    before = Sharded.gather_calls
    g = {k: multi.to_devices(v, rows=n) for k, v in gcm.items()}
    p = {k: multi.to_devices(v, rows=n) for k, v in prof.items()}
    zf_s, zh_s = multi.to_devices(zf), multi.to_devices(zh)
    f = multi.forward(g, zf_s, p, 1.0, 900.0, zh=zh_s, want_profiles=True, couple_surface=True)
    b = multi.backward(g, zf_s, p, 1.0, 900.0, conservative=True, zh=zh_s)
    d = multi.diagnostics(g, zf_s, p)
    for res, ref in ((f, ref_f), (b, ref_b), (d, ref_d)):
        for k, v in ref.items():
            assert numpy.array_equal(res[k].to_host(), v.cpu().numpy(), equal_nan=True), k
    Zh = ref_d["Zh"].cpu().numpy()
    spcpl.set_engine(one)
    try:
        r1 = (sputils.iexner(gcm["Pfull"]), sputils.interp_c(Zh, zh, prof["QT"], prof["Rhobf"]), sputils.rms(prof["U"], axis=-1),
              sputils.searchsorted(zh, Zh, side="right"))
        cases = [make_case(64, 64, 40, seed=70 + i) for i in range(3)]
        F = {k: numpy.stack([c[k] for c in cases]) for k in ("qt", "qsat", "ql_av", "qt_av", "ql_ref", "presf", "thl", "ql")}
        Rs = numpy.stack([c["R"] for c in cases])
        v1 = spcpl._vnudge_launch({k: v.copy() for k, v in F.items()}, Rs, True)
        spcpl.set_engine(multi)
        r2 = (sputils.iexner(gcm["Pfull"]), sputils.interp_c(Zh, zh, prof["QT"], prof["Rhobf"]), sputils.rms(prof["U"], axis=-1),
              sputils.searchsorted(zh, Zh, side="right"))
        v2 = spcpl._vnudge_launch({k: v.copy() for k, v in F.items()}, Rs, True)
    finally:
        spcpl.set_engine(None)
    for a, c in zip(r1, r2):
        assert numpy.array_equal(a, c, equal_nan=True)
    for k in v1[0]:
        assert numpy.array_equal(v1[0][k], v2[0][k], equal_nan=True), k
    assert numpy.array_equal(v1[1], v2[1]) and numpy.array_equal(v1[2], v2[2])
    assert Sharded.gather_calls == before
