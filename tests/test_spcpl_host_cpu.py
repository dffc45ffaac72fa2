"""CPU tests of the product's HOST LOGIC (sp_coupler_amd.spcpl fan-out and profile registry, driver
sequencing, spifs hook) with the test-only oracle-backed engine of tests/fake_engine.py injected through
spcpl.set_engine().  Because the injected engine and the reference sequencing (tests/ref_driver.py) share
the oracle's arithmetic, every setter argument must be BIT-identical: any difference is a host-logic bug
(wrong row, wrong order, stale profile, wrong firststep handling).  The kernels themselves are tested on
the GPU (tests/test_parity_gpu.py, tests/test_spcpl_gpu.py)."""
import numpy
import pytest

from sp_coupler_amd import models, spcpl
from tests.fake_engine import OracleEngine
from tests.ref_driver import RefCoupler
from tests.test_spcpl_gpu import Recorder, _by_key


@pytest.fixture(autouse=True)
def oracle_engine():
    spcpl.set_engine(OracleEngine())
    yield
    spcpl.set_engine(None)
    spcpl.writer = None


@pytest.mark.parametrize("cplsurf,conservative", [(False, False), (True, False), (False, True)])
def test_driver_sequencing_bit_identical_to_reference_loop(cplsurf, conservative):
    from sp_coupler_amd.driver import Coupler
    nsteps, n_les = 3, 5
    gcm_a, les_a = models.make_models(n_les, nG=19, nL=160, seed=3)
    gcm_b, les_b = models.make_models(n_les, nG=19, nL=160, seed=3)
    rec = Recorder(gcm_a, les_a)
    cpl = Coupler(gcm_a, les_a, cplsurf=cplsurf, les_forcing_factor=0.8, gcm_forcing_factor=1.2,
                  conservative_coarsening=conservative)
    ref = RefCoupler(gcm_b, les_b, cplsurf=cplsurf, les_forcing_factor=0.8, gcm_forcing_factor=1.2,
                     conservative=conservative)
    for _ in range(nsteps):
        cpl.step()
        ref.step()
    got, want = _by_key(rec.log), _by_key(ref.log)
    assert set(got) == set(want)
    for key in want:
        assert len(got[key]) == nsteps
        for s in range(nsteps):
            assert numpy.array_equal(got[key][s], want[key][s], equal_nan=True), (key, s)
    for var in gcm_b.state:
        assert numpy.array_equal(gcm_a.state[var], gcm_b.state[var]), var
    for la, lb in zip(les_a, les_b):
        assert numpy.array_equal(la.p["THL"], lb.p["THL"]) and la.rain == lb.rain
        assert la.gcm_Zf.shape == (19,) and la.gcm_Zh.shape == (20,) and la.ql_ref.shape == (160,)
    assert not cpl.firststep and len(cpl.timing_rows) == nsteps


def test_reference_loop_shape_with_resolved_profiles_next_to_pending_requests():
    from tests.ref_driver import check_reference_loop_shape
    check_reference_loop_shape(tol=0.0)          # the test engine shares the oracle's arithmetic: bit-identical


def test_per_les_calls_need_all_profiles_and_stale_profiles_are_not_reused():
    gcm, les_models = models.make_models(3, nG=19, nL=160, seed=6)
    spcpl.gather_gcm_data(gcm, les_models, False)
    with pytest.raises(RuntimeError, match="profiles of 2 columns are unknown"):
        spcpl.set_les_forcings(les_models[0], gcm, True, False, {"U": 0}, dt_gcm=900.0, factor=1.0, couple_surface=False)
    with pytest.raises(RuntimeError, match="gather_gcm_data"):
        spcpl.set_les_forcings(models.SyntheticLES(gcm, 2), gcm, True, True, {}, 900.0, 1.0, False)

    # first step: live getters, all columns in one (fake) launch, request dict per column
    reqs = spcpl.set_les_forcings_batched(les_models, gcm, True, True, {}, 900.0, 1.0, False)
    assert len(reqs) == 3 and all(set(r) == {"U", "V", "THL", "QT", "SP", "QL", "QLp"} for r in reqs)
    b1 = spcpl.current_batch()
    assert b1.fwd is not None and b1.fwd["f_u"].shape == (3, 160)
    # a second gather of the SAME columns refills the batch (per-LES handles stay attached, per-step results are
    # dropped); the profiles fetched after the LES step carry over to it
    for les in les_models:
        les.evolve_model(900.0)
        spcpl.get_les_profiles(les, True)
    step1 = b1.step_id
    spcpl.gather_gcm_data(gcm, les_models, False)
    b2 = spcpl.current_batch()
    assert b2 is b1 and b2.step_id == step1 + 1 and b2.fwd is None and b2.bwd is None and len(b2.profiles) == 3
    assert les_models[1]._spc_batch is b2 and les_models[1]._spc_row == 1
    spcpl.set_les_forcings(les_models[1], gcm, True, False, None, dt_gcm=900.0, factor=1.0, couple_surface=False)
    assert b2.fwd is not None
    # a different set of columns makes a new batch (and the profiles of the columns it shares carry over)
    spcpl.gather_gcm_data(gcm, les_models[:2], False)
    b3 = spcpl.current_batch()
    assert b3 is not b2 and b3.n == 2 and len(b3.profiles) == 2 and les_models[0]._spc_batch is b3
    # convert_profiles (init path) must not freeze the heights: after the next gather they are those of the NEW GCM state
    spcpl.convert_profiles(les_models[0], write=False)
    z_old = numpy.array(les_models[0].gcm_Zf)
    gcm.state["Zgfull"][les_models[0].grid_index] += 98.1                  # + 10 m at every full level
    spcpl.gather_gcm_data(gcm, les_models[:2], False)
    assert numpy.allclose(numpy.asarray(les_models[0].gcm_Zf) - z_old, 10.0)
    b3 = spcpl.current_batch()
    # heights are computed on first read, per step, and behave like the float64 row
    spcpl.gather_gcm_data(gcm, les_models[:2], False)
    zf = les_models[0].gcm_Zf
    assert b3.diag_host is None and zf.shape == (19,) and b3.diag_host is not None
    assert numpy.array_equal(numpy.asarray(zf), b3.diag_host["Zf"][0]) and float(zf[0]) > float(zf[-1]) and len(les_models[0].gcm_Zh) == 20
    assert numpy.array_equal(zf - numpy.asarray(zf), numpy.zeros(19))


def test_unit_wrapper_and_quantity_inputs():
    class Q:                                   # minimal stand-in for an AMUSE quantity
        def __init__(self, number, unit):
            self.number, self.unit = number, unit
    gcm, les_models = models.make_models(2, nG=19, nL=160, seed=8)
    seen = []
    spcpl.set_unit_wrapper(lambda name, arr: (seen.append(name), Q(arr, spcpl.output_units[name]))[1])
    try:
        orig = gcm.get_profile_fields
        gcm.get_profile_fields = lambda var, cols: Q(orig(var, cols), "si")
        spcpl.gather_gcm_data(gcm, les_models, False)
        spcpl.set_les_forcings_batched(les_models, gcm, False, True, {}, Q(900.0, "s"), 1.0, False)
        assert {"f_u", "f_v", "f_thl", "f_qt", "f_ps", "f_ql", "ql_ref"} <= set(seen) and "Zf" not in seen
        assert les_models[0].gcm_Zf.number.shape == (19,) and les_models[1].gcm_Zh.unit == "m"      # on first read
        assert {"Zf", "Zh"} <= set(seen)
        assert les_models[0].tend["U"].shape == (160,)            # the model saw the bare numbers
    finally:
        spcpl.set_unit_wrapper(None)


def test_spifs_hook_writes_once_per_launch(tmp_path):
    from sp_coupler_amd import spio
    from sp_coupler_amd.driver import Coupler
    gcm, les_models = models.make_models(3, nG=19, nL=160, seed=2)
    path = str(tmp_path / "spifs.nc")
    calls = []
    w = spio.SpifsWriter(path, [m.grid_index for m in les_models], [0] * 3, [0] * 3, les_models[0].zf_cache, 19)
    orig = w.write
    w.write = lambda rows=None, **kw: (calls.append(sorted(kw)), orig(rows=rows, **kw))[1]
    spcpl.writer = w
    Coupler(gcm, les_models, cplsurf=True, write=True).run(2)
    w.close()
    assert len(calls) == 2 * 3                 # per step: forward state+forcings, surface block, backward block
    c = spio.read_column(path, 2)
    assert c["Time"].tolist() == [0.0, 1800.0] and numpy.isfinite(c["f_T"]).all() and numpy.isfinite(c["t"]).all()


def test_spinup_steps_and_write_les_profiles(tmp_path):
    """step_spinup (splib/splib.py:355-402): forcings with dt = spinup length, LES stepped, slab means written"""
    from sp_coupler_amd import spio
    from sp_coupler_amd.driver import Coupler
    gcm, les_models = models.make_models(3, nG=19, nL=160, seed=12)
    path = str(tmp_path / "spifs.nc")
    spcpl.writer = spio.SpifsWriter(path, [m.grid_index for m in les_models], [0] * 3, [0] * 3, les_models[0].zf_cache, 19)
    cpl = Coupler(gcm, les_models, write=True)
    thl0 = les_models[0].p["THL"].copy()
    for s_ in range(2):
        spcpl.writer.update_time(100.0 * (s_ + 1))
        cpl.step_spinup(100.0, les_spinup_forcing_factor=0.5)
    spcpl.writer.close()
    spcpl.writer = None
    assert les_models[0].model_time == 200.0 and not numpy.array_equal(les_models[0].p["THL"], thl0)
    c = spio.read_column(path, 1)
    assert numpy.array_equal(c["thl"][1], les_models[1].p["THL"].astype(numpy.float32))
    assert numpy.isfinite(c["t"]).all() and numpy.isfinite(c["ql_water"]).all()
    one = spcpl.write_les_profiles(les_models[2])
    assert one["u"].shape == (160,) and numpy.array_equal(one["thl"], les_models[2].p["THL"])


def test_extra_output_columns_are_converted_and_written(tmp_path):
    """gather_gcm_data's extra output columns (spcpl.py:89-129): output_column_conversion + surface fluxes of
    columns WITHOUT an LES, written into their rows of the spifs file."""
    from oracle import spcpl_oracle as orc
    from sp_coupler_amd import spio
    gcm, les_models = models.make_models(2, npoints=8, nG=19, nL=160, seed=5)
    extra = [5, 6]
    path = str(tmp_path / "spifs.nc")
    idxs = [m.grid_index for m in les_models] + extra
    spcpl.writer = spio.SpifsWriter(path, idxs, [0] * 4, [0] * 4, les_models[0].zf_cache, 19)
    spcpl.writer_rows = {5: 2, 6: 3}
    spcpl.writer.update_time(900.0)
    batch = spcpl.gather_gcm_data(gcm, les_models, True, extra, write=True)
    spcpl.writer.close()
    spcpl.writer, spcpl.writer_rows = None, {}
    assert batch.n == 2 and batch.gcm["T"].shape == (2, 19) and batch.gcm_host["T"].shape == (4, 19)
    c = spio.read_column(path, 3)                                     # GCM grid index 6
    col = {"T": gcm.state["T"][6], "SH": gcm.state["SH"][6], "QL": gcm.state["QL"][6], "QI": gcm.state["QI"][6],
           "Pf": gcm.state["Pfull"][6], "Ph": gcm.state["Phalf"][6], "Zgfull": gcm.state["Zgfull"][6],
           "Zghalf": gcm.state["Zghalf"][6]}
    orc.output_column_conversion(col)
    for k in ("Tv", "THL", "QT", "Zf", "Zh", "Ph"):
        assert numpy.array_equal(c[k][0], col[k].astype(numpy.float32)), k
    assert c["Psurf"][0] == numpy.float32(col["Psurf"])
    sf = {"Phalf": gcm.state["Phalf"][6], "T": gcm.state["T"][6]}
    sf.update({k: gcm.state[k][6] for k in ("Z0M", "Z0H", "QLflux", "QIflux", "SHflux", "TSflux")})
    z0m, z0h, wthl, wqt = orc.convert_surface_fluxes(sf)
    assert c["wthl"][0] == numpy.float32(wthl) and c["wqt"][0] == numpy.float32(wqt) and c["z0m"][0] == numpy.float32(z0m)
    assert numpy.isnan(spio.read_column(path, 0)["Tv"][0]).all()       # SP columns were not written by gather itself


def test_full_steps_with_extra_output_columns_and_a_writer(tmp_path):
    """Regression (round-1 advisor finding): a spifs file that ALSO holds extra output columns has n_les+extra
    rows; the per-step writes of the SP columns must land in rows 0..n_les-1 and leave the extra columns'
    rows (written by gather_gcm_data, spcpl.py:89-129) alone."""
    from sp_coupler_amd import spio
    from sp_coupler_amd.driver import Coupler
    gcm, les_models = models.make_models(2, npoints=8, nG=19, nL=160, seed=5)
    extra = [5, 6]
    path = str(tmp_path / "spifs.nc")
    idxs = [m.grid_index for m in les_models] + extra
    spcpl.writer = spio.SpifsWriter(path, idxs, [0] * 4, [0] * 4, les_models[0].zf_cache, 19)
    spcpl.writer_rows = {5: 2, 6: 3}
    Coupler(gcm, les_models, cplsurf=True, write=True, output_column_indices=extra).run(2)
    spcpl.writer.close()
    spcpl.writer, spcpl.writer_rows = None, {}
    sp, ex = spio.read_column(path, 1), spio.read_column(path, 3)
    assert sp["Time"].tolist() == [0.0, 1800.0]
    assert numpy.isfinite(sp["f_T"]).all() and numpy.isfinite(sp["f_u"]).all() and numpy.isfinite(sp["Tv"]).all()
    assert numpy.isfinite(ex["Tv"]).all() and numpy.isfinite(ex["wthl"]).all()      # extra column kept its rows
    assert numpy.isnan(ex["f_T"]).all()                                              # and received no SP tendencies


@pytest.mark.parametrize("cplsurf,conservative,write", [(False, False, False), (True, False, True), (False, True, False)])
def test_batched_model_protocol_equals_the_per_les_path(tmp_path, cplsurf, conservative, write):
    """The optional batched protocol (LES ensemble + GCM out=/set_profile_tendencies; models.py, INTEGRATION.md
    section 4) must drive the models to EXACTLY the state the reference-style per-LES calls produce: same kernels,
    same arithmetic, only the transport differs (one call per variable instead of ~20 per column)."""
    from sp_coupler_amd import spio
    from sp_coupler_amd.driver import Coupler
    n_les, nsteps = 6, 3
    gcm_a, les_a = models.make_models(n_les, nG=19, nL=160, seed=9)
    gcm_b, les_b = models.make_models(n_les, nG=19, nL=160, seed=9)
    gcm_b.__class__ = models.BatchedSyntheticGCM                       # same state, batched protocol switched on
    ens = models.SyntheticLESEnsemble.from_models(les_b)
    files = []
    for tag, gcm, les in (("a", gcm_a, les_a), ("b", gcm_b, ens)):
        if write:
            path = str(tmp_path / ("spifs_%s.nc" % tag))
            idx = [m.grid_index for m in les_a]
            spcpl.writer = spio.SpifsWriter(path, idx, [0] * n_les, [0] * n_les, les_a[0].zf_cache, 19)
            files.append(path)
        cpl = Coupler(gcm, les, cplsurf=cplsurf, les_forcing_factor=0.7, gcm_forcing_factor=1.3,
                      conservative_coarsening=conservative, write=write)
        cpl.run(nsteps)
        assert not cpl.firststep and len(cpl.timing_rows) == nsteps
        if write:
            spcpl.writer.close()
            spcpl.writer = None
    for var in gcm_a.state:
        assert numpy.array_equal(gcm_a.state[var], gcm_b.state[var]), var
    for i, m in enumerate(les_a):
        for k in ("U", "V", "THL", "QT", "QL", "T", "PS", "Rain"):
            assert numpy.array_equal(numpy.asarray(m.p[k]), ens.p[k][i]), (k, i)
    assert ens.model_time == les_a[0].model_time == 2700.0
    if write:
        ca, cb = spio.read_column(files[0], 4), spio.read_column(files[1], 4)
        assert set(ca) == set(cb)
        for k in ca:
            assert numpy.array_equal(ca[k], cb[k], equal_nan=True), k
        assert numpy.isfinite(cb["f_T"]).all() and numpy.isfinite(cb["rainrate"]).all()
    # the per-column faces of an ensemble speak the reference's per-LES protocol too
    assert ens[2].get_profile_U().shape == (160,) and ens[2].grid_index == les_a[2].grid_index


def test_spinup_with_an_ensemble_equals_the_per_les_spinup(tmp_path):
    """step_spinup (splib/splib.py:355-402) through the batched protocol: same LES state and same spifs rows as the
    per-LES calls."""
    from sp_coupler_amd import spio
    from sp_coupler_amd.driver import Coupler
    gcm_a, les_a = models.make_models(4, nG=19, nL=160, seed=12)
    gcm_b, les_b = models.make_models(4, nG=19, nL=160, seed=12)
    gcm_b.__class__ = models.BatchedSyntheticGCM
    ens = models.SyntheticLESEnsemble.from_models(les_b)
    out = []
    for tag, gcm, les in (("a", gcm_a, les_a), ("b", gcm_b, ens)):
        path = str(tmp_path / ("spin_%s.nc" % tag))
        spcpl.writer = spio.SpifsWriter(path, [m.grid_index for m in les_a], [0] * 4, [0] * 4, les_a[0].zf_cache, 19)
        cpl = Coupler(gcm, les, write=True)
        for s_ in range(2):
            spcpl.writer.update_time(100.0 * (s_ + 1))
            cpl.step_spinup(100.0, les_spinup_forcing_factor=0.5)
        spcpl.writer.close()
        spcpl.writer = None
        out.append(spio.read_column(path, 2))
    for i, m in enumerate(les_a):
        for k in ("U", "THL", "QT", "T"):
            assert numpy.array_equal(numpy.asarray(m.p[k]), ens.p[k][i]), (k, i)
    assert ens.model_time == les_a[0].model_time == 200.0
    for k in out[0]:
        assert numpy.array_equal(out[0][k], out[1][k], equal_nan=True), k


def test_variance_forcing_on_the_ensemble_path_equals_the_per_les_path():
    """Round-2 advisor: Coupler(ensemble, qt_forcing='variance') raised AttributeError on the first step with model time
    > 0 (ql_ref never handed over, no 3-D field access on the batched protocol).  The ensemble now offers
    get_fields_batched / set_fields_batched; the nudged 3-D qt (and thl with constant T) of every column must equal what
    the reference's per-LES loop (spcpl.py:377-382, variability_nudge per les) produces from the same fields and the
    same global random stream."""
    from tests.test_vnudge import FieldLES, make_les_fields
    n, nG, nL = 3, 19, 40
    gcm_a, les_a = models.make_models(n, nG=nG, nL=nL, seed=21)
    gcm_b = models.BatchedSyntheticGCM(gcm_a.npoints, nG, 21)
    ens = models.SyntheticLESEnsemble.from_models(les_a)
    fs = [make_les_fields(8, 8, nL, seed=60 + i) for i in range(n)]
    # per-LES side: the reference's face (get_field / get_profile / fields namespace)
    for les, f in zip(les_a, fs):
        les.f3, les.fields = f, FieldLES._Fields()
        les.get_field = lambda name, f=f: {"Qsat": f["qsat"], "QT": f["qt"], "THL": f["thl"], "QL": f["ql"]}[name].copy()
        les.get_profile = lambda name, f=f: {"QL": f["ql_av"], "QT": f["qt_av"]}[name].copy()
        les.p["presf"] = f["presf"].copy()
        les.model_time = 900.0
    # ensemble side: the same numbers stacked
    ens.attach_fields({"Qsat": numpy.stack([f["qsat"] for f in fs]), "QT": numpy.stack([f["qt"] for f in fs]),
                       "THL": numpy.stack([f["thl"] for f in fs]), "QL": numpy.stack([f["ql"] for f in fs])})
    ens.p["presf"] = numpy.stack([f["presf"] for f in fs])
    ens.model_time = 900.0
    slab = {"QL": numpy.stack([f["ql_av"] for f in fs]), "QT": numpy.stack([f["qt_av"] for f in fs])}
    orig = ens.get_profiles_batched

    def profiles(keys, out):                 # the nudge asks for the slab means of the 3-D fields (les.get_profile)
        if tuple(keys) == ("QL", "QT", "presf"):
            numpy.copyto(out["QL"], slab["QL"]); numpy.copyto(out["QT"], slab["QT"]); numpy.copyto(out["presf"], ens.p["presf"])
        else:
            orig(keys, out)
    ens.get_profiles_batched = profiles

    spcpl.gather_gcm_data(gcm_a, les_a, False, write=False)
    numpy.random.seed(11)
    spcpl.set_les_forcings_batched(les_a, gcm_a, True, True, {}, dt_gcm=900.0, factor=1.0, couple_surface=False,
                                   qt_forcing='variance', variability_nudge_constant_T=True)
    spcpl.gather_gcm_data(gcm_b, ens, False, write=False)
    numpy.random.seed(11)
    assert spcpl.set_les_forcings_batched(ens, gcm_b, True, True, None, dt_gcm=900.0, factor=1.0, couple_surface=False,
                                          qt_forcing='variance', variability_nudge_constant_T=True) == []
    for i, les in enumerate(les_a):
        assert numpy.array_equal(numpy.asarray(ens.ql_ref[i]), numpy.asarray(les.ql_ref))
        assert numpy.array_equal(ens.fields3d["QT"][i], les.fields.QT) and not numpy.array_equal(les.fields.QT, fs[i]["qt"])
        assert numpy.array_equal(ens.fields3d["THL"][i], les.fields.THL)
    # an ensemble without 3-D field access is refused up front, not with an AttributeError deep inside
    ens2 = models.SyntheticLESEnsemble.from_models(les_a)
    del_fields = type("NoFields", (), {})()
    for name in ("batched", "grid_indices", "zf_cache", "zh_cache", "get_profiles_batched", "set_forcings_batched", "model_time"):
        setattr(del_fields, name, getattr(ens2, name))
    del_fields.__class__.__len__ = lambda self: n
    spcpl.gather_gcm_data(gcm_b, del_fields, False, write=False)
    with pytest.raises(NotImplementedError, match="get_fields_batched"):
        spcpl.set_les_forcings_batched(del_fields, gcm_b, True, True, None, 900.0, 1.0, False, qt_forcing='variance')


def test_set_gcm_tendencies_from_file_replays_the_nearest_record(tmp_path):
    """splib/spcpl.py:558-570: the tendencies a run wrote to spifs, set again on a GCM from the file (float32 as stored),
    the record chosen by the GCM's model time"""
    from sp_coupler_amd import spio
    from sp_coupler_amd.driver import Coupler
    gcm, les_models = models.make_models(3, nG=19, nL=160, seed=2)
    path = str(tmp_path / "spifs.nc")
    spcpl.writer = spio.SpifsWriter(path, [m.grid_index for m in les_models], [0] * 3, [0] * 3, les_models[0].zf_cache, 19)
    rec = Recorder(gcm, les_models)
    Coupler(gcm, les_models, write=True).run(3)
    want = _by_key(rec.log)                                   # what set_profile_tendency received, per step
    # replay from the still-open file: model time 900 s -> the record written with Time = 900 (the second step's)
    gcm2, _ = models.make_models(3, nG=19, nL=160, seed=2)
    gcm2.model_time = 930.0
    for les in les_models:
        spcpl.set_gcm_tendencies_from_file(gcm2, les)
    spcpl.writer.close()
    spcpl.writer = None
    c = spio.read_column(path, 0)
    ti = int(numpy.abs(c["Time"] - 930.0).argmin())
    for les in les_models:
        for var in ("U", "V", "T", "SH", "QL", "QI", "A"):
            got = gcm2.tendencies[var][les.grid_index]
            assert numpy.array_equal(got, numpy.asarray(want[("gcm", les.grid_index, "f_" + var)][ti], dtype=numpy.float32).astype(numpy.float64)), var
    # by path, after the run
    gcm3, _ = models.make_models(3, nG=19, nL=160, seed=2)
    spcpl.set_gcm_tendencies_from_file(gcm3, les_models[1], path=path)          # model time 0: the first record
    assert numpy.array_equal(gcm3.tendencies["T"][les_models[1].grid_index],
                             numpy.asarray(want[("gcm", les_models[1].grid_index, "f_T")][0], dtype=numpy.float32).astype(numpy.float64))
    with pytest.raises(RuntimeError, match="no spifs file"):
        spcpl.set_gcm_tendencies_from_file(gcm3, les_models[1])
