"""-m gpu: the reference-named host API (sp_coupler_amd.spcpl) and the per-step driver, closed loop,
against the oracle-driven reference sequencing (tests/ref_driver.py) on identical synthetic models."""
import numpy
import pytest

from sp_coupler_amd import models
from tests.gpu_util import EPS
from tests.ref_driver import RefCoupler

pytestmark = pytest.mark.gpu


class Recorder:
    """wraps model setters to log what the product pushes to the models"""

    def __init__(self, gcm, les_models):
        self.log = []
        for les in les_models:
            for name, key in (("set_tendency_U", "f_u"), ("set_tendency_V", "f_v"), ("set_tendency_THL", "f_thl"),
                              ("set_tendency_QT", "f_qt"), ("set_tendency_surface_pressure", "f_ps"),
                              ("set_tendency_QL", "f_ql"), ("set_ref_profile_QL", "ql_ref"), ("set_z0m_surf", "z0m"),
                              ("set_z0h_surf", "z0h"), ("set_wt_surf", "wthl"), ("set_wq_surf", "wqt")):
                self._wrap_les(les, name, key)
            self._wrap_cf(les)
        orig = gcm.set_profile_tendency

        def rec(var, gi, values, _o=orig):
            self.log.append(("gcm", int(gi), "f_" + var, numpy.array(values)))
            return _o(var, gi, values)
        gcm.set_profile_tendency = rec

    def _wrap_les(self, les, name, key):
        orig = getattr(les, name)

        def rec(v, return_request=False, _o=orig):
            self.log.append(("les", les.grid_index, key, numpy.array(v)))
            return _o(v, return_request=return_request)
        setattr(les, name, rec)

    def _wrap_cf(self, les):
        orig = les.get_cloudfraction

        def rec(indices, return_request=False, _o=orig):
            self.log.append(("idx", les.grid_index, "idx", numpy.array(indices)))
            return _o(indices, return_request=return_request)
        les.get_cloudfraction = rec


def _by_key(log):
    out = {}
    for kind, col, key, arr in log:
        out.setdefault((kind, col, key), []).append(arr)
    return out


@pytest.mark.parametrize("cplsurf", [False, True])
def test_closed_loop_matches_reference_sequencing(cplsurf):
    from sp_coupler_amd import spcpl
    from sp_coupler_amd.driver import Coupler
    nsteps, n_les = 3, 11
    gcm_a, les_a = models.make_models(n_les, nG=91, nL=160, seed=21)
    gcm_b, les_b = models.make_models(n_les, nG=91, nL=160, seed=21)
    rec = Recorder(gcm_a, les_a)
    cpl = Coupler(gcm_a, les_a, cplsurf=cplsurf, les_forcing_factor=0.9, gcm_forcing_factor=1.1)
    ref = RefCoupler(gcm_b, les_b, cplsurf=cplsurf, les_forcing_factor=0.9, gcm_forcing_factor=1.1)
    for _ in range(nsteps):
        cpl.step()
        ref.step()
    got, want = _by_key(rec.log), _by_key(ref.log)
    assert set(got) == set(want)
    # thl passes through the device pow (a few ulp of ~300 K); the closed loop feeds it back into the LES
    # state, so later steps inherit ~1e-13 relative differences.  Indices stay bit-exact throughout.
    for key in want:
        assert len(got[key]) == len(want[key]) == nsteps, key
        for s in range(nsteps):
            g, w = got[key][s], want[key][s]
            if key[2] == "idx":
                assert numpy.array_equal(g, w), (key, s)
                continue
            scale = {"f_thl": 300.0 / 900.0, "f_T": 300.0 / 900.0, "wthl": 1.0}.get(key[2], None)
            if s == 0 and scale is None and key[2] not in ("f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A"):
                assert numpy.array_equal(g, w), (key, s)                 # no pow upstream on the first step
            else:
                tol = 64 * EPS * (scale if scale is not None else max(numpy.abs(w).max(), 1e-300) * 1e3)
                assert numpy.abs(g - w).max() <= tol, (key, s, numpy.abs(g - w).max(), tol)
    for var in ("U", "V", "T", "SH", "QL", "QI", "A"):
        a, b = gcm_a.state[var], gcm_b.state[var]
        assert numpy.abs(a - b).max() <= 1e-11 * max(numpy.abs(b).max(), 1e-30), var
    assert len(cpl.timing_rows) == nsteps and not cpl.firststep
    assert gcm_a.calls[:4] == ["until_cloud_scheme", "cloud_scheme", "from_cloud_scheme", "until_cloud_scheme"]
    assert spcpl.current_batch().n == n_les


def test_per_les_api_drop_in_with_unchanged_reference_loop():
    """The reference's own loop shape: per-les calls with `profile=profiles[les]` (splib.py:317-332)."""
    from tests.ref_driver import check_reference_loop_shape
    check_reference_loop_shape()


def test_gather_quirk_and_output_columns():
    from sp_coupler_amd import spcpl
    gcm, les_models = models.make_models(3, nG=91, nL=160, seed=9)
    assert spcpl.gather_gcm_data(gcm, [], False) is None                     # no columns at all
    les0 = models.SyntheticLES(gcm, 0, 160)
    les0.zf_cache, les0.zh_cache = les0.get_zf(), les0.get_zh()
    assert spcpl.gather_gcm_data(gcm, [les0], False) is None                 # cols == [0]: any() quirk, spcpl.py:63
    C = {"T": gcm.state["T"][:2], "SH": gcm.state["SH"][:2], "QL": gcm.state["QL"][:2], "QI": gcm.state["QI"][:2],
         "Pf": gcm.state["Pfull"][:2], "Ph": gcm.state["Phalf"][:2], "Zgfull": gcm.state["Zgfull"][:2],
         "Zghalf": gcm.state["Zghalf"][:2]}
    D = spcpl.output_column_conversion(dict(C))
    from oracle import spcpl_oracle as orc
    one = {k: v[1].copy() for k, v in C.items()}
    orc.output_column_conversion(one)
    for k in ("Tv", "Zh", "Zf", "Psurf", "Ph", "QT"):
        assert numpy.array_equal(D[k][1], one[k]), k
    assert numpy.abs(D["THL"][1] - one["THL"]).max() <= 8 * EPS * numpy.abs(one["THL"]).max()


def test_conservative_coarsening_through_the_driver():
    from sp_coupler_amd.driver import Coupler
    gcm_a, les_a = models.make_models(6, nG=91, nL=160, seed=33)
    gcm_b, les_b = models.make_models(6, nG=91, nL=160, seed=33)
    cpl = Coupler(gcm_a, les_a, conservative_coarsening=True)
    ref = RefCoupler(gcm_b, les_b, conservative=True)
    for _ in range(2):
        cpl.step()
        ref.step()
    for var in ("U", "V", "T", "SH", "QL", "QI", "A"):
        a, b = gcm_a.state[var], gcm_b.state[var]
        assert numpy.abs(a - b).max() <= 1e-11 * max(numpy.abs(b).max(), 1e-30), var


def test_spifs_output_through_the_driver(tmp_path):
    from sp_coupler_amd import spcpl, spio
    from sp_coupler_amd.driver import Coupler
    gcm, les_models = models.make_models(4, nG=91, nL=160, seed=8)
    rec = Recorder(gcm, les_models)
    path = str(tmp_path / "spifs.nc")
    spcpl.writer = spio.SpifsWriter(path, [m.grid_index for m in les_models], [m.lat for m in les_models],
                                    [m.lon for m in les_models], les_models[0].zf_cache, 91)
    try:
        cpl = Coupler(gcm, les_models, cplsurf=True, write=True)
        cpl.run(2)
        spcpl.writer.close()
    finally:
        spcpl.writer = None
    got = _by_key(rec.log)
    for ci, les in enumerate(les_models):
        c = spio.read_column(path, ci)
        assert c["Time"].tolist() == [0.0, 1800.0] and c["grid_index"] == les.grid_index   # record 0: splib.py:193
        for s in range(2):
            for name in ("f_u", "f_thl", "f_qt", "f_T", "f_SH", "f_A", "wthl"):
                want = got[("les" if name[2].islower() or name == "wthl" else "gcm", les.grid_index, name)][s]
                assert numpy.array_equal(c[name][s], numpy.asarray(want, dtype=numpy.float32)), (name, s)
            assert numpy.isfinite(c["Tv"][s]).all() and numpy.isfinite(c["t"][s]).all() and (c["Zf"][s][:-1] > c["Zf"][s][1:]).all()
            # ql_water is float32(ql - ql_ice) of the float64 slab means (spcpl.py:402); the file holds float32(ql) and
            # float32(ql_ice), so recomputing it from them may differ by the rounding of the two inputs
            diff = numpy.abs(c["ql_water"][s].astype(numpy.float64) - (c["ql"][s].astype(numpy.float64) - c["ql_ice"][s]))
            assert diff.max() <= 2 * numpy.finfo(numpy.float32).eps * max(float(numpy.abs(c["ql"][s]).max()), 1e-30)


def test_surface_fluxes_dict_form_and_set_les_state():
    """convert_surface_fluxes on a dict of GCM data (extra output columns, spcpl.py:112-115) and the
    init-state path convert_profiles -> set_les_state (splib.py:202-204)."""
    from oracle import spcpl_oracle as orc
    from sp_coupler_amd import spcpl
    gcm, les_models = models.make_models(3, nG=91, nL=160, seed=17)
    st = gcm.state
    C = {"Ph": st["Phalf"][4:7], "T": st["T"][4:7]}
    C.update({k: st[k][4:7] for k in ("Z0M", "Z0H", "QLflux", "QIflux", "SHflux", "TLflux", "TSflux")})
    z0m, z0h, wthl, wqt = spcpl.convert_surface_fluxes(C)
    for i in range(3):
        col = {"Phalf": C["Ph"][i], "T": C["T"][i]}
        col.update({k: C[k][i] for k in ("Z0M", "Z0H", "QLflux", "QIflux", "SHflux", "TSflux")})
        r = orc.convert_surface_fluxes(col)
        assert wqt[i] == r[3] and abs(wthl[i] - r[2]) <= 8 * EPS * abs(r[2]) and z0m[i] == r[0]
    one = {k: v[1] for k, v in C.items()}
    assert spcpl.convert_surface_fluxes(one)[3] == wqt[1]
    spcpl.gather_gcm_data(gcm, les_models, True)
    numpy.random.seed(42)                                                     # splib.py:181
    u, v, thl, qt, ps, ql = spcpl.convert_profiles(les_models[0])
    spcpl.set_les_state(les_models[0], u, v, thl, qt, ps)
    assert abs(les_models[0].p["U"] - u).max() < 0.5 and les_models[0].p["PS"] == ps
    numpy.random.seed(42)
    noise = 0.5 * numpy.random.uniform(-1., 1., (8, 8, 160))
    assert numpy.allclose(les_models[0].p["U"], (noise + u).mean(axis=(0, 1)), rtol=0, atol=1e-12)


def test_get_cloudfraction_like_the_reference_test():
    """The reference's own spcpl test (splib/test/spcpl_test.py:10-16), through the drop-in API: a dummy LES with
    spdummy's grid (k=20, dz=200: zf=k*dz, zh=(k+0.5)*dz, spdummy.py:185-222) and cloud-fraction profile
    A=0.5+0.2cos(6 zf/(dz k)) clipped to 0..k-1 (spdummy.py:261-262,319-321); gcm_Zh=[1e5,1e3,100,10,1,0]."""
    from sp_coupler_amd import spcpl
    from tests.test_parity_gpu import _ulp_search
    k, dz, grav = 20, 200.0, 9.81

    class DummyLes:
        grid_index = 1
        zf_cache = numpy.arange(k) * dz
        zh_cache = (numpy.arange(k) + 0.5) * dz

        def get_cloudfraction(self, i):
            indices = numpy.clip(i, 0, k - 1)
            return (0.5 + 0.2 * numpy.cos(6. * self.zf_cache / (dz * k)))[indices]

    class DummyGcm:
        def get_profile_fields(self, var, cols):
            Zh = numpy.array([100000., 1000., 100., 10., 1., 0.])
            if var == "Zghalf":
                return numpy.array([[_ulp_search(z, grav) if z else 0.0 for z in Zh]])
            if var == "Zgfull":
                return numpy.array([0.5 * (Zh[:-1] + Zh[1:]) * grav])
            if var == "Phalf":
                return numpy.array([[0.0, 2e4, 5e4, 8e4, 9.5e4, 1e5]])
            if var in ("Pfull", "T"):
                return numpy.array([[1e4, 3.5e4, 6.5e4, 8.7e4, 9.7e4]]) if var == "Pfull" else numpy.full((1, 5), 280.0)
            return numpy.zeros((1, 5))

    les = DummyLes()
    spcpl.gather_gcm_data(DummyGcm(), [les], False, write=False)
    A = spcpl.get_cloud_fraction(les)
    assert numpy.array_equal(les.gcm_Zh, [100000., 1000., 100., 10., 1., 0.])
    assert spcpl.cloud_fraction_indices(les).tolist() == [0, 0, 1, 5, 20]
    tolerance = 1.e-10                                                         # spcpl_test.py:7
    assert abs(A[0] - (0.5 + 0.2 * numpy.cos(6. * (1. - k) / k))) < tolerance    # spcpl_test.py:15
    assert abs(A[-1] - (0.5 + 0.2)) < tolerance                                # spcpl_test.py:16


@pytest.mark.parametrize("cplsurf,conservative", [(False, False), (True, False), (False, True)])
def test_batched_model_protocol_on_the_gpu_equals_the_per_les_path(cplsurf, conservative):
    """The fast drop-in transport (LES ensemble + batched GCM calls: one pinned upload, one launch, one download per
    direction) against the reference-style per-LES calls on identical models: the SAME kernels run either way, so
    the model states must stay bit-identical over a closed loop of several steps."""
    from sp_coupler_amd import spcpl
    from sp_coupler_amd.driver import Coupler
    spcpl.set_engine(None)
    n_les, nsteps = 37, 3
    gcm_a, les_a = models.make_models(n_les, nG=91, nL=160, seed=23)
    gcm_b, les_b = models.make_models(n_les, nG=91, nL=160, seed=23)
    gcm_b.__class__ = models.BatchedSyntheticGCM
    ens = models.SyntheticLESEnsemble.from_models(les_b)
    Coupler(gcm_a, les_a, cplsurf=cplsurf, les_forcing_factor=0.9, gcm_forcing_factor=1.1,
            conservative_coarsening=conservative).run(nsteps)
    Coupler(gcm_b, ens, cplsurf=cplsurf, les_forcing_factor=0.9, gcm_forcing_factor=1.1,
            conservative_coarsening=conservative).run(nsteps)
    for var in gcm_a.state:
        assert numpy.array_equal(gcm_a.state[var], gcm_b.state[var]), var
    for i, m in enumerate(les_a):
        for k in ("U", "V", "THL", "QT", "QL", "T", "PS", "Rain"):
            assert numpy.array_equal(numpy.asarray(m.p[k]), ens.p[k][i]), (k, i)
    assert not numpy.array_equal(gcm_a.state["T"], models.make_models(n_les, nG=91, nL=160, seed=23)[0].state["T"])
