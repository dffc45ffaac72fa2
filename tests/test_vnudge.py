"""variability nudge (splib/spcpl.py:613-744).  PARITY UNPINNED: the reference's tests hold no
fixture for it.  CPU part: the oracle (NumPy + scipy.optimize.brentq, as the reference) does what the algorithm
promises, and the scalar restatements of brentq / ndarray.sum() that the HIP kernel re-implements equal the
library routines bit for bit.  GPU part (-m gpu): the kernel against the oracle."""

import numpy
import pytest
from scipy.optimize import brentq

from oracle import vnudge_oracle as vo


def make_les_fields(itot, jtot, ktot, seed, cloudy=True):
    """3-D qt / qsat / thl / ql of one synthetic LES plus slab means and a GCM ql_ref with levels in every branch of
    the algorithm: clouds to match (brentq), too little variability (no bracket -> additive noise), LES cloudier than
    the GCM (barely-unsaturated branch), clear levels (no nudge)."""
    rng = numpy.random.default_rng(seed)
    z = (numpy.arange(ktot) + 0.5) / ktot
    qsat_prof = 0.016 * numpy.exp(-3.0 * z)
    qt_prof = qsat_prof * (0.75 + 0.3 * numpy.exp(-((z - 0.35) / 0.15) ** 2))
    sig = 0.04 * qt_prof * (0.2 + z)
    qt = qt_prof[None, None, :] + sig[None, None, :] * rng.normal(size=(itot, jtot, ktot))
    qsat = qsat_prof[None, None, :] * (1 + 0.01 * rng.normal(size=(itot, jtot, ktot)))
    ql = numpy.maximum(qt - qsat, 0.0)
    ql_av, qt_av = ql.mean(axis=(0, 1)), qt.mean(axis=(0, 1))
    presf = 1e5 * numpy.exp(-0.5 * z)
    thl = 290.0 + 10 * z[None, None, :] + 0.1 * rng.normal(size=(itot, jtot, ktot))
    ql_ref = ql_av * rng.uniform(0.3, 3.0, ktot)
    ql_ref[::7] = 0.0                                   # GCM says clear
    cl = numpy.nonzero(ql_av > 1e-7)[0]
    ql_ref[cl[1::3]] = 0.0                              # ... also where the LES has cloud: "barely unsaturated" branch
    ql_ref[3::11] = 5e-4                                # far more cloud than the LES variability can produce
    if cloudy:
        ql_ref[5::13] = ql_av[5::13] + 2e-5
    return dict(qt=qt, qsat=qsat, ql=ql, thl=thl, ql_av=ql_av, qt_av=qt_av, presf=presf, ql_ref=ql_ref)


def test_brentq_restatement_equals_scipy_bit_for_bit():
    rng = numpy.random.default_rng(0)
    compared = 0
    for t in range(1500):
        kind = t % 3
        if kind == 0:
            c = rng.uniform(0.1, 4.9)
            f = lambda x, c=c: (x - c) ** 3 + 0.1 * (x - c)                                        # noqa: E731
        elif kind == 1:
            qt = rng.normal(8e-3, 1e-3, size=(16, 16)); qs = rng.normal(8.5e-3, 5e-4, size=(16, 16))
            av, ref = qt.mean(), rng.uniform(0, 2e-4)
            f = lambda x, qt=qt, qs=qs, av=av, ref=ref: float(numpy.maximum(x * (qt - av) + av - qs, 0).sum() / 256 - ref)   # noqa: E731
        else:
            qt = rng.normal(8e-3, 1e-3, size=(8, 8)); qs = rng.normal(9.5e-3, 5e-4, size=(8, 8)); R = rng.normal(size=(8, 8))
            ref = rng.uniform(0, 5e-4)
            f = lambda x, qt=qt, qs=qs, R=R, ref=ref: float(numpy.maximum(qt + x * R * 1e-3 - qs, 0).sum() / 64 - ref)      # noqa: E731
        try:
            r1, info = brentq(f, 0, 5, full_output=True)
        except ValueError:
            with pytest.raises(ValueError):
                vo.brentq_restated(f, 0, 5)
            continue
        r2, calls = vo.brentq_restated(f, 0, 5)
        assert r1 == r2 and calls == info.function_calls, t
        compared += 1
    assert compared > 800


def test_npsum_restatement_equals_ndarray_sum_bit_for_bit():
    rng = numpy.random.default_rng(1)
    for n in list(range(1, 300)) + [511, 512, 1000, 4095, 4096, 4097, 8191, 8192, 8193, 12345, 16384]:
        a = rng.normal(size=n) * 10.0 ** rng.integers(-8, 8, size=n)
        assert float(a.sum()) == vo.npsum_restated(a), n
    a = numpy.maximum(rng.normal(size=(64, 64)) * 1.3 - 0.2, 0)      # the shape and expression class of get_ql_diff
    assert float(a.sum()) == vo.npsum_restated(a)


@pytest.mark.parametrize("constantT", [False, True])
def test_oracle_does_what_the_algorithm_promises(constantT):
    f = make_les_fields(16, 12, 40, seed=3)
    numpy.random.seed(42)
    R = vo.make_R(16, 12)
    assert abs(R.sum()) < 1e-12
    r = vo.variability_nudge(f["qt"], f["qsat"], f["ql_av"], f["qt_av"], f["presf"], f["ql_ref"], R, 900.0, constantT,
                             thl=f["thl"], ql=f["ql"])
    assert r["error"] is None
    st = r["status"]
    assert (st & 1).any() and (st & 2).any() and (st & 4).any() and (st == 0).any()      # every branch exercised
    ql_new = numpy.maximum(r["qt"] - f["qsat"], 0).mean(axis=(0, 1))
    for k in numpy.nonzero(((st & 1) != 0) & ((st & 4) == 0) & ((st & 8) == 0))[0]:                # multiplicative root:
        assert abs(ql_new[k] - f["ql_ref"][k]) < 1e-9 * max(f["ql_ref"][k], 1e-6)                  # mean ql == ql_ref
        assert abs(r["qt"][:, :, k].mean() - f["qt_av"][k]) < 1e-12                                # mean qt preserved
    for k in numpy.nonzero((st & 4) != 0)[0]:                                                      # additive root
        assert abs(ql_new[k] - f["ql_ref"][k]) < 1e-9 * max(f["ql_ref"][k], 1e-6) and r["beta"][k] == 1
    for k in numpy.nonzero(st == 0)[0]:
        assert numpy.array_equal(r["qt"][:, :, k], f["qt"][:, :, k]) and r["beta"][k] == 1
    for k in numpy.nonzero(st == 2)[0]:                                                            # barely unsaturated
        assert ql_new[k] <= f["ql_av"][k] * (1 + 1e-12)        # the plane is pulled toward saturation at its wettest point
    assert numpy.array_equal(r["alpha"], numpy.log(r["beta"]) / 900.0)
    if constantT:
        touched = st != 0
        assert not numpy.array_equal(r["thl"][:, :, touched], f["thl"][:, :, touched])
        assert numpy.array_equal(r["thl"][:, :, ~touched], f["thl"][:, :, ~touched])
    else:
        assert r["thl"] is not None and numpy.array_equal(r["thl"], f["thl"])


class FieldLES:
    """minimal LES face for spcpl.variability_nudge: the getters of splib/spcpl.py:618-636 and a ``fields`` namespace"""

    class _Fields:
        pass

    def __init__(self, f, ql_ref, grid_index=1):
        self.f, self.ql_ref, self.grid_index = f, ql_ref, grid_index
        self.fields = FieldLES._Fields()
        self.itot, self.jtot, self.ktot = f["qt"].shape

    def get_itot(self):
        return self.itot

    def get_jtot(self):
        return self.jtot

    def get_field(self, name):
        return {"Qsat": self.f["qsat"], "QT": self.f["qt"], "THL": self.f["thl"], "QL": self.f["ql"]}[name].copy()

    def get_profile(self, name):
        return {"QL": self.f["ql_av"], "QT": self.f["qt_av"]}[name].copy()

    def get_presf(self):
        return self.f["presf"].copy()


@pytest.mark.gpu
@pytest.mark.parametrize("lds", ["1", "pair", "strided", "global"])
@pytest.mark.parametrize("constantT", [False, True])
def test_kernel_matches_the_numpy_scipy_oracle(constantT, lds, monkeypatch):
    """K6 through the drop-in API against the oracle, every device path -- planes resident in LDS (the default where they
    fit: 512-thread workgroups; "pair": paired 256-thread workgroups as for many LES; "strided": loaded without the
    transposed workspace) and planes streamed from the workspace by one workgroup per level (the default for large planes;
    "global": at every size) -- on synthetic LES: two small planes with odd extents (leaf tails of the pairwise sum; 16
    levels per workgroup), 32 x 32 (8 levels per workgroup), 64 x 32 (4), 64 x 64 x 160, the bundled DALES case (2),
    90 x 90 (1), 92 x 92 (two 8192-element chunks in LDS), 96 x 96, 128 x 128, 200 x 170 (too large for the LDS): beta, a,
    the updated qt and qt_std BIT-exact, status equal; thl (constantT) within 8 ulp."""
    from sp_coupler_amd import spcpl
    spcpl.set_engine(None)
    monkeypatch.setenv("SPC_VN_GLOBAL", "1" if lds == "global" else "0")        # planes streamed from the workspace at EVERY size
    monkeypatch.setenv("SPC_VN_PAIR", "2" if lds == "pair" else "1")            # two 256-thread workgroups per CU
    monkeypatch.setenv("SPC_VN_TRANSPOSE", "0" if lds == "strided" else "1")    # without the transposed workspace
    shapes = [(16, 12, 40, 3), (9, 7, 23, 4), (64, 64, 160, 5), (96, 96, 12, 6), (32, 32, 24, 7), (64, 32, 21, 8), (90, 90, 12, 10),
              (92, 92, 12, 11), (128, 128, 12, 12), (200, 170, 5, 13)]
    # 92 x 92 = 8464 points: the LDS path with TWO chunks of numpy's 8192-element blocking; 96 x 96, 128 x 128 and 200 x 170
    # (4 chunks + a ragged last one) do not fit the LDS: one workgroup per level streams the planes from the transposed
    # workspace (k_vnudge_solve<true>; "global" runs EVERY shape that way); without a workspace ("strided") they are refused
    for group in ([0, 1], [2], [3], [4], [5], [6], [7], [8], [9]):     # one launch per extent (a launch has one plane geometry)
        if lds == "strided" and shapes[group[0]][0] * shapes[group[0]][1] > 9000:
            continue
        fs = [make_les_fields(*shapes[g][:3], seed=shapes[g][3]) for g in group]
        if len(group) == 2:                             # same geometry needed inside one launch: pad the second to the first
            fs[1] = make_les_fields(*shapes[group[0]][:3], seed=shapes[group[1]][3])
        les = [FieldLES(f, f["ql_ref"].copy(), grid_index=i + 1) for i, f in enumerate(fs)]
        numpy.random.seed(42)
        got = spcpl.variability_nudge_batched(les, 900.0, constantT, write=False)
        numpy.random.seed(42)
        for m, f, g in zip(les, fs, got):
            R = vo.make_R(m.itot, m.jtot)
            r = vo.variability_nudge(f["qt"], f["qsat"], f["ql_av"], f["qt_av"], f["presf"], f["ql_ref"], R, 900.0,
                                     constantT, thl=f["thl"], ql=f["ql"])
            assert r["error"] is None
            assert numpy.array_equal(g["status"], r["status"]), (g["status"], r["status"])
            assert m.ktot < 12 or ((r["status"] & 1).any() and (r["status"] & 4).any())     # the data reach both root finders
            assert numpy.array_equal(g["beta"], r["beta"]) and numpy.array_equal(g["a"], r["a"])
            assert numpy.array_equal(m.fields.QT, r["qt"])
            assert numpy.array_equal(g["qt_std"], r["qt_std"]) and numpy.array_equal(g["alpha"], r["alpha"])
            if constantT:
                err = numpy.abs(m.fields.THL - r["thl"]).max()
                assert err <= 8 * 2.220446049250313e-16 * numpy.abs(r["thl"]).max(), err
                assert not numpy.array_equal(m.fields.THL, f["thl"])
            else:
                assert not hasattr(m.fields, "THL")


@pytest.mark.gpu
def test_no_sign_change_for_the_additive_noise_raises_like_scipy():
    """Where scipy's brentq raises ValueError (the reference does not guard the additive search, spcpl.py:713) the
    drop-in raises ValueError too, after the launch."""
    from sp_coupler_amd import spcpl
    spcpl.set_engine(None)
    f = make_les_fields(8, 8, 10, seed=9)
    f["ql_ref"][:] = 0.0
    f["ql_ref"][4] = 50.0                  # no amount of noise a R, a in [0, 5], reaches a mean ql of 50
    les = FieldLES(f, f["ql_ref"].copy())
    numpy.random.seed(1)
    with pytest.raises(ValueError, match="different signs"):
        spcpl.variability_nudge(les, 900.0)
    numpy.random.seed(1)
    r = vo.variability_nudge(f["qt"], f["qsat"], f["ql_av"], f["qt_av"], f["presf"], f["ql_ref"], vo.make_R(8, 8), 900.0)
    assert isinstance(r["error"], ValueError)


@pytest.mark.gpu
def test_variance_forcing_through_set_les_forcings_like_splib_step():
    """qt_forcing='variance' through the drop-in call the reference's step makes (splib/splib.py:320 ->
    spcpl.py:377-382): after the forcings of a step, every LES whose model time is > 0 gets its 3-D qt nudged toward
    the GCM's ql profile interpolated to LES levels (les.ql_ref, K1's output) -- compared with the oracle fed the same
    ql_ref; an LES still at model time 0 is left alone; R is drawn from numpy's global generator in les order."""
    from sp_coupler_amd import models, spcpl
    spcpl.set_engine(None)

    class FieldSyntheticLES(models.SyntheticLES):
        def attach_fields(self, seed):
            f = make_les_fields(8, 8, self.nL, seed)
            self.f3 = f
            self.fields = FieldLES._Fields()

        def get_itot(self):
            return 8

        def get_jtot(self):
            return 8

        def get_field(self, name):
            return {"Qsat": self.f3["qsat"], "QT": self.f3["qt"], "THL": self.f3["thl"], "QL": self.f3["ql"]}[name].copy()

        def get_profile(self, name):
            return {"QL": self.f3["ql_av"], "QT": self.f3["qt_av"]}[name].copy()

    gcm = models.SyntheticGCM(8, 91, seed=4)
    les_models = []
    for i in (1, 2, 3):
        les = FieldSyntheticLES(gcm, i, 160, seed=4)
        les.zf_cache, les.zh_cache = les.get_zf(), les.get_zh()
        les.attach_fields(seed=40 + i)
        les_models.append(les)
    les_models[0].model_time = les_models[1].model_time = 900.0        # the third LES has not been stepped yet
    spcpl.gather_gcm_data(gcm, les_models, False, write=False)
    numpy.random.seed(7)
    reqs = spcpl.set_les_forcings_batched(les_models, gcm, True, True, {}, dt_gcm=900.0, factor=1.0, couple_surface=False,
                                          qt_forcing='variance', write=False, variability_nudge_constant_T=True)
    assert len(reqs) == 3 and set(reqs[0]) == {"U", "V", "THL", "QT", "SP", "QL", "QLp"}
    assert not hasattr(les_models[2].fields, "QT")                                   # model time 0: no nudge
    numpy.random.seed(7)
    for les in les_models[:2]:
        f = les.f3
        r = vo.variability_nudge(f["qt"], f["qsat"], f["ql_av"], f["qt_av"], les.get_presf(), numpy.asarray(les.ql_ref),
                                 vo.make_R(8, 8), 900.0, True, thl=f["thl"], ql=f["ql"])
        assert r["error"] is None
        assert numpy.array_equal(les.fields.QT, r["qt"])
        assert numpy.abs(les.fields.THL - r["thl"]).max() <= 8 * 2.220446049250313e-16 * numpy.abs(r["thl"]).max()
        assert (r["status"] != 0).any()


def _chunked_vs_one_launch(monkeypatch, n_les, chunk, itot, nL):
    """spcpl.variability_nudge_batched on n_les LES, once in one launch and once in ceil(n_les / chunk) launches: same
    random stream, same bits (columns are independent; the reference loops over any number of LES, spcpl.py:377-382)"""
    from sp_coupler_amd import spcpl
    runs = []
    for limit in (32767, chunk):
        monkeypatch.setattr(spcpl, "VN_MAX_COLS", limit)
        les = [FieldLES(make_les_fields(itot, itot, nL, seed=70 + i), make_les_fields(itot, itot, nL, seed=70 + i)["ql_ref"], i + 1)
               for i in range(n_les)]
        numpy.random.seed(3)
        out = spcpl.variability_nudge_batched(les, 900.0, constantT=True, write=False)
        runs.append((out, [m.fields.QT for m in les], [m.fields.THL for m in les]))
    (o1, q1, t1), (o2, q2, t2) = runs
    for i in range(n_les):
        for k in ("beta", "alpha", "qt_std", "a", "status"):
            assert numpy.array_equal(o1[i][k], o2[i][k], equal_nan=True), (i, k)
        assert numpy.array_equal(q1[i], q2[i]) and numpy.array_equal(t1[i], t2[i]), i
    assert any((o["status"] != 0).any() for o in o1)


def test_nudge_in_chunks_gives_the_bits_of_one_launch_host_logic(monkeypatch):
    """CPU, oracle-backed test engine: 5 LES in chunks of 2 (three launches)"""
    from sp_coupler_amd import spcpl
    from tests.fake_engine import OracleEngine
    spcpl.set_engine(OracleEngine())
    try:
        _chunked_vs_one_launch(monkeypatch, 5, 2, 8, 24)
    finally:
        spcpl.set_engine(None)


@pytest.mark.gpu
def test_nudge_in_chunks_gives_the_bits_of_one_launch_on_the_gpu(monkeypatch):
    """K6: 7 LES of 16 x 16 x 40 in chunks of 3 (three launches, the last one short)"""
    from sp_coupler_amd import spcpl
    spcpl.set_engine(None)
    try:
        _chunked_vs_one_launch(monkeypatch, 7, 3, 16, 40)
    finally:
        spcpl.set_engine(None)


def _load_vnudge_golden():
    import os
    z = numpy.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vnudge_small.npz"))   # no pickle
    return z, {k[3:]: z[k] for k in z.files if k.startswith("in_")}


@pytest.mark.parametrize("constantT", [False, True])
def test_oracle_reproduces_the_vnudge_golden(constantT):
    """tests/golden/vnudge_small.npz (oracle-generated, see make_vnudge_golden.py): the oracle, numpy's sum order,
    numpy's global generator and scipy's brentq still give the committed numbers."""
    z, f = _load_vnudge_golden()
    numpy.random.seed(42)
    assert numpy.array_equal(vo.make_R(12, 10), f["R"])
    r = vo.variability_nudge(f["qt"], f["qsat"], f["ql_av"], f["qt_av"], f["presf"], f["ql_ref"], f["R"], 900.0, constantT,
                             thl=f["thl"], ql=f["ql"])
    tag = "cT%d_" % int(constantT)
    for k in ("qt", "thl", "beta", "alpha", "qt_std", "a", "status"):
        assert numpy.array_equal(r[k], z[tag + k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("constantT", [False, True])
def test_kernel_reproduces_the_vnudge_golden(constantT):
    from sp_coupler_amd import spcpl
    spcpl.set_engine(None)
    z, f = _load_vnudge_golden()
    les = FieldLES(f, f["ql_ref"].copy())
    numpy.random.seed(42)
    g = spcpl.variability_nudge(les, 900.0, constantT, write=False)
    tag = "cT%d_" % int(constantT)
    for k in ("beta", "alpha", "qt_std", "a", "status"):
        assert numpy.array_equal(g[k], z[tag + k]), k
    assert numpy.array_equal(les.fields.QT, z[tag + "qt"])
    if constantT:
        assert numpy.abs(les.fields.THL - z[tag + "thl"]).max() <= 8 * 2.220446049250313e-16 * numpy.abs(z[tag + "thl"]).max()
