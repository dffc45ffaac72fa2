"""K7: the helpers of splib/sputils.py as standalone GPU operators (sp_coupler_amd/sputils.py over spc_exner_* /
spc_interp_* / spc_searchsorted_* / spc_interp_c_* / spc_rms_* of the C ABI).

* the reference's own test file for these helpers (splib/test/sputils_test.py:10-47) restated against the GPU module:
  same numbers, same assertions, same tolerance (1e-10) -- these pin exner / iexner / rms;
* bit parity with NumPy (numpy.interp / numpy.searchsorted are what the reference calls) and with the oracle's
  restatement of integral / interp_c / interp_rho (parity unpinned by the reference: it holds no fixture for them)."""
import json
import os

import numpy
import pytest
import torch

from oracle import spcpl_oracle as orc
from sp_coupler_amd import synthetic
from tests.gpu_util import EPS, assert_bits

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


@pytest.fixture(scope="module")
def sputils():
    from sp_coupler_amd import spcpl, sputils as su
    from sp_coupler_amd.engine import Engine
    spcpl.set_engine(Engine("cuda:0"))
    yield su
    spcpl.set_engine(None)


class TestReferenceSputilsTest:
    """splib/test/sputils_test.py, every test of the class, through the GPU module"""
    tolerance = 1.e-10

    def test_rms(self, sputils):
        a, b, c = 3.204, -1.2092, 9.6231
        assert abs(sputils.rms(numpy.array([a, b, c])) - numpy.sqrt((a * a + b * b + c * c) / 3)) < self.tolerance

    def test_rms_repeat(self, sputils):
        a, n = 3.204, 23
        assert abs(sputils.rms(numpy.array([a for i in range(n)])) - a) < self.tolerance

    def test_exner(self, sputils):
        a = 2.03947
        p = a * sputils.pref0
        assert abs(numpy.log(sputils.exner(p)) - numpy.log(a) * sputils.rd / sputils.cp) < self.tolerance

    def test_exner_unity(self, sputils):
        assert abs(sputils.exner(sputils.pref0) - 1) < self.tolerance
        assert numpy.ndim(sputils.exner(sputils.pref0)) == 0          # a scalar in, a scalar out

    def test_iexner(self, sputils):
        p = 12.03947 * sputils.pref0
        assert abs(sputils.exner(p) * sputils.iexner(p) - 1) < self.tolerance

    def test_get_closest_points(self, sputils):
        points = [(52.314970, 4.824198), (52.379932, 4.897997), (52.387264, 5.082968), (52.278097, 5.021635)]
        assert sputils.find_closest_points(points, (52.356591, 4.954541))[0] == 1


def test_known_answers_of_the_reference(sputils):
    g = GOLD
    assert abs(sputils.exner(g["exner"]["a"] * g["exner"]["pref0"]) - g["exner"]["expected"]) < g["tolerance"]
    assert abs(sputils.exner(g["exner_unity"]["p"]) - 1.0) < g["tolerance"]
    assert abs(sputils.rms(numpy.array(g["rms"]["numbers"])) - g["rms"]["expected"]) < g["tolerance"]
    cf = g["cloud_fraction"]                   # spcpl.py:26: searchsorted(zh, Zh, side="right")[:-1][::-1] == [0, 0, 1, 5, 20]
    zh, Zh = numpy.array(cf["zh"], dtype=float), numpy.array(cf["gcm_Zh"], dtype=float)
    idx = sputils.searchsorted(zh, Zh, side="right")
    assert idx.dtype == numpy.int64 and idx[:-1][::-1].tolist() == cf["indices"]


def test_exner_accuracy_and_shapes(sputils):
    rng = numpy.random.default_rng(11)
    p = rng.uniform(1.0, 1.1e5, size=(37, 91))
    for fn, ref in ((sputils.exner, orc.exner), (sputils.iexner, orc.iexner)):
        got, want = fn(p), ref(p)
        assert got.shape == want.shape and got.dtype == numpy.float64
        assert (numpy.abs(got - want) <= 2 * EPS * numpy.abs(want)).all()          # own pow: <= 2 ulp of numpy.power
    dev = torch.from_numpy(p).cuda()
    r = sputils.iexner(dev)
    assert isinstance(r, torch.Tensor) and r.is_cuda and numpy.array_equal(r.cpu().numpy(), sputils.iexner(p))


def test_exner_quotient_window_and_special_values(sputils):
    """the operator forms p / pref0 by Markstein's iteration for p in [2^-900, 2^900] and by a division outside; both sides of
    both edges, subnormal, huge, zero, negative, infinite and NaN pressures against numpy's (p / pref0) ** k"""
    p = numpy.array([2.0 ** -900, numpy.nextafter(2.0 ** -900, 0), 2.0 ** -901, 2.0 ** 900, numpy.nextafter(2.0 ** 900, numpy.inf),
                     2.0 ** 901, 5e-324, 2.3e-308, 1.7e308, 0.0, -0.0, -1.0, -1e5, numpy.inf, -numpy.inf, numpy.nan, 1e5, 101325.0, 1.0])
    for fn, k in ((sputils.exner, sputils.rd / sputils.cp), (sputils.iexner, -sputils.rd / sputils.cp)):
        with numpy.errstate(all="ignore"):
            want = (p / sputils.pref0) ** k
        got = fn(p)
        assert numpy.array_equal(numpy.isnan(got), numpy.isnan(want)), (got, want)
        fin = numpy.isfinite(want)
        assert numpy.array_equal(got[~fin & ~numpy.isnan(want)], want[~fin & ~numpy.isnan(want)])          # +-inf where numpy has them
        assert (numpy.abs(got[fin] - want[fin]) <= 2 * EPS * numpy.abs(want[fin])).all(), (got, want)
    # (the quotient bit for bit: K1's thl, which divides, against this operator on 3e7 pressures in tests/test_cross_kernel_gpu.py)


def _interp_rows(x, xp, fp):
    n = max(a.shape[0] if a.ndim == 2 else 1 for a in (x, xp, fp))
    row = lambda a, r: a[r] if a.ndim == 2 else a      # noqa: E731
    with numpy.errstate(all="ignore"):
        return numpy.stack([numpy.interp(row(x, r), row(xp, r), row(fp, r)) for r in range(n)])


def test_interp_is_numpy_interp_bit_for_bit(sputils):
    rng = numpy.random.default_rng(5)
    n, n_xp, n_x = 300, 91, 160
    xp = numpy.sort(rng.uniform(0, 3e4, size=(n, n_xp)), axis=1)
    fp = rng.normal(size=(n, n_xp)) * 10
    x = rng.uniform(-500, 3.1e4, size=(n, n_x))
    x[:, 3] = xp[:, 7]                       # exact hits
    x[:, 4] = xp[:, 0]; x[:, 5] = xp[:, -1]
    x[5, 9] = numpy.nan; x[6, 10] = numpy.inf; x[7, 11] = -numpy.inf
    fp[9, 20] = numpy.nan; fp[10, 30] = numpy.inf
    xp[11, 40] = xp[11, 41]                  # a repeated sample point: zero dx
    assert_bits("interp per-row", sputils.interp(x, xp, fp), _interp_rows(x, xp, fp))
    assert_bits("interp shared xp", sputils.interp(x, xp[0], fp), _interp_rows(x, xp[0], fp))
    assert_bits("interp shared x", sputils.interp(x[0], xp, fp), _interp_rows(x[0], xp, fp))
    # one column, the way the reference calls it (spcpl.py:224: reversed views of the GCM arrays)
    Zf, thl = xp[3][::-1], fp[3][::-1]
    assert_bits("interp 1-D reversed views", sputils.interp(x[3], Zf[::-1], thl[::-1]), numpy.interp(x[3], xp[3], fp[3]))
    assert sputils.interp(1234.5, xp[3], fp[3]) == numpy.interp(1234.5, xp[3], fp[3])        # scalar x
    # a single sample point, and sample arrays too long for the LDS (read from global memory)
    assert_bits("interp n_xp=1", sputils.interp(x[:4], xp[:4, :1], fp[:4, :1]), _interp_rows(x[:4], xp[:4, :1], fp[:4, :1]))
    big = numpy.sort(rng.uniform(0, 1, size=(3, 9000)), axis=1)
    bf, bx = rng.normal(size=(3, 9000)), rng.uniform(-0.1, 1.1, size=(3, 50))
    assert_bits("interp n_xp=9000", sputils.interp(bx, big, bf), _interp_rows(bx, big, bf))
    # device tensors in, device tensor out
    r = sputils.interp(torch.from_numpy(x).cuda(), torch.from_numpy(xp).cuda(), torch.from_numpy(fp).cuda())
    assert r.is_cuda
    assert_bits("interp device", r.cpu().numpy(), _interp_rows(x, xp, fp))


def test_interp_errors_like_numpy(sputils):
    with pytest.raises(ValueError):
        sputils.interp(numpy.zeros(3), numpy.zeros(4), numpy.zeros(5))       # fp and xp are not of the same length
    with pytest.raises(ValueError):
        sputils.interp(numpy.zeros(3), numpy.zeros(0), numpy.zeros(0))       # array of sample points is empty
    with pytest.raises(ValueError):
        sputils.interp(numpy.zeros(3), numpy.arange(4.), numpy.arange(4.), period=0)      # period must be a non-zero value
    # numpy.interp(..., period=): numpy's host-side normalisation + the same kernel = numpy's bits
    rng = numpy.random.default_rng(33)
    ang, val, q = rng.uniform(-400, 800, size=(40, 91)), rng.normal(size=(40, 91)), rng.uniform(-1000, 1000, size=(40, 160))
    want = numpy.stack([numpy.interp(q[r], ang[r], val[r], period=360.0) for r in range(40)])
    assert_bits("interp period", sputils.interp(q, ang, val, period=360.0), want)
    assert_bits("interp period, one column", sputils.interp(q[3], ang[3], val[3], period=360.0), want[3])


def test_row_pitches_of_2_pow_24_elements_are_refused():
    """the staged kernels form row offsets with 24-bit multiplies: a pitch of 2^24 elements or more is SPC_ERR_UNSUPPORTED, one
    element less is served (and right)"""
    from sp_coupler_amd import _abi, spcpl
    eng = spcpl.get_engine()
    wide = torch.zeros((2, 2 ** 24), dtype=torch.float64, device="cuda")
    xp = torch.tensor([0.0, 1.0, 2.0], dtype=torch.float64, device="cuda")
    wide[:, :3] = torch.tensor([[10.0, 20.0, 30.0], [1.0, 2.0, 3.0]], dtype=torch.float64, device="cuda")
    x = torch.tensor([0.5, 1.5], dtype=torch.float64, device="cuda")
    with pytest.raises(_abi.SpcError) as e:
        eng.interp(x, xp, wide[:, :3])
    assert e.value.code == _abi.SPC_ERR_UNSUPPORTED and "2^24" in str(e.value)
    near = wide.view(-1)[:2 * (2 ** 24 - 1)].view(2, 2 ** 24 - 1)            # rows 2^24 - 1 elements apart
    near[:, :3] = torch.tensor([[10.0, 20.0, 30.0], [1.0, 2.0, 3.0]], dtype=torch.float64, device="cuda")
    assert numpy.array_equal(eng.interp(x, xp, near[:, :3]).cpu().numpy(), [[15.0, 25.0], [1.5, 2.5]])
    # a ONE-row launch never multiplies the pitch (its row index is 0): one long vector of 2^24 points and more is served
    # (round-4 advisor: sputils.interp / searchsorted on a single long vector were refused since round 4's second K7 pass)
    long_x = torch.linspace(-1.0, 3.0, 2 ** 24 + 5, dtype=torch.float64, device="cuda")
    fp1 = torch.tensor([10.0, 20.0, 30.0], dtype=torch.float64, device="cuda")
    got = eng.interp(long_x, xp, fp1).cpu().numpy()
    assert numpy.array_equal(got, numpy.interp(long_x.cpu().numpy(), [0.0, 1.0, 2.0], [10.0, 20.0, 30.0]))
    idx = eng.searchsorted(xp, long_x, side="right").cpu().numpy()
    assert numpy.array_equal(idx, numpy.searchsorted([0.0, 1.0, 2.0], long_x.cpu().numpy(), side="right"))
    # rows padded to millions of elements: the slab height is cut so that 32-bit byte offsets hold (rb * pitch * 8 < 2^32)
    pad = wide.view(-1)[:4 * (2 ** 23)].view(4, 2 ** 23)
    pad[:, :3] = torch.tensor([[10.0, 20.0, 30.0], [1.0, 2.0, 3.0], [5.0, 6.0, 7.0], [-1.0, -2.0, -3.0]], dtype=torch.float64, device="cuda")
    x4 = torch.tensor([[0.5], [1.5], [0.25], [2.0]], dtype=torch.float64, device="cuda")
    assert numpy.array_equal(eng.interp(x4, xp, pad[:, :3]).cpu().numpy(), [[15.0], [2.5], [5.25], [-3.0]])


def test_searchsorted_is_numpy_searchsorted(sputils):
    rng = numpy.random.default_rng(6)
    n, n_a, n_v = 200, 160, 92
    a = numpy.sort(rng.uniform(0, 4000, size=(n, n_a)), axis=1)
    v = rng.uniform(-100, 4100, size=(n, n_v))
    v[:, 2] = a[:, 17]; v[:, 3] = a[:, 0]; v[:, 4] = a[:, -1]; v[3, 5] = numpy.nan
    a[4, 50] = a[4, 51]
    for side in ("left", "right"):
        want = numpy.stack([numpy.searchsorted(a[r], v[r], side=side) for r in range(n)])
        got = sputils.searchsorted(a, v, side=side)
        assert got.dtype == numpy.int64 and numpy.array_equal(got, want), side
        assert numpy.array_equal(sputils.searchsorted(a[0], v, side=side),
                                 numpy.stack([numpy.searchsorted(a[0], v[r], side=side) for r in range(n)]))
        assert numpy.array_equal(sputils.searchsorted(a[7], v[7], side=side), numpy.searchsorted(a[7], v[7], side=side))
    assert sputils.searchsorted(a[7], 1234.5) == numpy.searchsorted(a[7], 1234.5)
    assert numpy.array_equal(sputils.searchsorted(numpy.zeros(0), v[0]), numpy.searchsorted(numpy.zeros(0), v[0]))


def _coarse_inputs(n, nG, nL, seed, per_column_grid=False):
    gcm, zf, zh, prof = synthetic.make_batch(n, nG, nL, seed=seed, couple_surface=False, per_column_grid=per_column_grid)
    Zh = numpy.stack([orc.convert_profiles({k: v[c] for k, v in gcm.items()}, zf[c] if zf.ndim == 2 else zf)["Zh"] for c in range(n)])
    return Zh, zh, prof["QT"], prof["Rhobf"]


@pytest.mark.parametrize("nG,nL,per_col", [(91, 160, False), (19, 160, True), (31, 2000, False)])
def test_interp_c_and_interp_rho_match_the_oracle(sputils, nG, nL, per_col):
    n = 40
    Zh, zh, q, rho = _coarse_inputs(n, nG, nL, seed=77, per_column_grid=per_col)
    Zh[5, nG // 2] = -3.0                                   # an end point below zh[0]: integral() returns None -> NaN
    z = lambda c: zh[c] if zh.ndim == 2 else zh             # noqa: E731
    with numpy.errstate(all="ignore"):
        want_c = numpy.stack([orc.interp_c(Zh[c], z(c), q[c], rho[c]) for c in range(n)])
    got_c = sputils.interp_c(Zh, zh, q, rho)
    assert_bits("interp_c", got_c, want_c)
    assert numpy.isnan(got_c[5]).any() and (got_c != 0).any()
    assert_bits("interp_c one column", sputils.interp_c(Zh[2], z(2), q[2], rho[2]), want_c[2])
    Zr = numpy.delete(Zh, 5, axis=0)                         # interp_rho divides None by a number in the reference
    zr = numpy.delete(zh, 5, axis=0) if zh.ndim == 2 else zh
    rr = numpy.delete(rho, 5, axis=0)
    want_r = numpy.stack([orc.interp_rho(Zr[c], zr[c] if zr.ndim == 2 else zr, rr[c]) for c in range(n - 1)])
    assert_bits("interp_rho", sputils.interp_rho(Zr, zr, rr), want_r)


def test_integral_like_the_reference(sputils):
    rng = numpy.random.default_rng(8)
    z = numpy.cumsum(rng.uniform(5, 40, size=300))
    q, w = rng.normal(size=299), rng.uniform(0.5, 1.3, size=299)
    for a, b in ((z[3] + 1.0, z[200] - 2.0), (z[250], z[10] + 0.5), (z[0], z[-1]), (z[7], z[7]), (z[4] + 1, z[4] + 2)):
        for ww in (None, w):
            got, want = sputils.integral(a, b, z, q, ww), orc.integral(a, b, z, q, ww)
            assert got == want or (got != got and want != want), (a, b, ww is None, got, want)
    assert sputils.integral(z[0] - 1.0, z[5], z, q) is None                    # sputils.py:113-115
    assert sputils.integral(z[3], z[-1] + 1.0, z, q, w) is None
    # many intervals at once (beyond the reference): rows outside the grid give NaN
    a = rng.uniform(z[0], z[-1], size=64); b = rng.uniform(z[0], z[-1], size=64); a[9] = z[0] - 5
    got = sputils.integral(a, b, z, q, w)
    want = numpy.array([numpy.nan if i == 9 else orc.integral(a[i], b[i], z, q, w) for i in range(64)])
    assert_bits("integral batch", got, want)


def test_rms_rows_in_numpy_order(sputils):
    rng = numpy.random.default_rng(9)
    a = rng.normal(size=(33, 160)) * rng.uniform(1e-3, 1e3, size=(33, 1))
    assert_bits("rms rows", sputils.rms(a, axis=-1), numpy.array([orc.rms(a[r]) for r in range(33)]))
    long = rng.normal(size=20011)                                              # > 8192: numpy's chunked pairwise order
    assert sputils.rms(long) == orc.rms(long)
    assert sputils.rms(a) == numpy.sqrt(numpy.mean(a ** 2))                    # the reference's semantics: ONE number
    cube = rng.normal(size=(16, 16, 40))                                       # e.g. a 3-D LES field
    assert sputils.rms(cube) == numpy.sqrt(numpy.mean(cube ** 2))


def test_operator_plans_and_caller_provided_outputs():
    """round-3 verdict, item 3: every K7 operator as a plan (arguments frozen, one foreign call per launch) writing into a
    caller-provided ``out=`` -- contiguous, pitched, reused across launches -- with the bits of the convenience call"""
    from sp_coupler_amd.engine import Engine
    eng = Engine("cuda:0")
    rng = numpy.random.default_rng(12)
    n, nG, nL = 300, 91, 160
    dev = lambda a: torch.from_numpy(numpy.ascontiguousarray(a)).cuda()       # noqa: E731
    xp = numpy.sort(rng.uniform(0, 4000, size=(n, nG)), axis=1)
    fp, x = rng.normal(size=(n, nG)), rng.uniform(-100, 4100, size=(n, nL))
    zh = numpy.arange(nL) * 25.0
    Zh = numpy.sort(rng.uniform(0, 6000, size=(n, nG + 1)), axis=1)[:, ::-1].copy()
    q, rho = rng.normal(size=(n, nL)), rng.uniform(0.5, 1.3, size=(n, nL))
    big = torch.full((n, 256), -1.0, dtype=torch.float64, device="cuda")       # pitched destination: rows 256 apart
    cases = [
        (lambda out: eng.plan_interp(dev(x), dev(xp), dev(fp), out=out), lambda: eng.interp(dev(x), dev(xp), dev(fp)), big[:, :nL], torch.float64),
        (lambda out: eng.plan_searchsorted(dev(xp), dev(x), side="right", out=out), lambda: eng.searchsorted(dev(xp), dev(x), side="right"),
         torch.full((n, 200), -1, dtype=torch.int64, device="cuda")[:, :nL], torch.int64),
        (lambda out: eng.plan_interp_c(dev(Zh), dev(zh), dev(q), dev(rho), out=out), lambda: eng.interp_c(dev(Zh), dev(zh), dev(q), dev(rho)),
         big[:, nL:nL + nG], torch.float64),
        (lambda out: eng.plan_exner(dev(numpy.abs(fp) * 1e5), inverse=True, out=out), lambda: eng.exner(dev(numpy.abs(fp) * 1e5), inverse=True),
         torch.empty(n, nG, dtype=torch.float64, device="cuda"), torch.float64),
        (lambda out: eng.plan_rms(dev(q), out=out), lambda: eng.rms(dev(q)), torch.empty(n, dtype=torch.float64, device="cuda"), torch.float64),
    ]
    for make, call, out, dt in cases:
        want = call().cpu().numpy()
        plan = make(out)
        for _ in range(3):                                   # relaunching the frozen plan rewrites the same destination
            out.fill_(7 if dt == torch.int64 else 7.0)
            got = plan.run()
            assert got.data_ptr() == out.data_ptr()
            assert numpy.array_equal(got.cpu().numpy(), want, equal_nan=True)
        assert numpy.array_equal(make(None).run().cpu().numpy(), want, equal_nan=True)
    assert (big[:, nL + nG:] == -1.0).all()                  # nothing written outside the views
    with pytest.raises(ValueError):
        eng.plan_interp(dev(x), dev(xp), dev(fp), out=torch.empty(n, nL + 1, dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        eng.plan_rms(dev(q), out=torch.empty(n, dtype=torch.float32, device="cuda"))


def test_a_plan_on_a_view_it_cannot_read_in_place_follows_the_callers_tensor():
    """round-4 advisor (medium): a K7 plan built on a tensor the kernels cannot read in place (a transposed view, a shared fp,
    q / rho of different pitch, a strided pressure field) holds a PRIVATE packed copy -- which run() refreshes from the
    caller's tensor before every launch, so an in-place update of the source is seen (rounds 3-4 recomputed from the stale
    snapshot without a word).  Row-contiguous arguments are read in place: no copy, run() is one foreign call."""
    from sp_coupler_amd.engine import Engine
    eng = Engine("cuda:0")
    rng = numpy.random.default_rng(21)
    n, nG, nL = 64, 91, 160
    dev = lambda a: torch.from_numpy(numpy.ascontiguousarray(a)).cuda()       # noqa: E731
    xp = dev(numpy.sort(rng.uniform(0, 4000, size=(n, nG)), axis=1))
    x = dev(rng.uniform(-100, 4100, size=(n, nL)))
    fpT = dev(rng.normal(size=(nG, n)))                         # fp as a TRANSPOSED view: [n x nG] with row stride 1
    plan = eng.plan_interp(x, xp, fpT.t())
    assert len(plan._refresh) == 1
    import ctypes
    with pytest.raises(RuntimeError, match="use run"):          # the bare foreign call would skip the refresh: refused
        plan.launch_raw(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    for _ in range(2):
        want = numpy.stack([numpy.interp(x[r].cpu().numpy(), xp[r].cpu().numpy(), fpT.t()[r].cpu().numpy()) for r in range(n)])
        assert numpy.array_equal(plan.run().cpu().numpy(), want)
        fpT.mul_(-3.0).add_(1.0)                               # the caller updates ITS tensor in place; the plan follows
    shared = dev(rng.normal(size=nG))                           # one fp row for all rows: spread per run
    plan = eng.plan_interp(x, xp, shared)
    for _ in range(2):
        want = numpy.stack([numpy.interp(x[r].cpu().numpy(), xp[r].cpu().numpy(), shared.cpu().numpy()) for r in range(n)])
        assert numpy.array_equal(plan.run().cpu().numpy(), want)
        shared.add_(2.5)
    Zh = dev(numpy.sort(rng.uniform(0, 6000, size=(n, nG + 1)), axis=1)[:, ::-1])
    zh = dev(numpy.arange(nL) * 25.0)
    q, rho_wide = dev(rng.normal(size=(n, nL))), dev(rng.uniform(0.5, 1.3, size=(n, nL + 8)))
    rho = rho_wide[:, :nL]                                      # another pitch than q: both packed, and re-packed per run
    plan = eng.plan_interp_c(Zh, zh, q, rho)
    for _ in range(2):
        want = eng.interp_c(Zh, zh, q.clone(), rho.contiguous()).cpu().numpy()
        assert numpy.array_equal(plan.run().cpu().numpy(), want, equal_nan=True)
        q.mul_(0.5)
        rho_wide.add_(0.1)
    p = dev(rng.uniform(1e3, 1e5, size=(n, 2 * nG)))
    plan = eng.plan_exner(p[:, ::2], inverse=True)             # every second pressure: strided
    for _ in range(2):
        assert numpy.array_equal(plan.run().cpu().numpy(), eng.exner(p[:, ::2].contiguous(), inverse=True).cpu().numpy())
        p.mul_(0.9)
    # what can be read in place is: the plan holds the caller's pointer and no copy
    plan = eng.plan_interp(x, xp, fpT.t().contiguous())
    assert plan._refresh == () and eng.plan_rms(q)._refresh == () and eng.plan_searchsorted(xp, x)._refresh == ()


def test_interp_left_right_on_the_device(sputils):
    rng = numpy.random.default_rng(10)
    xp = numpy.sort(rng.uniform(0, 100, size=(50, 91)), axis=1)
    fp, x = rng.normal(size=(50, 91)), rng.uniform(-30, 130, size=(50, 160))
    for kw in ({"left": -7.5}, {"right": 9.25}, {"left": 0.0, "right": numpy.inf}):
        want = numpy.stack([numpy.interp(x[r], xp[r], fp[r], **kw) for r in range(50)])
        assert_bits("interp %s" % kw, sputils.interp(x, xp, fp, **kw), want)


@pytest.mark.parametrize("n", [1, 2, 5, 40, 64, 91, 127, 128, 130, 300, 600, 1100, 5000])
def test_every_search_depth_and_the_unstaged_rows(sputils, n):
    """round 4: the staged kernels search NaN-padded LDS rows with a trip count fixed at compile time for rows of 64-1023
    entries (SL 7-10), at run time otherwise (SL 0), and rows beyond the LDS (5000 entries) read global memory (SL -1):
    every path against NumPy, shared and per-row sample arrays, exact hits, values outside, NaN / inf queries"""
    rng = numpy.random.default_rng(100 + n)
    rows = 37
    xp = numpy.sort(rng.uniform(0, 1e4, size=(rows, n)), axis=1)
    fp = rng.normal(size=(rows, n))
    x = rng.uniform(-50, 1.005e4, size=(rows, 23))
    x[:, 0], x[:, 1], x[:, 2] = xp[:, 0], xp[:, -1], xp[:, n // 2]          # exact hits incl. both ends
    x[3, 5], x[4, 6], x[5, 7] = numpy.nan, numpy.inf, -numpy.inf
    assert_bits("interp n=%d" % n, sputils.interp(x, xp, fp), _interp_rows(x, xp, fp))
    assert_bits("interp shared xp n=%d" % n, sputils.interp(x, xp[0], fp), _interp_rows(x, xp[0], fp))
    for side in ("left", "right"):
        want = numpy.stack([numpy.searchsorted(xp[r], x[r], side=side) for r in range(rows)])
        assert numpy.array_equal(sputils.searchsorted(xp, x, side=side), want), (n, side)
        want = numpy.stack([numpy.searchsorted(xp[0], x[r], side=side) for r in range(rows)])
        assert numpy.array_equal(sputils.searchsorted(xp[0], x, side=side), want), (n, side, "shared")
    if n >= 2:                                                            # the fine grid of interp_c: n points, n - 1 cells
        zh = numpy.cumsum(rng.uniform(0.5, 30, size=(rows, n)), axis=1)
        Zh = numpy.sort(rng.uniform(zh[:, :1] - 5, zh[:, -1:] * 1.1, size=(rows, 12)), axis=1)[:, ::-1].copy()
        Zh[:, -1] = zh[:, 0]                                              # a layer that ends exactly on the grid's first point
        q, rho = rng.normal(size=(rows, n - 1)), rng.uniform(0.5, 1.3, size=(rows, n - 1))
        with numpy.errstate(all="ignore"):
            want = numpy.stack([orc.interp_c(Zh[r], zh[r], q[r], rho[r]) for r in range(rows)])
        assert_bits("interp_c n=%d" % n, sputils.interp_c(Zh, zh, q, rho), want)
        if n <= 1100:
            Zh0 = numpy.sort(rng.uniform(zh[0, 0], zh[0, -1], size=(rows, 12)), axis=1)[:, ::-1].copy()
            with numpy.errstate(all="ignore"):
                want_rho = numpy.stack([orc.interp_rho(Zh0[r], zh[0], rho[r]) for r in range(rows)])
            assert_bits("interp_rho shared zh n=%d" % n, sputils.interp_rho(Zh0, zh[0], rho), want_rho)


def test_float32_engine_runs_the_helpers():
    from sp_coupler_amd.engine import Engine
    eng = Engine("cuda:0", dtype=torch.float32)
    rng = numpy.random.default_rng(10)
    xp = numpy.sort(rng.uniform(0, 3e4, size=(50, 91)), axis=1).astype(numpy.float32)
    fp = rng.normal(size=(50, 91)).astype(numpy.float32)
    x = rng.uniform(0, 3e4, size=(50, 160)).astype(numpy.float32)
    got = eng.interp(torch.from_numpy(x).cuda(), torch.from_numpy(xp).cuda(), torch.from_numpy(fp).cuda()).cpu().numpy()
    want = _interp_rows(x.astype(float), xp.astype(float), fp.astype(float))
    assert numpy.abs(got - want).max() <= 1e-4 * numpy.abs(want).max()
    idx = eng.searchsorted(torch.from_numpy(xp).cuda(), torch.from_numpy(x).cuda(), side="right").cpu().numpy()
    assert numpy.array_equal(idx, numpy.stack([numpy.searchsorted(xp[r], x[r], side="right") for r in range(50)]))
    p = torch.from_numpy(rng.uniform(1e4, 1.05e5, size=1000).astype(numpy.float32)).cuda()
    assert numpy.allclose(eng.exner(p).cpu().numpy(), orc.exner(p.cpu().numpy().astype(float)), rtol=2e-6)


def test_fuzz_random_shapes_sharing_and_pitches(sputils):
    """random row counts, sample / query lengths, shared vs per-row arrays and padded row pitches (device tensors that are
    column slices of wider buffers) against NumPy / the oracle; SPC_FUZZ_TRIALS raises the trial count"""
    from sp_coupler_amd import spcpl
    eng = spcpl.get_engine()
    trials = int(os.environ.get("SPC_FUZZ_TRIALS", "60"))
    rng = numpy.random.default_rng(int(os.environ.get("SPC_FUZZ_SEED", "4242")))

    def padded(a):
        """the same values as a device tensor whose rows are `pad` elements apart more than they need be"""
        if a.ndim == 1 or rng.random() < 0.5:
            return torch.from_numpy(numpy.ascontiguousarray(a)).cuda()
        pad = int(rng.integers(1, 9))
        buf = torch.full((a.shape[0], a.shape[1] + pad), float("nan"), dtype=torch.float64, device="cuda")
        buf[:, :a.shape[1]] = torch.from_numpy(numpy.ascontiguousarray(a)).cuda()
        return buf[:, :a.shape[1]]

    for t in range(trials):
        n, n_xp, n_x = int(rng.integers(1, 70)), int(rng.integers(1, 300)), int(rng.integers(1, 200))
        xp = numpy.sort(rng.uniform(0, 1e4, size=(n, n_xp)), axis=1)
        fp = rng.normal(size=(n, n_xp))
        x = rng.uniform(-100, 1.01e4, size=(n, n_x))
        if n_xp > 2 and rng.random() < 0.3:
            x[:, 0] = xp[:, n_xp // 2]
        xs, xps = rng.random() < 0.3, rng.random() < 0.3
        xa, xpa = (x[0] if xs else x), (xp[0] if xps else xp)
        got = eng.interp(padded(xa), padded(xpa), padded(fp)).cpu().numpy()
        assert_bits("fuzz interp %d" % t, got, _interp_rows(xa, xpa, fp))
        side = "left" if rng.random() < 0.5 else "right"
        gi = eng.searchsorted(padded(xpa), padded(xa), side=side).cpu().numpy()
        wi = numpy.stack([numpy.searchsorted(xpa if xps else xpa[r], xa if xs else xa[r], side=side) for r in range(n)])
        assert numpy.array_equal(gi, wi[0] if (xs and xps) else wi), ("fuzz searchsorted", t)      # two 1-D arguments: one row
        # conservative coarsening: coarse bounds descending from above the fine grid's top down to its bottom
        nL, nG = int(rng.integers(2, 400)), int(rng.integers(1, 60))
        zh = numpy.cumsum(rng.uniform(0.5, 30, size=(n, nL)), axis=1)
        zsh = rng.random() < 0.4
        za = zh[0] if zsh else zh
        top = (za[-1] if zsh else za[:, -1:]) * rng.uniform(0.6, 1.5)
        bot = (za[0] if zsh else za[:, :1])
        Zh = numpy.sort(rng.uniform(0, 1, size=(n, nG + 1)), axis=1)[:, ::-1] * (top - bot) + bot
        Zh[:, -1] = bot if zsh else bot[:, 0]
        q, rho = rng.normal(size=(n, nL)), rng.uniform(0.4, 1.3, size=(n, nL))
        with numpy.errstate(all="ignore"):
            want = numpy.stack([orc.interp_c(Zh[r], za if zsh else za[r], q[r], rho[r]) for r in range(n)])
        got = eng.interp_c(padded(numpy.ascontiguousarray(Zh)), padded(za), padded(q), padded(rho)).cpu().numpy()
        assert_bits("fuzz interp_c %d" % t, got, want)


@pytest.mark.parametrize("seed,n", [(1, 5003), (2, 9001), (3, 14011)])
def test_multi_row_slabs_equal_single_row_slabs(seed, n):
    """The staged operators give a workgroup a SLAB of several rows once a launch has enough rows (>= 2 048; below that every
    workgroup owns one row -- the shape all the oracle comparisons above run in; a slab has at most n_rows / 1024 rows, so 5 003 ...
    14 011 rows reach slabs of 4 ... 13).  Row counts that leave a partial last slab, random lengths, shared or per-row sample arrays, padded pitches, NaN / inf / out-of-range points: the launch
    over all rows must equal, bit for bit, the same rows launched 500 at a time (one row per workgroup, oracle-checked above)."""
    from sp_coupler_amd import spcpl
    eng = spcpl.get_engine()
    rng = numpy.random.default_rng(900 + seed)

    def dev(a, pad):
        if a.ndim == 1 or not pad:
            return torch.from_numpy(numpy.ascontiguousarray(a)).cuda()
        buf = torch.full((a.shape[0], a.shape[1] + pad), float("nan"), dtype=torch.float64, device="cuda")
        buf[:, :a.shape[1]] = torch.from_numpy(numpy.ascontiguousarray(a)).cuda()
        return buf[:, :a.shape[1]]

    def chunks(fn, *arrs):
        """fn over rows [i, i + 500) of every 2-D argument (1-D ones are shared by all rows)"""
        out = [fn(*[a if a.dim() == 1 else a[i:i + 500] for a in arrs]) for i in range(0, n, 500)]
        return torch.cat(out).cpu().numpy()

    def same(what, full, parts):
        full = full.cpu().numpy()
        assert full.shape == parts.shape, what
        assert numpy.array_equal(full.view(numpy.int64) if full.dtype == numpy.float64 else full,
                                 parts.view(numpy.int64) if parts.dtype == numpy.float64 else parts), what

    for trial in range(4):
        n_xp, n_x = int(rng.integers(2, 280)), int(rng.integers(1, 330))
        xp = numpy.sort(rng.uniform(0, 1e4, size=(n, n_xp)), axis=1)
        fp = rng.normal(size=(n, n_xp))
        x = rng.uniform(-100, 1.01e4, size=(n, n_x))
        x[:, 0] = xp[:, n_xp // 2]
        x[7, n_x // 2], x[4999, 0], x[n - 1, n_x - 1] = numpy.nan, numpy.inf, -numpy.inf
        xs, xps = rng.random() < 0.4, rng.random() < 0.4
        pad = int(rng.integers(0, 6))
        xd, xpd, fpd = dev(x[0] if xs else x, pad), dev(xp[0] if xps else xp, pad), dev(fp, pad)
        same("interp %d" % trial, eng.interp(xd, xpd, fpd), chunks(eng.interp, xd, xpd, fpd))
        for side in ("left", "right"):
            f = lambda a, v: eng.searchsorted(a, v, side=side)    # noqa: E731
            if not (xs and xps):
                same("searchsorted %s %d" % (side, trial), f(xpd, xd), chunks(f, xpd, xd))
        nL, nG = int(rng.integers(3, 300)), int(rng.integers(1, 120))
        zh = numpy.cumsum(rng.uniform(0.5, 30, size=(n, nL)), axis=1)
        zsh = rng.random() < 0.5
        za = zh[0] if zsh else zh
        top = (za[-1] if zsh else za[:, -1:]) * rng.uniform(0.6, 1.5, size=(n, 1))
        bot = (za[0] if zsh else za[:, :1])
        Zh = numpy.sort(rng.uniform(0, 1, size=(n, nG + 1)), axis=1)[:, ::-1] * (top - bot) + bot
        Zh[:, -1] = bot if zsh else bot[:, 0]
        Zh[11, nG // 2] = -3.0                                       # an end point below the grid: None -> NaN
        q, rho = rng.normal(size=(n, nL)), rng.uniform(0.4, 1.3, size=(n, nL))
        Zd, zd, qd, rd = dev(numpy.ascontiguousarray(Zh), pad), dev(za, pad), dev(q, pad), dev(rho, pad)
        same("interp_c %d" % trial, eng.interp_c(Zd, zd, qd, rd), chunks(eng.interp_c, Zd, zd, qd, rd))
        full = eng.interp_c(Zd, zd, qd, rd).cpu().numpy()
        assert numpy.isnan(full[11]).any() and (full != 0).any() and (full == 0).any()


def test_fp32_exner_is_the_host_evaluation_of_spc_powf_bit_for_bit(tmp_path):
    """the fp32 variant's power (csrc/spc_powf.h: evaluated inside double arithmetic, rounded once; round 5, replaces ocml's
    powf) is one source of exactly specified IEEE operations for device and host: the device's exner / iexner in float32
    equals the header compiled with gcc, bit for bit -- so the host sweep's bound (tests/test_pow_accuracy.py: correctly
    rounded except within 2^-14 ulp of a tie) IS the device's bound.  Reference: splib/sputils.py:28-34 in float32."""
    import ctypes
    import subprocess
    from sp_coupler_amd.engine import Engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libpowf_host.so")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", "-o", so,
                    os.path.join(root, "tools", "csrc", "pow_accuracy.c"), "-lm"], check=True)
    host = ctypes.CDLL(so)
    host.spc_powf_host.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_long]
    rng = numpy.random.default_rng(55)
    p = numpy.concatenate([
        numpy.exp(rng.uniform(numpy.log(10.0), numpy.log(1.2e5), 1 << 20)),                   # the atmosphere's pressures
        numpy.exp(rng.uniform(numpy.log(1e-30), numpy.log(1e30), 1 << 18)),                   # any exponent
        numpy.array([1e5, 101325.0, 2e-40, 1e-38, 3.0e38])]).astype(numpy.float32)           # subnormal quotients included
    e32 = Engine("cuda:0", dtype=torch.float32)
    x = p / numpy.float32(1e5)                                                                # the kernel's float division
    assert (x > 0).all() and (x[-3:-1] < 1.1754944e-38).all()
    rd, cp = numpy.float32(287.04), numpy.float32(1004.)
    for inverse, y in ((False, rd / cp), (True, (-rd) / cp)):
        got = e32.exner(torch.from_numpy(p).cuda(), inverse=inverse).cpu().numpy()
        want = numpy.empty_like(x)
        host.spc_powf_host(x.ctypes.data, ctypes.c_float(float(y)), want.ctypes.data, x.size)
        assert numpy.array_equal(got.view(numpy.uint32), want.view(numpy.uint32)), int((got.view(numpy.uint32) != want.view(numpy.uint32)).sum())
        # and against the exact power: half an ulp (+ the 2^-14 ulp of the double evaluation)
        exact = x.astype(numpy.float64) ** float(y)
        ulp = numpy.spacing(numpy.abs(exact).astype(numpy.float32)).astype(numpy.float64)
        assert (numpy.abs(got.astype(numpy.float64) - exact) <= 0.5001 * ulp).all()
    # C99's special values, produced inline as in the double kernel
    sp = numpy.array([0.0, -0.0, numpy.inf, -numpy.inf, numpy.nan, -1.0], dtype=numpy.float32)
    got = e32.exner(torch.from_numpy(sp).cuda()).cpu().numpy()                                # y > 0
    assert got[0] == 0 and got[1] == 0 and got[2] == numpy.inf and got[3] == numpy.inf and numpy.isnan(got[4]) and numpy.isnan(got[5])
    got = e32.exner(torch.from_numpy(sp).cuda(), inverse=True).cpu().numpy()                  # y < 0
    assert got[0] == numpy.inf and got[1] == numpy.inf and got[2] == 0 and got[3] == 0 and numpy.isnan(got[4]) and numpy.isnan(got[5])
