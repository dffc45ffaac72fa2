"""Host logic of sp_coupler_amd/sputils.py (argument handling: scalars, 1-D columns, batches, reversed views, quantities
with ``.number``, ``integral``'s None, error behaviour) on the CPU, with the oracle-backed TEST engine standing in for the
HIP one (tests/fake_engine.py).  The arithmetic of the K7 kernels themselves is tested on the GPU (test_sputils_gpu.py)."""
import numpy
import pytest

from oracle import spcpl_oracle as orc
from tests.fake_engine import OracleEngine


@pytest.fixture()
def su():
    from sp_coupler_amd import spcpl, sputils
    spcpl.set_engine(OracleEngine())
    yield sputils
    spcpl.set_engine(None)


class Unit:
    """a length unit with a factor to metres: just enough of an AMUSE unit for `number | unit` and `value_in`"""

    __array_priority__ = 100           # as amuse.units.core.unit: `ndarray | unit` defers to __ror__

    def __init__(self, name, factor):
        self.name, self.factor = name, factor

    def __ror__(self, number):
        return Quantity(number, self)


class Quantity:
    """the part of an AMUSE quantity the helpers touch"""

    def __init__(self, number, unit=None):
        self.number = numpy.asarray(number)
        self.unit = unit

    def value_in(self, unit):
        return self.number * (self.unit.factor / unit.factor)


M, KM, KELVIN = Unit("m", 1.0), Unit("km", 1000.0), Unit("K", 1.0)


def test_reference_sputils_test_file_through_the_host_layer(su):
    tol = 1.e-10                                                     # splib/test/sputils_test.py:8
    a, b, c = 3.204, -1.2092, 9.6231
    assert abs(su.rms(numpy.array([a, b, c])) - numpy.sqrt((a * a + b * b + c * c) / 3)) < tol
    assert abs(su.rms(numpy.array([a] * 23)) - a) < tol
    assert abs(numpy.log(su.exner(2.03947 * su.pref0)) - numpy.log(2.03947) * su.rd / su.cp) < tol
    assert abs(su.exner(su.pref0) - 1) < tol
    p = 12.03947 * su.pref0
    assert abs(su.exner(p) * su.iexner(p) - 1) < tol
    pts = [(52.314970, 4.824198), (52.379932, 4.897997), (52.387264, 5.082968), (52.278097, 5.021635)]
    assert su.find_closest_points(pts, (52.356591, 4.954541))[0] == 1


def test_shapes_scalars_views_and_quantities(su):
    rng = numpy.random.default_rng(1)
    xp = numpy.sort(rng.uniform(0, 100, size=(5, 12)), axis=1)
    fp, x = rng.normal(size=(5, 12)), rng.uniform(-5, 105, size=(5, 7))
    want = numpy.stack([numpy.interp(x[r], xp[r], fp[r]) for r in range(5)])
    assert numpy.array_equal(su.interp(x, xp, fp), want)
    assert numpy.array_equal(su.interp(x, xp[0], fp), numpy.stack([numpy.interp(x[r], xp[0], fp[r]) for r in range(5)]))
    assert numpy.array_equal(su.interp(x[2], xp[2], fp[2]), want[2])
    desc_x, desc_f = xp[2][::-1], fp[2][::-1]                          # the reference passes Zf[::-1], thl_[::-1] (spcpl.py:224)
    assert numpy.array_equal(su.interp(x[2], desc_x[::-1], desc_f[::-1]), want[2])
    assert numpy.array_equal(su.interp(Quantity(x[2]), Quantity(xp[2]), Quantity(fp[2])), want[2])
    r = su.interp(50.0, xp[1], fp[1])
    assert numpy.ndim(r) == 0 and r == numpy.interp(50.0, xp[1], fp[1])
    assert numpy.ndim(su.exner(9.0e4)) == 0 and su.exner(numpy.full((2, 3), 9.0e4)).shape == (2, 3)
    idx = su.searchsorted(xp[0], x[0], side="right")
    assert idx.dtype == numpy.int64 and numpy.array_equal(idx, numpy.searchsorted(xp[0], x[0], side="right"))
    assert su.searchsorted(xp[0], 50.0) == numpy.searchsorted(xp[0], 50.0)
    assert su.rms(numpy.ones((4, 9)), axis=-1).shape == (4,)
    # numpy.interp's `period` (what **kwargs of splib/sputils.py:82-86 passes through; round-4 verdict, missing 4): numpy's own
    # host-side normalisation, then the same operator -- one column, a batch of rows, unsorted / negative / wrapped abscissae
    ang, val = rng.uniform(-400, 800, size=(5, 12)), rng.normal(size=(5, 12))
    q = rng.uniform(-1000, 1000, size=(5, 7))
    assert numpy.array_equal(su.interp(q[0], ang[0], val[0], period=360.0), numpy.interp(q[0], ang[0], val[0], period=360.0))
    assert numpy.array_equal(su.interp(q[1], ang[1], val[1], period=-360.0, left=1.0), numpy.interp(q[1], ang[1], val[1], period=-360.0, left=1.0))
    assert numpy.array_equal(su.interp(q, ang, val, period=97.5), numpy.stack([numpy.interp(q[r], ang[r], val[r], period=97.5) for r in range(5)]))
    assert numpy.array_equal(su.interp(q, ang[0], val, period=97.5), numpy.stack([numpy.interp(q[r], ang[0], val[r], period=97.5) for r in range(5)]))
    with pytest.raises(ValueError):
        su.interp(q[0], ang[0], val[0], period=0)
    with pytest.raises(TypeError):
        su.interp(q[0], ang[0], val[0], perio=3)
    with pytest.raises(NotImplementedError):
        su.searchsorted(xp[0], x[0], sorter=None)
    with pytest.raises(ValueError):
        su.interp(numpy.zeros(3), numpy.zeros(4), numpy.zeros(5))


def test_rms_is_one_number_over_the_whole_array_like_the_reference(su):
    """splib/sputils.py:23-24: numpy.sqrt(numpy.mean(a ** 2)) reduces over EVERY axis"""
    rng = numpy.random.default_rng(5)
    for shape in ((4, 9), (3, 5, 7), (2, 3, 4, 5)):
        a = rng.normal(size=shape)
        r = su.rms(a)
        assert numpy.ndim(r) == 0 and r == numpy.sqrt(numpy.mean(a ** 2)), shape
    a = rng.normal(size=(6, 11))
    assert numpy.array_equal(su.rms(a, axis=-1), numpy.array([numpy.sqrt(numpy.mean(r ** 2)) for r in a]))
    with pytest.raises(ValueError):
        su.rms(rng.normal(size=(2, 3, 4)), axis=0)


def test_interp_left_right_like_numpy(su):
    rng = numpy.random.default_rng(6)
    xp = numpy.sort(rng.uniform(0, 100, size=(4, 10)), axis=1)
    fp = rng.normal(size=(4, 10))
    x = rng.uniform(-30, 130, size=(4, 25))
    x[1, 3], x[2, 4], x[3, 5] = xp[1, 0], xp[2, -1], numpy.nan             # exactly on the ends: inside; NaN stays NaN
    for kw in ({"left": -7.5}, {"right": 9.25}, {"left": 0.0, "right": numpy.inf}):
        want = numpy.stack([numpy.interp(x[r], xp[r], fp[r], **kw) for r in range(4)])
        assert numpy.array_equal(su.interp(x, xp, fp, **kw), want, equal_nan=True), kw
        assert numpy.array_equal(su.interp(x[0], xp[0], fp[0], **kw), want[0]), kw
    assert su.interp(-1e9, xp[0], fp[0], left=3.0) == 3.0 and su.interp(50.0, xp[0], fp[0], left=3.0) == numpy.interp(50.0, xp[0], fp[0])


def test_quantities_are_converted_and_keep_their_unit(su):
    """splib/sputils.py:82-91: xp.value_in(x.unit), v.value_in(a.unit), result | fp.unit -- kilometres against metres must
    not be compared number against number"""
    rng = numpy.random.default_rng(7)
    xp_m = numpy.sort(rng.uniform(0, 4000, size=12))
    fp, x_m = rng.normal(size=12) + 290, rng.uniform(0, 4000, size=9)
    want = numpy.interp(x_m, xp_m / 1000.0 * 1000.0, fp)
    got = su.interp(Quantity(x_m, M), Quantity(xp_m / 1000.0, KM), Quantity(fp, KELVIN))
    assert isinstance(got, Quantity) and got.unit is KELVIN and numpy.array_equal(got.number, want)
    assert not isinstance(su.interp(x_m, xp_m, fp), Quantity)
    idx = su.searchsorted(Quantity(xp_m, M), Quantity(x_m / 1000.0, KM), side="right")
    assert numpy.array_equal(idx, numpy.searchsorted(xp_m, x_m / 1000.0 * 1000.0, side="right"))
    r = su.rms(Quantity(fp, KELVIN))
    assert isinstance(r, Quantity) and r.unit is KELVIN and r.number == numpy.sqrt(numpy.mean(fp ** 2))


def test_integral_interp_c_interp_rho_semantics(su, capsys):
    rng = numpy.random.default_rng(2)
    z = numpy.cumsum(rng.uniform(5, 40, size=60))
    q, w = rng.normal(size=59), rng.uniform(0.5, 1.3, size=59)
    for a, b in ((z[3] + 1.0, z[40] - 2.0), (z[50], z[10] + 0.5), (z[0], z[-1])):
        for ww in (None, w):
            assert su.integral(a, b, z, q, ww) == orc.integral(a, b, z, q, ww)
    assert su.integral(z[0] - 1.0, z[5], z, q) is None                       # sputils.py:113-115: message + None
    assert "Interval end point outside range" in capsys.readouterr().out
    su.integral(z[1], z[5], z, numpy.append(q, 0.0))                         # sputils.py:111-112: length message, still computed
    assert "len(z) should be len(q) + 1" in capsys.readouterr().out
    a = rng.uniform(z[0], z[-1], size=9); b = rng.uniform(z[0], z[-1], size=9); a[4] = z[-1] + 3
    got = su.integral(a, b, z, q, w)
    want = numpy.array([numpy.nan if i == 4 else orc.integral(a[i], b[i], z, q, w) for i in range(9)])
    assert numpy.array_equal(got, want, equal_nan=True)
    got = su.integral(z[2], b, z, q, w)                                      # ONE lower end point for all rows: broadcast
    assert numpy.array_equal(got, numpy.array([orc.integral(z[2], b[i], z, q, w) for i in range(9)]))
    Zh = numpy.linspace(z[-1] * 1.2, z[0], 8)
    assert numpy.array_equal(su.interp_c(Zh, z, q, w), orc.interp_c(Zh, z, q, w))
    assert numpy.array_equal(su.interp_rho(Zh, z, w), orc.interp_rho(Zh, z, w))
    both = su.interp_c(numpy.stack([Zh, Zh]), z, numpy.stack([q, 2 * q]), numpy.stack([w, w]))
    assert both.shape == (2, 7) and numpy.array_equal(both[1], orc.interp_c(Zh, z, 2 * q, w))
