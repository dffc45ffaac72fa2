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


class Quantity:
    """the part of an AMUSE quantity the helpers touch"""

    def __init__(self, number):
        self.number = numpy.asarray(number)


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
    assert su.rms(numpy.ones((4, 9))).shape == (4,)
    with pytest.raises(NotImplementedError):
        su.interp(x[0], xp[0], fp[0], left=0.0)
    with pytest.raises(NotImplementedError):
        su.searchsorted(xp[0], x[0], sorter=None)
    with pytest.raises(ValueError):
        su.interp(numpy.zeros(3), numpy.zeros(4), numpy.zeros(5))


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
    Zh = numpy.linspace(z[-1] * 1.2, z[0], 8)
    assert numpy.array_equal(su.interp_c(Zh, z, q, w), orc.interp_c(Zh, z, q, w))
    assert numpy.array_equal(su.interp_rho(Zh, z, w), orc.interp_rho(Zh, z, w))
    both = su.interp_c(numpy.stack([Zh, Zh]), z, numpy.stack([q, 2 * q]), numpy.stack([w, w]))
    assert both.shape == (2, 7) and numpy.array_equal(both[1], orc.interp_c(Zh, z, 2 * q, w))
