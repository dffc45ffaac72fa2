"""Property-based CPU tests (hypothesis) of the oracle's numpy.interp / searchsorted restatements -- the
formulas the C oracle and the HIP kernels implement -- over adversarial inputs: duplicate abscissae,
signed zeros, infinities, NaNs, denormals, single-point tables."""
import ctypes

import numpy
from hypothesis import given, settings
from hypothesis import strategies as st
from hypothesis.extra import numpy as hnp

from oracle import spcpl_oracle as orc
from tests import oracle_c

finite = st.floats(allow_nan=False, allow_infinity=False, width=64, min_value=-1e12, max_value=1e12)
anyf = st.floats(allow_nan=True, allow_infinity=True, width=64)


@settings(max_examples=300, deadline=None)
@given(xp=hnp.arrays(numpy.float64, st.integers(1, 24), elements=finite),
       fp_seed=st.integers(0, 2 ** 31), x=hnp.arrays(numpy.float64, st.integers(1, 24), elements=anyf),
       special=st.integers(0, 3))
def test_interp_restatement_matches_numpy_bitwise(xp, fp_seed, x, special):
    xp = numpy.sort(xp)
    rng = numpy.random.default_rng(fp_seed)
    fp = rng.normal(size=xp.shape) * 10.0 ** rng.integers(-8, 8)
    if special == 1:
        fp[rng.integers(0, len(fp))] = numpy.inf
    elif special == 2:
        fp[rng.integers(0, len(fp))] = numpy.nan
    elif special == 3:
        fp[:] = -0.0
    with numpy.errstate(all="ignore"):
        want = numpy.interp(x, xp, fp)
    got = orc.interp_restated(x, xp, fp)
    assert numpy.array_equal(want, got, equal_nan=True)
    num = ~numpy.isnan(want)
    assert numpy.array_equal(numpy.signbit(want)[num], numpy.signbit(got)[num])


@settings(max_examples=200, deadline=None)
@given(zh=hnp.arrays(numpy.float64, st.integers(1, 40), elements=finite),
       Zh=hnp.arrays(numpy.float64, st.integers(2, 30), elements=anyf))
def test_c_cloud_index_map_matches_numpy_searchsorted(zh, Zh):
    """oracle_cloud_indices_f64 (the lo/hi/mid loop the HIP kernels use) vs numpy.searchsorted(side='right'),
    NaN ordering included"""
    zh = numpy.ascontiguousarray(numpy.sort(zh))
    Zh2 = numpy.ascontiguousarray(Zh[None, :])
    want = orc.cloud_fraction_indices(zh, Zh)
    got = oracle_c.cloud_indices(zh, Zh2)[0]
    assert want.tolist() == got.tolist()


@settings(max_examples=100, deadline=None)
@given(n=st.integers(0, 700), seed=st.integers(0, 2 ** 31))
def test_c_pairwise_sum_matches_numpy_sum(n, seed):
    """the pairwise summation replica used by the conservative-coarsening restatements"""
    lib = oracle_c.lib()
    if not hasattr(lib, "_pw_bound"):
        lib.oracle_pairwise_sum.restype = ctypes.c_double
        lib.oracle_pairwise_sum.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        lib._pw_bound = True
    rng = numpy.random.default_rng(seed)
    a = numpy.ascontiguousarray(rng.normal(size=n) * 10.0 ** rng.integers(-6, 6, size=n))
    assert lib.oracle_pairwise_sum(a.ctypes.data, n) == a.sum()
