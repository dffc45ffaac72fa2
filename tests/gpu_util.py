"""Helpers shared by the -m gpu parity tests."""
import numpy
import torch

EPS = 2.220446049250313e-16


def to_dev(d, device, dtype=torch.float64):
    out = {}
    for k, v in d.items():
        t = torch.from_numpy(numpy.ascontiguousarray(v))
        out[k] = t.to(device=device, dtype=dtype if t.is_floating_point() else t.dtype)
    return out


def host(t):
    return t.detach().cpu().numpy()


def assert_bits(name, got, want):
    """bit-exact including the sign of zero and NaN positions"""
    got, want = numpy.asarray(got), numpy.asarray(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    same = (got == want) | (numpy.isnan(got) & numpy.isnan(want))
    assert same.all(), "%s: %d of %d elements differ (max abs %.3e)" % (
        name, (~same).sum(), same.size, numpy.nanmax(numpy.abs(got - want)))
    num = ~numpy.isnan(want)   # the sign bit of a NaN carries no meaning (differs between x86 and gfx950)
    assert numpy.array_equal(numpy.signbit(got)[num], numpy.signbit(want)[num]), name + ": sign of zero differs"


def assert_close_scaled(name, got, want, tol, scale):
    err = numpy.abs(numpy.asarray(got) - numpy.asarray(want))
    assert numpy.isfinite(err).all(), name
    assert err.max() <= tol * scale, "%s: max abs err %.3e > %.3e (tol %.1e x scale %.3e)" % (
        name, err.max(), tol * scale, tol, scale)
