"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of
``spcpl.variability_nudge`` (splib/spcpl.py:613-744), the ``qt_forcing == 'variance'`` branch of
``set_les_forcings`` (spcpl.py:377-382).  It evaluates the reference's expressions with the routines the reference
itself calls -- NumPy array arithmetic, ``ndarray.sum()``, ``numpy.argmax``, ``scipy.optimize.brentq`` -- for one
LES (one column), without AMUSE units (every unit on this path is SI-coherent with factor 1).

PARITY UNPINNED: the reference's tests hold no fixture for this function; the restatement follows the source line
by line (citations below) and the product kernel is compared with it.

Also here, for the CPU suite: scalar restatements of the two third-party algorithms the HIP kernel re-implements,
``brentq_restated`` (scipy/optimize/Zeros/brentq.c of the installed scipy 1.15.3, absent from /root/reference) and
``npsum_restated`` (numpy 2.2.6 ``ndarray.sum()`` of a contiguous float64 array: pairwise blocks of 128 with 8
accumulators, halves split at multiples of 8, 8192-element chunks), each checked bit for bit against the library
routine in tests/test_vnudge.py.
"""
import math

import numpy
from scipy.optimize import brentq

rlv, cp, rd, pref0 = 2.53e6, 1004., 287.04, 1e5          # splib/sputils.py:14-20


def exner(p):                                             # splib/sputils.py:28-29
    return (p / pref0) ** (rd / cp)


def make_R(itot, jtot):
    """spcpl.py:620-621: the zero-mean Gaussian field, from numpy's GLOBAL generator like the reference"""
    R = numpy.random.normal(size=(itot, jtot))
    R -= R.sum() / (itot * jtot)
    return R


def variability_nudge(qt, qsat, ql_av, qt_av, presf, ql_ref, R, DT, constantT=False, thl=None, ql=None):
    """One LES. qt, qsat (, thl, ql): [itot, jtot, k]; ql_av, qt_av, presf, ql_ref: [k]; R: [itot, jtot].
    Returns dict(qt, thl, beta, alpha, qt_std, a, status); qt / thl are updated COPIES.  ``error`` holds the
    exception scipy raised (the reference would propagate it), levels processed so far are still returned."""
    qt = numpy.array(qt, dtype=numpy.float64)
    thl = None if thl is None else numpy.array(thl, dtype=numpy.float64)
    itot, jtot, kmax = qt.shape
    beta_min, beta_max = 0, 5                                                   # spcpl.py:659-660
    beta = numpy.ones(kmax)                                                     # spcpl.py:662
    a_used = numpy.zeros(kmax)
    status = numpy.zeros(kmax, dtype=numpy.int32)
    error = None
    for k in range(kmax):                                                       # spcpl.py:663
        def get_ql_diff(b):                                                     # spcpl.py:646-648
            return numpy.maximum((b * (qt[:, :, k] - qt_av[k]) + qt_av[k] - qsat[:, :, k]), 0).sum() / (itot * jtot) - ql_ref[k]

        def get_ql_diff_additive(a):                                            # spcpl.py:653-656
            return numpy.maximum((qt[:, :, k] + (a * R[:, :]) - qsat[:, :, k]), 0).sum() / (itot * jtot) - ql_ref[k]

        if ql_ref[k] > 1e-9:                                                    # spcpl.py:665
            q_min, q_max = get_ql_diff(beta_min), get_ql_diff(beta_max)
            if q_min > 0 or q_max < 0:                                          # spcpl.py:669
                beta[k] = beta_max                                              # spcpl.py:673
                status[k] = 16
            else:
                try:
                    beta[k] = brentq(get_ql_diff, beta_min, beta_max)           # spcpl.py:676
                    status[k] = 1
                except (ValueError, RuntimeError) as e:
                    error = e
                    status[k] = 1 | (256 if isinstance(e, ValueError) else 512)
                    continue
        elif ql_av[k] > ql_ref[k]:                                              # spcpl.py:679
            i, j = numpy.unravel_index(numpy.argmax(qt[:, :, k] - qsat[:, :, k]), qt[:, :, k].shape)
            beta[k] = (qsat[i, j, k] - qt_av[k]) / (qt[i, j, k] - qt_av[k])     # spcpl.py:683
            if beta[k] < 0:                                                     # spcpl.py:692-695
                beta[k] = 1
            status[k] = 2
        else:
            continue                                                            # spcpl.py:697
        if beta[k] >= beta_max:                                                 # spcpl.py:703
            if ql_ref[k] > ql_av[k]:                                            # spcpl.py:712
                try:
                    a = brentq(get_ql_diff_additive, 0, 5)                      # spcpl.py:713
                except (ValueError, RuntimeError) as e:
                    error = e
                    status[k] |= 4 | (256 if isinstance(e, ValueError) else 512)
                    beta[k] = 1
                    continue
                a_used[k] = a
                status[k] |= 4
                qt[:, :, k] += a * R                                            # spcpl.py:716,719
            else:
                status[k] |= 8
            beta[k] = 1                                                         # spcpl.py:722
        else:
            qt[:, :, k] += (beta[k] - 1) * (qt[:, :, k] - qt_av[k])             # spcpl.py:724-725
        if constantT:                                                           # spcpl.py:726-733
            ql_target = numpy.maximum((qt[:, :, k] - qsat[:, :, k]), 0)
            dQL = ql_target - ql[:, :, k]
            dTHL = - rlv / (cp * exner(presf[k])) * dQL
            thl[:, :, k] += dTHL
    alpha = numpy.log(beta) / DT                                                # spcpl.py:739
    qt_std = qt.std(axis=(0, 1))                                                # spcpl.py:743
    return dict(qt=qt, thl=thl, beta=beta, alpha=alpha, qt_std=qt_std, a=a_used, status=status, error=error)


# ---- scalar restatements of the library algorithms the kernel re-implements ---------------------------------
def brentq_restated(f, xa, xb, xtol=2e-12, rtol=8.881784197001252e-16, maxiter=100):
    """scipy/optimize/Zeros/brentq.c. Returns (root, function_calls); ValueError / RuntimeError like scipy."""
    xpre, xcur = float(xa), float(xb)
    xblk = fblk = spre = scur = 0.0
    fpre, fcur = f(xpre), f(xcur)
    calls = 2
    if fpre == 0:
        return xpre, calls
    if fcur == 0:
        return xcur, calls
    if math.copysign(1.0, fpre) == math.copysign(1.0, fcur):
        raise ValueError("f(a) and f(b) must have different signs")
    for _ in range(maxiter):
        if fpre != 0 and fcur != 0 and math.copysign(1.0, fpre) != math.copysign(1.0, fcur):
            xblk, fblk = xpre, fpre
            spre = scur = xcur - xpre
        if abs(fblk) < abs(fcur):
            xpre, xcur = xcur, xblk
            xblk = xpre
            fpre, fcur = fcur, fblk
            fblk = fpre
        delta = (xtol + rtol * abs(xcur)) / 2
        sbis = (xblk - xcur) / 2
        if fcur == 0 or abs(sbis) < delta:
            return xcur, calls
        if abs(spre) > delta and abs(fcur) < abs(fpre):
            if xpre == xblk:
                stry = -fcur * (xcur - xpre) / (fcur - fpre)
            else:
                dpre = (fpre - fcur) / (xpre - xcur)
                dblk = (fblk - fcur) / (xblk - xcur)
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre))
            if 2 * abs(stry) < min(abs(spre), 3 * abs(sbis) - delta):
                spre, scur = scur, stry
            else:
                spre = scur = sbis
        else:
            spre = scur = sbis
        xpre, fpre = xcur, fcur
        if abs(scur) > delta:
            xcur += scur
        else:
            xcur += delta if sbis > 0 else -delta
        fcur = f(xcur)
        calls += 1
    raise RuntimeError("Failed to converge after %d iterations." % maxiter)


def _leaf(a, lo, n):
    if n < 8:
        res = 0.0
        for i in range(n):
            res += a[lo + i]
        return res
    r = [a[lo + j] for j in range(8)]
    i = 8
    while i < n - (n % 8):
        for j in range(8):
            r[j] += a[lo + i + j]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        res += a[lo + i]
        i += 1
    return res


def _pairwise(a, lo, n):
    if n <= 128:
        return _leaf(a, lo, n)
    n2 = n // 2
    n2 -= n2 % 8
    return _pairwise(a, lo, n2) + _pairwise(a, lo + n2, n - n2)


def npsum_restated(a):
    """ndarray.sum() of a contiguous float64 array, scalar by scalar"""
    a = [float(x) for x in numpy.asarray(a, dtype=numpy.float64).ravel()]
    res, lo = 0.0, 0
    while lo < len(a):
        c = min(8192, len(a) - lo)
        res += _pairwise(a, lo, c)
        lo += c
    return res
