/* spc_pow.h -- x**y for the two exponents of the coupling path, y = -+rd/cp (splib/sputils.py:28-34): |y| <= 1, x = p / pref0.
 * ONE source for the device (spc_hip.hip includes it with SPC_POW_FN = __device__ __forceinline__) and for the host
 * accuracy sweep (tools/csrc/pow_accuracy.c: the same IEEE operations, checked against powl in 80-bit arithmetic).
 *
 * ocml's general pow() is 245 instructions and an out-of-line call (which made every K1 wave reserve its 100 registers);
 * with the exponent's size known ~80 instructions reach a tighter bound than round 2-3's 1.2 ulp:
 *   log x = e ln2 + log m, m in [sqrt 1/2, sqrt 2), log m = 2 atanh f = 2 f + f s P(s), f = (m - 1) / (m + 1), s = f f;
 *           f is carried as f + f_lo (the quotient's own rounding error, from the exact remainder of the division) and the
 *           sum e ln2_hi + 2 f as A + a_err (TwoSum), so log x = A + B with ~2^-70 relative error;
 *   t = y log x as t + tl (one fma recovers the product's rounding error); n = rint(t log2 e);
 *   r = t - n ln2 as r + r_lo; exp = 1 + r + r^2/2 + (r^3 q(r) + r_lo (1 + r)) with 1 + r + r^2/2 accumulated exactly
 *           (two Fast2Sums, r^2 with its fma remainder), so the ONLY rounding of full size is the last addition and the
 *           rounded small terms are below 0.008.
 * Measured (tools/pow_accuracy.py, 4e7 points per exponent against powl, x from 1e-6 to 1.2 and over the whole exponent
 * range): see profiles/r04_pow_accuracy.log -- worst error and the fraction of points above 0.5 / 0.55 ulp.
 * x must be finite and > 0 (subnormals included: frexp normalises them); the callers handle the rest. */
#ifndef SPC_POW_H
#define SPC_POW_H
#ifndef SPC_POW_FN
#include <math.h>
#define SPC_POW_FN static inline
#endif
/* 1 / x to (at least) double precision.  Device: v_rcp_f64 refined by two Newton steps (what the compiler's own division
 * expansion starts with) -- the two quotients below then cost 10 instructions instead of two IEEE divisions (~30); host:
 * the correctly rounded quotient.  The two differ by a few 2^-53, which only enters through f_lo (itself < 2^-52 f). */
#ifdef __HIP_DEVICE_COMPILE__
SPC_POW_FN double spc_pow_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}
#else
SPC_POW_FN double spc_pow_rcp(double x) { return 1.0 / x; }
#endif

SPC_POW_FN double spc_pow_pos(double x, double y)
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10, LOG2E = 1.44269504088896338700e+00;
    int e;
    double m = frexp(x, &e);                                         /* [0.5, 1) */
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }            /* [sqrt 1/2, sqrt 2) */
    const double num = m - 1.0;                                      /* exact */
    const double den = m + 1.0, den_lo = m - (den - 1.0);            /* m + 1 = den + den_lo exactly */
    const double rc = spc_pow_rcp(den);
    const double f = num * rc;                                       /* any f within a few ulp of the quotient will do: */
    const double f_lo = (__builtin_fma(-f, den, num) - f * den_lo) * rc;      /* (m-1)/(m+1) = f + f_lo, f_lo from the exact remainder */
    const double s = f * f;
    double P = 2.0 / 21.0;
    P = __builtin_fma(P, s, 2.0 / 19.0); P = __builtin_fma(P, s, 2.0 / 17.0); P = __builtin_fma(P, s, 2.0 / 15.0);
    P = __builtin_fma(P, s, 2.0 / 13.0); P = __builtin_fma(P, s, 2.0 / 11.0); P = __builtin_fma(P, s, 2.0 / 9.0);
    P = __builtin_fma(P, s, 2.0 / 7.0); P = __builtin_fma(P, s, 2.0 / 5.0); P = __builtin_fma(P, s, 2.0 / 3.0);
    const double lo = __builtin_fma(f * s, P, 2.0 * f_lo);           /* log m = 2 f + lo */
    const double ed = (double)e;
    const double L_hi = ed * LN2_HI;                                 /* exact: LN2_HI has 32 significant bits */
    const double f2 = 2.0 * f;
    const double A = L_hi + f2, bb = A - L_hi;
    const double a_err = (L_hi - (A - bb)) + (f2 - bb);              /* TwoSum: L_hi + 2 f = A + a_err exactly */
    const double B = __builtin_fma(ed, LN2_LO, lo) + a_err;          /* log x = A + B */
    const double t_hi = y * A;
    const double t_lo = __builtin_fma(y, B, __builtin_fma(y, A, -t_hi));
    const double t = t_hi + t_lo, tl = t_lo - (t - t_hi);            /* y log x = t + tl */
    const double n = rint(t * LOG2E);
    const double r0 = __builtin_fma(-n, LN2_HI, t);                  /* exact */
    const double c = __builtin_fma(-n, LN2_LO, tl);
    const double r = r0 + c, r_lo = c - (r - r0);                    /* t - n ln2 = r + r_lo */
    double q = 1.0 / 6227020800.0;
    q = __builtin_fma(q, r, 1.0 / 479001600.0); q = __builtin_fma(q, r, 1.0 / 39916800.0); q = __builtin_fma(q, r, 1.0 / 3628800.0);
    q = __builtin_fma(q, r, 1.0 / 362880.0); q = __builtin_fma(q, r, 1.0 / 40320.0); q = __builtin_fma(q, r, 1.0 / 5040.0);
    q = __builtin_fma(q, r, 1.0 / 720.0); q = __builtin_fma(q, r, 1.0 / 120.0); q = __builtin_fma(q, r, 1.0 / 24.0);
    q = __builtin_fma(q, r, 1.0 / 6.0);                              /* exp r = 1 + r + r^2 / 2 + r^3 q */
    const double rr = r * r, rr_err = __builtin_fma(r, r, -rr);      /* r^2 = rr + rr_err exactly */
    const double h = 0.5 * rr;
    const double hi = 1.0 + r, e1 = r - (hi - 1.0);                  /* 1 + r = hi + e1 exactly (Fast2Sum) */
    const double hi2 = hi + h, e2 = h - (hi2 - hi);                  /* hi + h = hi2 + e2 exactly (hi >= 0.65 > h) */
    const double tail = __builtin_fma(rr * r, q, __builtin_fma(r_lo, r, r_lo));        /* r^3 q + r_lo (1 + r): < 0.008 */
    return ldexp(hi2 + (((e1 + e2) + 0.5 * rr_err) + tail), (int)n); /* the only rounding of full size */
}
#endif
