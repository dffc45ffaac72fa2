/* spc_powf.h -- x**y of the fp32 arithmetic variant (BASELINE config 5's tolerance sweep) for the two exponents of the coupling
 * path, y = -+rd/cp (splib/sputils.py:28-34), x = p / pref0.  ONE source for the device (spc_hip.hip includes it with SPC_POW_FN =
 * __device__ __forceinline__) and for the host sweep (tools/csrc/pow_accuracy.c, mode `f`).
 *
 * Rounds 1-4 called ocml's powf(): an out-of-line call, so every K1<float> wave reserved the callee's registers (the form
 * that cost the fp64 K1 two waves per SIMD, spc_hip.hip).  Here the power is evaluated INSIDE double arithmetic (v_fma_f64
 * issues at the rate of v_fma_f32 on gfx950) to ~2^-39 relative and rounded ONCE to float:
 *   log x = e ln2 + 2 atanh f,  f = (m - 1) / (m + 1),  m in [sqrt 1/2, sqrt 2)  (|f| <= 0.1716: Taylor up to f^13, 1.3e-12)
 *   t = y log x;  n = rint(t log2 e);  r = t - n ln2  (|r| <= 0.347: Taylor of exp up to r^10, 2.2e-13);  2^n by ldexp
 * i.e. the result is the correctly rounded float except where the exact power lies within 2^-15 ulp of a rounding boundary:
 * <= 0.5 + 2^-14 ulp.  Every operation is an exactly specified IEEE one (frexp, ldexp, rint, fma, /, conversions; the
 * division is a true division, not a refined reciprocal), so the host build computes the device's bits: the sweep's bound
 * IS the device's bound (tests/test_sputils_gpu.py compares the two bit for bit on the GPU box).
 * x must be finite and > 0 (subnormal floats included: the conversion to double normalises them); any finite y. */
#ifndef SPC_POWF_H
#define SPC_POWF_H
#ifndef SPC_POW_FN
#include <math.h>
#define SPC_POW_FN static inline
#endif
SPC_POW_FN float spc_powf_pos(float xf, float yf)
{
    const double LN2 = 6.93147180559945286227e-01, LOG2E = 1.44269504088896338700e+00;
    int e;
    double m = frexp((double)xf, &e);                                /* [0.5, 1) */
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }            /* [sqrt 1/2, sqrt 2) */
    const double f = (m - 1.0) / (m + 1.0);
    const double s = f * f;
    /* the coefficients of the SMALL terms (s^3 and beyond here, r^6 and beyond below) are rounded to doubles whose low 32 bits are
     * zero: such a constant is ONE 32-bit literal of a v_fma_f64 (the hardware takes it as the high word) instead of a pair of
     * scalar registers or two v_mov per use (K1<float>: 48 -> 44 VGPRs, 7 -> 4 scalar spills in its float2 form; it stays at
     * seven waves per SIMD -- 106 SGPRs, half of them its 26 array pointers).  Their 2^-21 relative rounding acts on terms below
     * 8e-6 of the result: 2^-38, inside the error budget (the sweep's worst case and its > 0.5-ulp fraction did not move). */
    double P = 0x1.3b13b00000000p-3;                                 /* 2/13 */
    P = __builtin_fma(P, s, 0x1.745d100000000p-3);                   /* 2/11 */
    P = __builtin_fma(P, s, 0x1.c71c700000000p-3);                   /* 2/9 */
    P = __builtin_fma(P, s, 0x1.2492500000000p-2);                   /* 2/7 */
    P = __builtin_fma(P, s, 2.0 / 5.0); P = __builtin_fma(P, s, 2.0 / 3.0);
    const double logm = __builtin_fma(f * s, P, 2.0 * f);            /* 2 atanh f */
    double t = (double)yf * __builtin_fma((double)e, LN2, logm);     /* y log x */
    t = t < -800.0 ? -800.0 : (t > 800.0 ? 800.0 : t);               /* beyond: 0 / inf after the conversion anyway */
    const double n = rint(t * LOG2E);
    const double r = __builtin_fma(-n, LN2, t);
    double q = 0x1.27e5000000000p-22;                                /* 1/10! */
    q = __builtin_fma(q, r, 0x1.71de400000000p-19);                  /* 1/9! */
    q = __builtin_fma(q, r, 0x1.a01a000000000p-16);                  /* 1/8! */
    q = __builtin_fma(q, r, 0x1.a01a000000000p-13);                  /* 1/7! */
    q = __builtin_fma(q, r, 0x1.6c16c00000000p-10);                  /* 1/6! */
    q = __builtin_fma(q, r, 1.0 / 120.0); q = __builtin_fma(q, r, 1.0 / 24.0);
    q = __builtin_fma(q, r, 1.0 / 6.0); q = __builtin_fma(q, r, 0.5); q = __builtin_fma(q, r, 1.0);
    q = __builtin_fma(q, r, 1.0);                                    /* exp r */
    return (float)ldexp(q, (int)n);                                  /* the one rounding to float (gradual underflow included) */
}
#endif
