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
 * Host and device differ in ONE operation, spc_pow_rcp (below): the device refines v_rcp_f64 by two Newton steps (within 1 ulp of
 * 1 / x), the host divides (correctly rounded).  The sweep therefore also runs with the host reciprocal pushed -2 ... +2 ulp off
 * (mode `p` of tools/csrc/pow_accuracy.c: a superset of what the device can produce) and takes the worst case:
 * profiles/r05_pow_accuracy.log -- that figure, not the unperturbed one, is the bound claimed for the device.
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
#elif defined(SPC_POW_RCP_HOST)
SPC_POW_FN double spc_pow_rcp(double x) { return SPC_POW_RCP_HOST(x); }    /* the host sweep's stand-in for the device's refined v_rcp_f64 */
#else
SPC_POW_FN double spc_pow_rcp(double x) { return 1.0 / x; }
#endif

/* p / 1e5 (the reference's p / pref0, splib/sputils.py:29,34) WITHOUT a division instruction sequence, for the standalone exner
 * operator (bound by VALU issue): Markstein's iteration.  With rc = RN(1 / c): q0 = RN(p rc) is within 2 ulp of p / c; one
 * residual step r = p - q c (fma), q1 = RN(q0 + r rc) makes it faithful (error below 1 ulp); a second one then yields the
 * CORRECTLY ROUNDED quotient (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic, the theorem on division
 * iterations: faithful q, |rc - 1/c| < 2^-53 / c, r exact => RN(q + r rc) = RN(p / c)).  5 operations instead of the ~11 (one of
 * them quarter rate) of the compiler's division expansion.  Valid while neither the quotient nor the residuals leave the
 * normal range: the caller keeps p in [2^-900, 2^900] and divides otherwise.  tools/csrc/pow_accuracy.c compares it with
 * the division on 4e8 random and structured arguments. */
SPC_POW_FN double spc_div_pref0_markstein(double p)
{
    const double c = 1e5, rc = 1e-5;                                 /* the literal 1e-5 = RN(10^-5) = RN(1 / c) */
    double q = p * rc;
    q = __builtin_fma(__builtin_fma(-q, c, p), rc, q);
    return __builtin_fma(__builtin_fma(-q, c, p), rc, q);
}

#include "spc_pow_coefs.h"
#ifdef __HIPCC__
static __device__ const double spc_pow_lit[21] = {SPC_POW_COEFS};
#else
static const double spc_pow_lit[21] = {SPC_POW_COEFS};
#endif
/* spc_pow_pos: coefficients as literals (constant indices into a const array fold).  What K1 / K5 / K6 use: they are bound by
 * memory and short of scalar registers. */
#define SPC_POW_NAME spc_pow_pos
#define SPC_PC(i) spc_pow_lit[i]
#include "spc_pow_body.inc"
#undef SPC_POW_NAME
#undef SPC_PC
#if defined(__HIPCC__) && defined(SPC_POW_TABLE)
/* spc_pow_pos_tab: the same operations with the coefficients read from the __constant__ table SPC_POW_TABLE, i.e. held in SCALAR
 * registers and used as the scalar operand of the fma -- a 64-bit literal costs two v_mov per use instead (26 % of the VALU
 * instructions of a pow).  For the standalone exner operator, which is bound by VALU issue (profiles/r04_k7_counters.log).  The
 * table must have EXTERNAL linkage (the includer defines it at file scope): one the optimizer can prove constant is folded back
 * into literals.  Same operations, same bits. */
#define SPC_POW_NAME spc_pow_pos_tab
#define SPC_PC(i) SPC_POW_TABLE[i]
#include "spc_pow_body.inc"
#undef SPC_POW_NAME
#undef SPC_PC
#endif
#endif
