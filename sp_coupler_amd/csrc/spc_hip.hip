// spc_hip.hip -- gfx950 (MI355X / CDNA4) kernels and C ABI of the batched SP coupling step.
//
// Replaces the serial per-column Python loops of the reference (splib/splib.py:317-323, 330-332)
// and the NumPy helpers they call (splib/spcpl.py:171-246, 299-385, 388-555, 761-764;
// splib/sputils.py:28-34, 82-91) with three launches over ALL columns:
//   K1 k_forward   GCM state -> LES-level profiles + nudging forcings (+ fused K2 index map,
//                  surface fluxes, rain rate)
//   K2 k_cloud_idx cloud-fraction level-index map (standalone form)
//   K3 k_backward  LES slab means -> GCM tendencies, masked above the LES top
//   K4 k_backward_cons  the same with conservative (rho-weighted layer-mean) coarsening
//   K5 k_diag      spifs.nc diagnostics
//   K6 k_vnudge_*  variability nudge (qt_forcing == 'variance'): spc_vnudge.hpp, spc_vnudge2.hpp
// The path is 1-D interpolation over short columns: HBM-bound, no MFMA.  Design:
// a 256-thread workgroup owns CB consecutive columns; the source profiles of those columns are
// loaded with flat, fully coalesced accesses over the contiguous [CB x n_lev] slab, converted and
// staged in LDS (reversal of the top-down GCM arrays is index arithmetic while staging); then every
// thread produces output levels of the flat [CB x n_out] slab, searching its column's LDS copy.
// Arithmetic follows numpy.interp / numpy.searchsorted operation by operation, compiled with
// FP contraction OFF so no FMA changes a rounding: level indices are bit-exact, interpolated
// values differ from the CPU only through pow().
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <initializer_list>
#include <type_traits>
#include <unordered_map>

#include "spc.h"

#pragma clang fp contract(off)

// spc_pow's coefficients as a __constant__ table with EXTERNAL linkage (file scope, outside the unnamed namespace below): the
// standalone exner kernel reads them into scalar registers (spc_pow.h: spc_pow_pos_tab); an internal table would be proved
// constant and folded back into literals
#include "spc_pow_coefs.h"
__constant__ double spc_pow_coef_table[21] = {SPC_POW_COEFS};
#define SPC_POW_TABLE spc_pow_coef_table

// Diagnostic builds only (never the shipped library; tools/exp_variants.sh): apportion kernel time.
//   -DSPC_EXP=1  every division becomes a multiplication by v_rcp_f64 (NOT bit-exact)
//   -DSPC_EXP=3  as 1, and pow() becomes a multiplication
//   -DSPC_EXP=2  as 3, and the searches are replaced by a constant index (memory + LDS traffic only)
#ifndef SPC_EXP
#define SPC_EXP 0
#endif
// Mutation control of the semantic tests (tools/mutation_control.py, never the shipped library): -DSPC_MUTANT=n perturbs ONE
// line of a kernel -- the slip a transcription of the reference could contain -- and tests/test_semantic_gpu.py must fail.
#ifndef SPC_MUTANT
#define SPC_MUTANT 0
#endif
#define SPC_MUT(n, mutated, original) (SPC_MUTANT == (n) ? (mutated) : (original))
#if SPC_EXP
__device__ __forceinline__ double spc_exp_rcp(double b) { return __builtin_amdgcn_rcp(b); }
__device__ __forceinline__ float spc_exp_rcp(float b) { return __builtin_amdgcn_rcpf(b); }
#define SPC_DIV(a, b) ((a) * spc_exp_rcp(b))
#else
#define SPC_DIV(a, b) ((a) / (b))
#endif

namespace {

#ifndef SPC_BLOCK
#define SPC_BLOCK 256
#endif
constexpr int BLOCK = SPC_BLOCK;
constexpr int MAX_LDS_BYTES = 64 * 1024;    // preferred ceiling (default dynamic-LDS limit, >= 2 workgroups per CU)
constexpr int HARD_LDS_BYTES = 160 * 1024;  // gfx950: 160 KiB per CU, reachable for one column per workgroup

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *a = "", long long b = 0, long long c = 0)
{
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}

// ---- constants: splib/sputils.py:14-20 ----------------------------------------------------------
template <typename T> struct K {
    static constexpr T pref0 = T(1e5), rd = T(287.04), rv = T(461.5), cp = T(1004.), rlv = T(2.53e6),
                       grav = T(9.81);
};

#if SPC_FASTPOW
__device__ __forceinline__ double spc_pow(double x, double y) { return exp(y * log(x)); }
__device__ __forceinline__ float spc_pow(float x, float y) { return expf(y * logf(x)); }
#else
#if SPC_EXP >= 2
__device__ __forceinline__ double spc_pow(double x, double y) { return x * y; }
#elif defined(SPC_OCML_POW)
__device__ __forceinline__ double spc_pow(double x, double y) { return pow(x, y); }
#else
// x**y for the two exponents of this path, y = -+rd/cp (sputils.py:28-34), |y| <= 1: spc_pow.h (one source for this file and
// for the host accuracy sweep tools/csrc/pow_accuracy.c; round 4: <= 0.56 ulp against an 80-bit reference, was 1.2).
// Arguments outside (0, inf) get C99 pow()'s special values for a non-integer exponent, inline (0 -> inf or 0, inf -> 0 or
// inf, negative -> NaN, -inf like +inf, NaN -> NaN); subnormal x goes through the same code (frexp normalises it).  No
// call: an out-of-line ocml pow() made every K1 wave reserve ITS 100 registers (4 waves per SIMD instead of 6).
#define SPC_POW_FN __device__ __forceinline__
#include "spc_pow.h"
// C99 pow()'s value for an x outside (0, inf) and a non-integer y
__device__ __forceinline__ double spc_pow_special(double x, double y)
{
    if (x != x) return x;                                                          // NaN
    const double big = __builtin_huge_val();
    if (x == 0.0) return y < 0.0 ? big : 0.0;                                      // +-0 (not an odd integer y)
    if (x == big || x == -big) return y < 0.0 ? 0.0 : big;                         // +-inf (not an odd integer y)
    return __builtin_nan("");                                                      // negative finite x, non-integer y
}
__device__ __forceinline__ double spc_pow(double x, double y)
{
    if (!(x > 0.0 && x <= 1.7976931348623157e308)) return spc_pow_special(x, y);
    return spc_pow_pos(x, y);
}
// (p / pref0) ** y of the standalone exner operator (sputils.py:29,34), which is bound by VALU issue: the polynomial
// coefficients come from scalar registers (spc_pow.h: spc_pow_pos_tab), and pressures in [2^-900, 2^900] -- all there are -- take
// the quotient from Markstein's iteration (spc_pow.h: correctly rounded, 5 operations) and are known to be positive and
// finite afterwards; anything else divides and may end in the special values.  Same bits as spc_pow(p / pref0, y).
__device__ __forceinline__ double spc_exner_pow(double p, double y)
{
    double x;
    if (__builtin_expect(p >= 0x1p-900 && p <= 0x1p+900, 1)) {
        x = spc_div_pref0_markstein(p);
    } else {
        x = p / 1e5;
        if (!(x > 0.0 && x <= 1.7976931348623157e308)) return spc_pow_special(x, y);
    }
    return spc_pow_pos_tab(x, y);
}
#endif
// the fp32 variant's power: spc_powf.h -- evaluated inside double arithmetic and rounded once (<= 0.5 + 2^-14 ulp, the host
// sweep computes the device's bits), inline; rounds 1-4 called ocml's powf() out of line.  Special values as for double.
#ifndef SPC_POW_FN
#define SPC_POW_FN __device__ __forceinline__
#endif
#include "spc_powf.h"
__device__ __forceinline__ float spc_pow(float x, float y)
{
#if SPC_EXP >= 2
    return x * y;
#elif defined(SPC_OCML_POW) || defined(SPC_OCML_POWF)     // A/B builds: ocml's powf (what rounds 1-4 shipped)
    return powf(x, y);
#else
    if (!(x > 0.0f && x <= 3.4028234663852886e38f)) {
        if (x != x) return x;                                                      // NaN
        const float big = __builtin_huge_valf();
        if (x == 0.0f) return y < 0.0f ? big : 0.0f;
        if (x == big || x == -big) return y < 0.0f ? 0.0f : big;
        return __builtin_nanf("");
    }
    return spc_powf_pos(x, y);
#endif
}
#endif

#if SPC_FASTPOW || SPC_EXP >= 2 || defined(SPC_OCML_POW)      // diagnostic builds: one pow for everything
__device__ __forceinline__ double spc_exner_pow(double p, double y) { return spc_pow(SPC_DIV(p, 1e5), y); }
#endif
__device__ __forceinline__ float spc_exner_pow(float p, float y) { return spc_pow(SPC_DIV(p, 1e5f), y); }

// Streaming accesses of the hot kernels: every input element is read once and every output written
// once per launch.  -DSPC_NT=1 marks them non-temporal (experiment switch); 2: the loads only, 3: the
// stores only (only the plain stores: write-through launches keep their sc1 stores).
#ifndef SPC_NT
#define SPC_NT 0
#endif
template <typename T> __device__ __forceinline__ T ldg(const T *q)
{
#if SPC_NT == 1 || SPC_NT == 2
    return __builtin_nontemporal_load(q);
#else
    return *q;
#endif
}
// WT = 1: write-through (sc1) store: nothing is left dirty in L2 for the end-of-kernel release to
// flush.  Measured on MI355X: -5 % (K1) / -7 % (K3) at 1024 columns where that flush is ~1 us of a
// ~10 us kernel, but +6 % on K3 at 35k columns -- so only the small-batch launches use it.
template <int WT, typename T> __device__ __forceinline__ void stg(T *q, T v)
{
#if SPC_NT == 1
    __builtin_nontemporal_store(v, q);
#else
    if constexpr (WT == 1)
        __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (SPC_NT == 3)
        __builtin_nontemporal_store(v, q);
    else
        *q = v;
#endif
}

// -DSPC_FASTPOW=1: x**y as exp(y*log(x)) (|y log x| < 2 on this path, ~4 ulp) instead of ocml pow (<1 ulp)
#ifndef SPC_FASTPOW
#define SPC_FASTPOW 0
#endif

// Every quotient on this path is a true IEEE division (x / y), never x * (1/y): the reference divides,
// and bit-parity of the u/v/qt/ql forcings and of all tendencies depends on it.  (Tried and measured
// slower on gfx950: RN(1/y) shared by the 5-7 slopes of a level + two FMA Newton steps + v_div_fixup;
// the magnitude-window checks it needs cost more than hipcc's v_div_scale/v_rcp/v_div_fmas expansion.)
//
// Divisor<T>: a divisor prepared once and applied to several dividends (the 5-7 slopes of a level share x1 - x0, every forcing
// of a launch divides by dt).  double: the division itself, nothing prepared -- same instructions, same bits as before.
// float (the fp32 arithmetic variant, round 5): a / b = (float)((double)a * r) with r = 1 / (double)b to ~2^-52 -- the
// CORRECTLY ROUNDED float quotient for every normal result: a quotient of two 24-bit floats is never closer than 2^-49
// (relative) to a rounding boundary of the 24-bit format, and the double product is within 2^-51.  v_cvt / v_mul_f64 / v_cvt
// issue at the rate of v_fma_f32 (tools/issue_rate.py, profiles/r05_issue_rate.log): 3 instructions per quotient + ~7 per
// distinct divisor against the ~12 (with two denormal-mode switches) of the compiler's IEEE float division; measured on the
// fp32 K1 / K3: profiles/r05_f32_div_ab.log.  0, inf and NaN divisors keep v_rcp_f64's own answer (the Newton steps would
// turn it into NaN), so x / 0 = +-inf, 0 / 0 = NaN, x / inf = 0 as IEEE has them; a subnormal QUOTIENT may differ from the
// IEEE one in its last bit (double rounding), nothing on this path is that small.
template <typename T> struct Divisor;
template <> struct Divisor<double> {
    double b;
    __device__ __forceinline__ explicit Divisor(double b_) : b(b_) {}
    __device__ __forceinline__ double div(double a) const { return SPC_DIV(a, b); }
};
template <> struct Divisor<float> {
    double r;
    __device__ __forceinline__ explicit Divisor(float b)
    {
        const double bd = (double)b, r0 = __builtin_amdgcn_rcp(bd);
        double r1 = __builtin_fma(r0, __builtin_fma(-bd, r0, 1.0), r0);
        r1 = __builtin_fma(r1, __builtin_fma(-bd, r1, 1.0), r1);
        r = (r0 != 0.0 && r0 - r0 == 0.0) ? r1 : r0;                     // finite and non-zero: refined
    }
    __device__ __forceinline__ explicit Divisor(double r_, int) : r(r_) {}       // r = RN(1 / b) known at compile time
    __device__ __forceinline__ float div(float a) const { return (float)((double)a * r); }
};
#ifdef SPC_F32_IEEE_DIV
struct DivisorF32Ieee { float b; __device__ __forceinline__ explicit DivisorF32Ieee(float b_) : b(b_) {} __device__ __forceinline__ float div(float a) const { return a / b; } };
#define SPC_DIVISOR(T) typename std::conditional<std::is_same<T, float>::value, DivisorF32Ieee, Divisor<T>>::type
#else
#define SPC_DIVISOR(T) Divisor<T>
#endif
template <typename T> __device__ __forceinline__ T div_grav(T x) { return SPC_DIV(x, K<T>::grav); }
template <typename T> __device__ __forceinline__ T div_cp(T x) { return SPC_DIV(x, K<T>::cp); }
template <typename T> __device__ __forceinline__ T div_pref0(T x) { return SPC_DIV(x, K<T>::pref0); }
#if !SPC_EXP && !defined(SPC_F32_IEEE_DIV)         // (-DSPC_F32_IEEE_DIV: the A/B build with the compiler's float division)
template <> __device__ __forceinline__ float div_grav<float>(float x) { return Divisor<float>(1.0 / (double)K<float>::grav, 0).div(x); }
template <> __device__ __forceinline__ float div_cp<float>(float x) { return Divisor<float>(1.0 / (double)K<float>::cp, 0).div(x); }
template <> __device__ __forceinline__ float div_pref0<float>(float x) { return Divisor<float>(1.0 / (double)K<float>::pref0, 0).div(x); }
#endif

// numpy NaN-aware "a < b" used by searchsorted (NaN sorts to the end)
template <typename T> __device__ __forceinline__ bool np_lt(T a, T b) { return a < b || (b != b && a == a); }

// numpy.searchsorted(a, key, side='right'): first i with key < a[i]   (splib/sputils.py:88-91)
template <typename T> __device__ __forceinline__ int ss_right(const T *a, int n, T key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = lo + ((hi - lo) >> 1);
        if (np_lt(key, a[mid])) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// numpy.searchsorted(-a, -v) (side='left'): first i with !(-a[i] < -v)   (splib/spcpl.py:498)
template <typename T> __device__ __forceinline__ int ss_left_neg(const T *a, int n, T v)
{
    const T key = -v;
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = lo + ((hi - lo) >> 1);
        if (np_lt(-a[mid], key)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// numpy.searchsorted(a, key) (side='left'): first i with !(a[i] < key) -- only the mutation control uses it (SPC_MUTANT 5)
template <typename T> __device__ __forceinline__ int ss_left_pos(const T *a, int n, T key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = lo + ((hi - lo) >> 1);
        if (np_lt(a[mid], key)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Count of xp[i] <= x for ascending xp (== numpy.interp's j + 1), fixed trip count: `p2` is the
// largest power of two <= n, so every lane runs the same floor(log2 n)+1 steps (no divergence).
template <typename T> __device__ __forceinline__ int upper_count(const T *xp, int n, int p2, T x)
{
#if SPC_EXP == 2
    return ((int)(x * T(0.001)) & 15) + 1;
#endif
    int pos = 0;
    for (int s = p2; s > 0; s >>= 1) {
        const int t = pos + s;
        const int ti = (t <= n ? t : n) - 1;
        if (t <= n && xp[ti] <= x) pos = t;
    }
    return pos;
}

// One numpy.interp evaluation given the bracketing samples (arr_interp of numpy 2.2):
//   slope = (f1-f0)/(x1-x0); r = slope*(x-x0)+f0; NaN fallbacks as in numpy.
template <typename T, typename D> __device__ __forceinline__ T lerp_np(T x, T x0, T x1, T f0, T f1, const D &dx)
{
    const T slope = dx.div(f1 - f0);               // (f1 - f0) / (x1 - x0), the divisor prepared once per level
    T r = slope * (x - x0) + f0;
    if (r != r) {
        r = slope * (x - x1) + f1;
        if (r != r && f0 == f1) r = f0;
    }
    return r;
}

template <typename T> __device__ __forceinline__ T lerp_np(T x, T x0, T x1, T f0, T f1)
{
    const T slope = (f1 - f0) / (x1 - x0);
    T r = slope * (x - x0) + f0;
    if (r != r) {
        r = slope * (x - x1) + f1;
        if (r != r && f0 == f1) r = f0;
    }
    return r;
}

// Interpolation state shared by all fields of one output level.
template <typename T> struct Bracket {
    int j;       // clamped lower sample index (0..n-2), valid when mode == 0
    int mode;    // 0 interpolate, 1 take sample `j`, 2 result is x itself (NaN)
    T x, x0, x1;
};

template <typename T> __device__ __forceinline__ Bracket<T> bracket(const T *xp, int n, int p2, T x)
{
    Bracket<T> b;
    b.x = x;
    if (n == 1) { b.mode = 1; b.j = 0; b.x0 = b.x1 = x; return b; }   // numpy lenxp == 1: fp[0], NaN x included
    if (x != x) { b.mode = 2; b.j = 0; b.x0 = b.x1 = x; return b; }
    const int j = upper_count(xp, n, p2, x) - 1;
    if (j < 0) { b.mode = 1; b.j = 0; b.x0 = b.x1 = x; return b; }                  // x < xp[0] -> fp[0]
    if (j >= n - 1) { b.mode = 1; b.j = n - 1; b.x0 = b.x1 = x; return b; }         // x >= xp[n-1] -> fp[n-1]
    b.j = j;
    b.x0 = xp[j];
    b.x1 = xp[j + 1];
    b.mode = (b.x0 == x) ? 1 : 0;                                                   // exact hit -> fp[j]
    return b;
}

template <typename T> __device__ __forceinline__ T interp_at(const Bracket<T> &b, const T *fp)
{
    if (b.mode == 2) return b.x;
    if (b.mode == 1) return fp[b.j];
    return lerp_np(b.x, b.x0, b.x1, fp[b.j], fp[b.j + 1]);
}

// ---- branch-light form used by the hot kernels -------------------------------------------------
// Every case of numpy.interp expressed as ONE predicated code path, so that the 5 (K1) / 7 (K3)
// independent slope divisions of a level sit in one basic block and interleave:
//   take : the result is the sample fp[j0] itself (x outside [xp[0], xp[n-1]], x == xp[j], n == 1)
//   nanx : the result is x itself (NaN x, n > 1)
//   else : numpy's slope form between samples j0 and j1 = j0 + 1
// For take / nanx lanes (x0, x1) = (0, 1) and j1 == j0, so the (discarded) slope arithmetic stays finite.
template <typename T> struct Br {
    int j0, j1;
    bool take, nanx;
    T x, x0, x1;
};

template <typename T> __device__ __forceinline__ Br<T> bracket2(const T *xp, int n, int p2, T x)
{
    Br<T> b;
    const int j = upper_count(xp, n, p2, x) - 1;           // NaN x: every comparison false -> j = -1
    const bool below = j < 0, above = j >= n - 1;
    const int jmax = n >= 2 ? n - 2 : 0;
    const int jc = j < 0 ? 0 : (j > jmax ? jmax : j);
    const T x0 = xp[jc], x1 = xp[jc + 1 < n ? jc + 1 : n - 1];
    b.take = (n == 1) | below | above | (x0 == x);
    b.nanx = (x != x) & (n != 1);
    b.j0 = above ? n - 1 : jc;
    b.j1 = b.take ? b.j0 : jc + 1;
    b.x = x;
    b.x0 = b.take ? T(0) : x0;
    b.x1 = b.take ? T(1) : x1;
    return b;
}

// r[k] = numpy.interp result of field k given the samples f0[k] = fp_k[j0], f1[k] = fp_k[j1]
template <int NF, typename T> __device__ __forceinline__ void interp_fields(const Br<T> &b, const T (&f0)[NF], const T (&f1)[NF], T (&r)[NF])
{
    const T t0 = b.x - b.x0;
    const SPC_DIVISOR(T) dx(b.x1 - b.x0);
    T slope[NF];
    bool any_nan = false;
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        slope[k] = dx.div(f1[k] - f0[k]);
        r[k] = slope[k] * t0 + f0[k];
        any_nan |= (r[k] != r[k]);
    }
    if (any_nan & !b.take & !b.nanx) {   // numpy's NaN fallbacks: rare, one masked block for all fields
        const T t1 = b.x - b.x1;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            if (r[k] != r[k]) {
                T q = slope[k] * t1 + f1[k];
                if (q != q && f0[k] == f1[k]) q = f0[k];
                r[k] = q;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NF; ++k) r[k] = b.nanx ? b.x : (b.take ? f0[k] : r[k]);
}

// ---- kernel parameter blocks (typed copies of the C structs) ------------------------------------
struct DimsP {
    int64_t n_cols, pitchG, pitchGh, pitchL;
    int nG, nL, cb, p2G, p2L, shared_grid, xcd_remap;
};

struct Empty {};

// optional outputs / surface coupling of the forward pass: only in the FULL kernel variant, so that the
// lean hot-path variant keeps its ~26 pointers in SGPRs without spilling
template <typename T> struct FwdOpt {
    const T *rain, *rain_last;
    T *u, *v, *thl, *qt, *ps, *Zf, *Zh, *rainrate;
    const T *Z0M, *Z0H, *QLflux, *QIflux, *SHflux, *TSflux;
    T *z0m, *z0h, *wthl, *wqt;
};

template <typename T, bool FULL> struct FwdP {
    DimsP d;
    const T *U, *V, *Tm, *SH, *QL, *QI, *Pf, *Ph, *Zgfull, *Zghalf, *zf, *zh;
    const T *u_d, *v_d, *thl_d, *qt_d, *ql_d, *ps_d;
    T factor, dt;
    T *f_u, *f_v, *f_thl, *f_qt, *f_ql, *ql_ref, *f_ps;
    int32_t *idx;
    typename std::conditional<FULL, FwdOpt<T>, Empty>::type o;
};

template <typename T> using FwdFull = FwdP<T, true>;

template <typename T> struct BwdP {
    DimsP d;
    const T *Tm, *SH, *QL, *QI, *U, *V, *A, *Zf, *Zgfull, *Zghalf, *zf;
    const T *t_d, *qt_d, *ql_d, *ql_ice_d, *u_d, *v_d, *A_prof;
    const T *zh, *Zh, *rhobf_d;    // conservative coarsening only (K4)
    T factor, dt;
    T *f_T, *f_SH, *f_QL, *f_QI, *f_U, *f_V, *f_A;
    int32_t *start_index;
};

template <typename T> struct DiagP {
    DimsP d;
    const T *Tm, *SH, *QL, *QI, *Pf, *Zgfull, *Zghalf, *zf, *thl_d, *ql_d, *ql_ice_d;
    T *Tv, *THL, *QT, *Zf, *Zh, *pf, *t, *ql_water;
};

extern __shared__ __align__(16) unsigned char spc_smem[];

// XCD-aware workgroup -> column-slab mapping.  The dispatcher deals workgroups round-robin over the 8
// XCDs (b and b+8 share one, each XCD has its own L2), while rows of 91 doubles (728 B) are not 128-B
// aligned: with the identity mapping the cache line shared by two neighbouring slabs is fetched by two
// different XCDs.  Giving each XCD a CONTIGUOUS range of slabs keeps those lines in one L2.  Speed only,
// never correctness (every slab is still processed exactly once).  Used for slabs of <= 2 columns, where
// slab boundaries are frequent (K3: +4-6 % at 35k-349k columns; 8-column slabs of K1: -1.5 %, so not there).
// -DSPC_XCD_REMAP=0 disables it altogether (A/B).
#ifndef SPC_XCD_REMAP
#define SPC_XCD_REMAP 1
#endif
__device__ __forceinline__ unsigned slab_index(int remap)
{
#if SPC_XCD_REMAP
    if (remap) {
        const unsigned b = blockIdx.x, nb = gridDim.x, x = b & 7u, j = b >> 3, q = nb >> 3, r = nb & 7u;
        return x * q + (x < r ? x : r) + j;
    }
#endif
    return blockIdx.x;
}

// Diagnostic build only (-DSPC_STAMPS, tools/stamps.py): thread 0 of each workgroup drains its memory
// counters and writes the 100 MHz wall clock at phase boundaries into a buffer no kernel code reads.
#ifdef SPC_STAMPS
__device__ unsigned long long *g_stamps = nullptr;
#define STAMP(i)                                                                     \
    do {                                                                             \
        if (threadIdx.x == 0 && g_stamps && (SPC_STAMPS == 1 || (i) == 0 || (i) == 5)) {  \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");              \
            g_stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64();                 \
        }                                                                            \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// =================================================================================================
// K1 forward: splib/spcpl.py:171-246 (convert_profiles) + 299-385 (set_les_forcings) for CB columns
// per workgroup; optional fused K2 (spcpl.py:764) and surface fluxes (spcpl.py:136-167).
// LDS per column: xp=Zf reversed | thl_ | qt_ | QL | U | V, each [nG] in ascending-height order;
// then (idx only) zh: [nL] when the LES grid is shared, else [CB x nL].
// =================================================================================================
// LES-side inputs of one output level (loaded early so their latency hides behind phase 1)
template <typename T> struct LesIn {
    T h, ud, vd, thld, qtd, qld;
};

template <typename P, typename T = decltype(+*P().zf)>
__device__ __forceinline__ LesIn<T> load_les(const P &p, int l, int64_t o)
{
    LesIn<T> r;
    r.h = p.d.shared_grid ? ldg(&p.zf[l]) : ldg(&p.zf[o]);                                        // spcpl.py:222
    r.ud = ldg(&p.u_d[o]); r.vd = ldg(&p.v_d[o]); r.thld = ldg(&p.thl_d[o]); r.qtd = ldg(&p.qt_d[o]); r.qld = ldg(&p.ql_d[o]);
    return r;
}

constexpr int cfloor_pow2(int n) { int p = 1; while (p * 2 <= n) p *= 2; return p; }

// NG / NL != 0: level counts fixed at compile time and contiguous columns (pitch == level count): the
// flat-index divisions become multiply-shifts, the searches unroll, no pitch registers (hot geometries
// 91<->160, 137<->512, 19<->160); NG == NL == 0: everything from DimsP at run time.
#ifndef SPC_K1_WAVES
#define SPC_K1_WAVES 1
#endif
#ifndef SPC_K3_WAVES
#define SPC_K3_WAVES 1
#endif
#ifndef SPC_F32_UNROLL   // work items a thread of the FLOAT K3 keeps in flight per loop round (double: always 1)
#define SPC_F32_UNROLL 2
#endif
#ifndef SPC_K1_NF        // K1: fields whose slope divisions are interleaved (5 = all at once)
#define SPC_K1_NF 5
#endif
// BLK: workgroup size.  256 everywhere except the small-batch path (small_block()): there one workgroup of 512 / 1024
// threads takes 2 / 4 columns, still one work item per thread, so that <= 256 workgroups cover the batch.
// PRE: issue the first work item's LES-side inputs and the per-column scalars in the prologue, so that ONE memory round
//      trip covers them and the GCM slab: what a single-round launch (<= 1024 columns) needs.  Multi-round launches run
//      with PRE = false: those ~20 registers are live across phase 1, whose pow() sets the kernel's register peak, and
//      without them K1 fits 6 waves per SIMD instead of 5 (75 vs 94 VGPRs) -- K1's rate follows its resident waves
//      (profiles/r02_occupancy_ab_hot.log): -7 % at 35 718 columns, +12 % at 1024 (profiles/r02_k1_occupancy6_ab.log).
template <typename T, bool FULL, int NG, int NL, int WT, int BLK = BLOCK, bool PRE = true>
__global__ __launch_bounds__(BLK, SPC_K1_WAVES) void k_forward(const FwdP<T, FULL> p)
{
    const DimsP &d = p.d;
    // The ~20 optional pointers of the FULL variant are fetched from the kernarg block where they are used (a
    // volatile scalar load each) instead of living in SGPRs for the whole kernel: 68 -> few SGPR spills.
#define OPT(f) (*(decltype(FwdOpt<T>::f) const volatile __attribute__((address_space(4))) *)( \
    (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(FwdFull<T>, o) + offsetof(FwdOpt<T>, f)))
    const int nG = NG ? NG : d.nG, nL = NL ? NL : d.nL, cb = d.cb;
    const int64_t pitchG = NG ? NG : d.pitchG, pitchGh = NG ? NG + 1 : d.pitchGh, pitchL = NL ? NL : d.pitchL;
    const int p2G = NG ? cfloor_pow2(NG ? NG : 1) : d.p2G;
    const int tid = threadIdx.x;
    const int64_t col0 = (int64_t)slab_index(d.xcd_remap) * cb;
    const int ncol = (int)((d.n_cols - col0) < cb ? (d.n_cols - col0) : cb);
    T *const lds = reinterpret_cast<T *>(spc_smem);
    T *const lzh = lds + (size_t)cb * 6 * nG;
    // work items after the barrier: [0, n2) LES levels to interpolate, [n2, n2 + nI) index-map entries
    const int n1 = ncol * nG, n2 = ncol * nL, nI = p.idx ? n1 : 0, nitems = n2 + nI;
    STAMP(0);

    // ---- prologue: issue every load that depends on nothing, so ONE memory round trip covers the
    //      GCM slab, this thread's first work item and the per-column scalars.  (Issuing the GCM loads
    //      FIRST -- the order of need -- was A/B-tested: +4 % slower here, while the same reordering
    //      gains 4.5 % in K3.) ---------------------------------------------------------------------
    LesIn<T> pre2 = {};
    T pre_zgh = T(0), pre_zs = T(0);
    if (!PRE) {
    } else if (tid < n2) {
        const int c = tid / nL, l = tid - c * nL;
        pre2 = load_les<FwdP<T, FULL>, T>(p, l, (col0 + c) * pitchL + l);
    } else if (tid < nitems) {
        const int ei = tid - n2, c = ei / nG, m = ei - c * nG;
        const int64_t gh = (col0 + c) * pitchGh;
        pre_zgh = ldg(&p.Zghalf[gh + (nG - 1 - m)]);
        pre_zs = ldg(&p.Zghalf[gh + nG]);
    }
    const int sc = BLK - 1 - tid;          // the LAST threads own the per-column scalars
    T sc_ps = T(0), sc_psd = T(0), sc_rain = T(0), sc_rl = T(0);
    if (PRE && sc < ncol) {
        sc_ps = ldg(&p.Ph[(col0 + sc) * pitchGh + nG]);                                   // spcpl.py:246
        sc_psd = ldg(&p.ps_d[col0 + sc]);
        if constexpr (FULL)
            if (OPT(rainrate)) { sc_rain = OPT(rain)[col0 + sc]; sc_rl = OPT(rain_last)[col0 + sc]; }
    }
    if (p.idx) {  // stage the LES half levels for the fused index map
        const int nz = d.shared_grid ? nL : n2;
        for (int e = tid; e < nz; e += BLK) {
            const int c = e / nL, l = e - c * nL;
            lzh[e] = d.shared_grid ? p.zh[e] : p.zh[(col0 + c) * pitchL + l];
        }
    }
    STAMP(1);

    // ---- phase 1: load GCM levels (flat over the [ncol x nG] slab), convert, stage reversed ----
    for (int e = tid; e < n1; e += BLK) {
        const int c = e / nG, k = e - c * nG;
        const int64_t col = col0 + c, g = col * pitchG + k;
        const T zsurf = ldg(&p.Zghalf[col * pitchGh + nG]);
        const T tt = ldg(&p.Tm[g]), sh = ldg(&p.SH[g]), ql = ldg(&p.QL[g]), qi = ldg(&p.QI[g]), pf = ldg(&p.Pf[g]), zg = ldg(&p.Zgfull[g]);
        const T uu = ldg(&p.U[g]), vv = ldg(&p.V[g]);
        const T zf_k = div_grav(zg - zsurf);                                          // spcpl.py:198
        T *const s = lds + (size_t)c * 6 * nG + (nG - 1 - k);                         // [::-1], spcpl.py:224
        s[0] = zf_k;
        s[2 * nG] = SPC_MUT(12, sh + ql, sh + ql + qi);                               // spcpl.py:215
        s[3 * nG] = ql;
        SPC_MUT(6, lds + (size_t)c * 6 * nG + k, s)[4 * nG] = uu;
        s[5 * nG] = vv;
        if constexpr (FULL)
            if (OPT(Zf)) OPT(Zf)[g] = zf_k;                                             // spcpl.py:200
        const T iex = spc_pow(div_pref0(pf), SPC_MUT(1, K<T>::rd, -K<T>::rd) / K<T>::cp);   // sputils.py:34
        s[nG] = SPC_MUT(8, tt + div_cp(K<T>::rlv * (ql + qi)), tt - div_cp(K<T>::rlv * (ql + qi))) * iex;   // spcpl.py:214
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);

    // ---- per-column scalars (inputs already in registers; stores drain behind phase 2) ----------
    if (sc < ncol) {
        const int64_t col = col0 + sc;
        if (!PRE) {
            sc_ps = ldg(&p.Ph[col * pitchGh + nG]); sc_psd = ldg(&p.ps_d[col]);             // spcpl.py:246
            if constexpr (FULL)
                if (OPT(rainrate)) { sc_rain = OPT(rain)[col]; sc_rl = OPT(rain_last)[col]; }
        }
        stg<WT>(&p.f_ps[col], SPC_DIVISOR(T)(p.dt).div(p.factor * SPC_MUT(14, sc_psd - sc_ps, sc_ps - sc_psd)));          // spcpl.py:332
        if constexpr (FULL) {
            if (OPT(ps)) OPT(ps)[col] = sc_ps;
            if (OPT(rainrate)) OPT(rainrate)[col] = SPC_MUT(23, sc_rl - sc_rain, sc_rain - sc_rl) / p.dt;   // spcpl.py:325
            if (OPT(wthl)) {                                                            // spcpl.py:136-167
                const T rho = sc_ps / (K<T>::rd * ldg(&p.Tm[col * pitchG + SPC_MUT(15, 0, nG - 1)]));      // spcpl.py:153
                OPT(wqt)[col] = -(OPT(QLflux)[col] + OPT(QIflux)[col] + OPT(SHflux)[col]) / rho;     // spcpl.py:159
                OPT(wthl)[col] = -OPT(TSflux)[col] * spc_pow(div_pref0(sc_ps), SPC_MUT(17, K<T>::rd, -K<T>::rd) / K<T>::cp)
                                / (K<T>::cp * rho);                                    // spcpl.py:161
                if (OPT(z0m)) OPT(z0m)[col] = OPT(Z0M)[col];
                if (OPT(z0h)) OPT(z0h)[col] = OPT(Z0H)[col];
            }
        }
    }

    // ---- phase 2: LES levels (interpolate 5 fields, form the forcings) and index-map entries ------
    const SPC_DIVISOR(T) ddt(p.dt);
    for (int e = tid; e < nitems; e += BLK) {
        if (e < n2) {
            const int c = e / nL, l = e - c * nL;
            const int64_t col = col0 + c, o = col * pitchL + l;
            const T *const s = lds + (size_t)c * 6 * nG;
            const LesIn<T> in = (PRE && e == tid) ? pre2 : load_les<FwdP<T, FULL>, T>(p, l, o);
            const Br<T> b = bracket2(s, nG, p2G, in.h);
            T r[5];
#if SPC_K1_NF == 5
            {
                T f0[5], f1[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    f0[k] = s[(k + 1) * nG + b.j0];
                    f1[k] = s[(k + 1) * nG + b.j1];
                }
                interp_fields<5>(b, f0, f1, r);
            }
#else
            // fields in groups of SPC_K1_NF: fewer slope divisions interleaved, fewer live registers, more waves per SIMD
#pragma unroll
            for (int k0 = 0; k0 < 5; k0 += SPC_K1_NF) {
                constexpr int G = SPC_K1_NF;
                T f0[G], f1[G], rr[G];
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const int kk = (k0 + k) < 5 ? (k0 + k) : 4;
                    f0[k] = s[(kk + 1) * nG + b.j0];
                    f1[k] = s[(kk + 1) * nG + b.j1];
                }
                interp_fields<G>(b, f0, f1, rr);
#pragma unroll
                for (int k = 0; k < G; ++k)
                    if (k0 + k < 5) r[k0 + k] = rr[k];
            }
#endif
            const T thl = r[0], qt = r[1], ql = r[2], u = r[3], v = r[4];               // spcpl.py:224-228
            stg<WT>(&p.f_u[o], ddt.div(p.factor * (u - SPC_MUT(2, in.vd, in.ud))));               // spcpl.py:328
            stg<WT>(&p.f_v[o], ddt.div(p.factor * (v - SPC_MUT(2, in.ud, in.vd))));               // spcpl.py:329
            stg<WT>(&p.f_thl[o], ddt.div(p.factor * (thl - in.thld)));         // spcpl.py:330
            stg<WT>(&p.f_qt[o], ddt.div(p.factor * (qt - in.qtd)));            // spcpl.py:331
            stg<WT>(&p.f_ql[o], ddt.div(p.factor * (ql - in.qld)));            // spcpl.py:333
            stg<WT>(&p.ql_ref[o], ql);                                                         // spcpl.py:347-348
            if constexpr (FULL) {
                if (OPT(u)) OPT(u)[o] = u;
                if (OPT(v)) OPT(v)[o] = v;
                if (OPT(thl)) OPT(thl)[o] = thl;
                if (OPT(qt)) OPT(qt)[o] = qt;
            }
        } else {                                                                      // fused K2, spcpl.py:764
            const int ei = e - n2, c = ei / nG, m = ei - c * nG;
            const int64_t col = col0 + c, gh = col * pitchGh;
            const T zgh = (PRE && e == tid) ? pre_zgh : ldg(&p.Zghalf[gh + (nG - 1 - m)]);
            const T zs = (PRE && e == tid) ? pre_zs : ldg(&p.Zghalf[gh + nG]);
            const T Zh_k = div_grav(zgh - zs);                                        // spcpl.py:197
            const T *const zh = d.shared_grid ? lzh : lzh + (size_t)c * nL;
            p.idx[col * pitchG + m] = SPC_MUT(5, ss_left_pos(zh, nL, Zh_k), ss_right(zh, nL, Zh_k));
        }
    }
    STAMP(4);

    // ---- half-level heights (optional output): spcpl.py:197 --------------------------------------
    if constexpr (FULL) {
        if (OPT(Zh)) {
            for (int e = tid; e < ncol * (nG + 1); e += BLK) {
                const int c = e / (nG + 1), k = e - c * (nG + 1);
                const int64_t gh = (col0 + c) * pitchGh;
                OPT(Zh)[gh + k] = div_grav(ldg(&p.Zghalf[gh + k]) - ldg(&p.Zghalf[gh + nG]));
            }
        }
    }
    STAMP(5);
#undef OPT
}

// =================================================================================================
// K2 standalone: splib/spcpl.py:26 / 764
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_cloud_idx(const DimsP d, const T *zh_, const T *Zh_, int32_t *idx)
{
    const int nG = d.nG, nL = d.nL, cb = d.cb, tid = threadIdx.x;
    const int64_t col0 = (int64_t)slab_index(d.xcd_remap) * cb;
    const int ncol = (int)((d.n_cols - col0) < cb ? (d.n_cols - col0) : cb);
    T *const lzh = reinterpret_cast<T *>(spc_smem);
    const int nz = d.shared_grid ? nL : ncol * nL;
    for (int e = tid; e < nz; e += BLOCK) {
        const int c = e / nL, l = e - c * nL;
        lzh[e] = d.shared_grid ? zh_[e] : zh_[(col0 + c) * d.pitchL + l];
    }
    __syncthreads();
    for (int e = tid; e < ncol * nG; e += BLOCK) {
        const int c = e / nG, m = e - c * nG;
        const int64_t col = col0 + c;
        const T *const zh = d.shared_grid ? lzh : lzh + (size_t)c * nL;
        idx[col * d.pitchG + m] = ss_right(zh, nL, Zh_[col * d.pitchGh + (nG - 1 - m)]);
    }
}

// =================================================================================================
// K3 backward: splib/spcpl.py:388-555, linear branch (468-478) + start_index (498) + tendencies
// (518-526) + masking (527-533).  LDS per column: t | qt | ql | ql_ice | u | v, each [nL]; then
// Zf [nG]; then h: [nL] when the LES grid is shared, else [CB x nL].
// =================================================================================================
template <typename T> struct GcmIn {
    T tt, sh, ql, qi, u, v, a, a_d;
};

template <typename T> __device__ __forceinline__ GcmIn<T> load_gcm(const BwdP<T> &p, int64_t g, int64_t g_rev)
{
    GcmIn<T> r;
    r.tt = ldg(&p.Tm[g]); r.sh = ldg(&p.SH[g]); r.ql = ldg(&p.QL[g]); r.qi = ldg(&p.QI[g]); r.u = ldg(&p.U[g]); r.v = ldg(&p.V[g]); r.a = ldg(&p.A[g]);
    r.a_d = ldg(&p.A_prof[g_rev]);                                                           // spcpl.py:404
    return r;
}

// PRE: the GCM-side inputs of a thread's first output level are loaded in the prologue (one memory round trip for a
// single-round launch).  Without it K3 needs 60 instead of 78 VGPRs (8 waves per SIMD instead of 6): +4-7 % at 2-4 k
// columns, nothing from 16 k on where K3 has saturated (profiles/r02_k3_pre_ab.log) -- used between 1 025 and 25 000 columns.
template <typename T, int NG, int NL, int WT, int BLK = BLOCK, bool PRE = true> __global__ __launch_bounds__(BLK, SPC_K3_WAVES) void k_backward(const BwdP<T> p)
{
    const DimsP &d = p.d;
    const int nG = NG ? NG : d.nG, nL = NL ? NL : d.nL, cb = d.cb, tid = threadIdx.x;
    const int64_t pitchG = NG ? NG : d.pitchG, pitchGh = NG ? NG + 1 : d.pitchGh, pitchL = NL ? NL : d.pitchL;
    const int p2L = NL ? cfloor_pow2(NL ? NL : 1) : d.p2L;
    const int64_t col0 = (int64_t)slab_index(d.xcd_remap) * cb;
    const int ncol = (int)((d.n_cols - col0) < cb ? (d.n_cols - col0) : cb);
    const size_t per_col = (size_t)6 * nL + nG;
    T *const lds = reinterpret_cast<T *>(spc_smem);
    T *const lh = lds + (size_t)cb * per_col;
    const int n1 = ncol * nG;
    STAMP(0);

    // Loads are issued in the order the data is NEEDED (memory returns roughly in issue order and
    // s_waitcnt vmcnt counts in issue order): first this thread's first staging element of every LES array
    // and of Zf, which the LDS writes in front of the barrier wait for; then the GCM-side inputs of its
    // first output level, which land while the staging completes.
    struct Stage { T t, qt, ql, qi, u, v, h; };
    auto load_stage = [&](int64_t o) {
        Stage r;
        r.t = ldg(&p.t_d[o]); r.qt = ldg(&p.qt_d[o]); r.ql = ldg(&p.ql_d[o]); r.qi = ldg(&p.ql_ice_d[o]);
        r.u = ldg(&p.u_d[o]); r.v = ldg(&p.v_d[o]);
        r.h = d.shared_grid ? T(0) : ldg(&p.zf[o]);
        return r;
    };
    auto load_zf = [&](int64_t col, int64_t g) {
        return p.Zf ? p.Zf[g] : div_grav(ldg(&p.Zgfull[g]) - ldg(&p.Zghalf[col * pitchGh + nG]));   // spcpl.py:198
    };
    const int n2 = ncol * nL;
    Stage st0 = {};
    T zf0 = T(0), hs0 = T(0);
    if (tid < n2) {
        const int c = tid / nL, l = tid - c * nL;
        st0 = load_stage((col0 + c) * pitchL + l);
    }
    if (tid < n1) {
        const int c = tid / nG, k = tid - c * nG;
        zf0 = load_zf(col0 + c, (col0 + c) * pitchG + k);
    }
    if (d.shared_grid && tid < nL) hs0 = ldg(&p.zf[tid]);
    GcmIn<T> pre = {};
    if (PRE && tid < n1) {
        const int c = tid / nG, k = tid - c * nG;
        const int64_t cg = (col0 + c) * pitchG;
        pre = load_gcm(p, cg + k, SPC_MUT(9, (col0 + ((c ^ 1) < ncol ? (c ^ 1) : c)) * pitchG, cg) + (nG - 1 - k));
    }
    STAMP(1);

    // UF work items per thread and loop round, all their loads issued before the first is used: 1 for double (the form of
    // rounds 1-4), 2 for float -- a 4-byte access puts half the bytes in flight.  Measured (profiles/r05_f32_ab.log): K3<float>
    // -5 % at config 3 with the quotients through fp64; the same scheme in K1<float> was SLOWER (86 against 78-80 us:
    // 63 instead of 48 VGPRs and 8 scalar spills) and is not used there.  -DSPC_F32_UNROLL=1: the A/B build.
    constexpr int UF = sizeof(T) == 4 ? SPC_F32_UNROLL : 1;
    for (int e0 = tid; e0 < n2; e0 += UF * BLK) {
        Stage st[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int e = e0 + u * BLK;
            if (e < n2) {
                const int c = e / nL, l = e - c * nL;
                st[u] = (e == tid) ? st0 : load_stage((col0 + c) * pitchL + l);
            }
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int e = e0 + u * BLK;
            if (e < n2) {
                const int c = e / nL, l = e - c * nL;
                T *const s = lds + (size_t)c * per_col + l;
                s[0] = st[u].t;
                s[nL] = st[u].qt;
                s[2 * nL] = st[u].ql;
                s[3 * nL] = st[u].qi;
                s[4 * nL] = st[u].u;
                s[5 * nL] = st[u].v;
                if (!d.shared_grid) lh[e] = st[u].h;
            }
        }
    }
    if (d.shared_grid)
        for (int e = tid; e < nL; e += BLK) lh[e] = (e == tid) ? hs0 : ldg(&p.zf[e]);
    for (int e = tid; e < n1; e += BLK) {
        const int c = e / nG, k = e - c * nG;
        const int64_t col = col0 + c;
        lds[(size_t)c * per_col + 6 * nL + k] = (e == tid) ? zf0 : load_zf(col, col * pitchG + k);
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);

    const SPC_DIVISOR(T) ddt(p.dt);
    auto gcm_item = [&](int e, const GcmIn<T> &in) {
        const int c = e / nG, k = e - c * nG;
        const int64_t col = col0 + c, cg = col * pitchG, g = cg + k;
        const T *const s = lds + (size_t)c * per_col;
        const T *const h = d.shared_grid ? lh : lh + (size_t)c * nL;
        const T *const Zf = s + 6 * nL;
        const T x = Zf[k];
        const int start_index = ss_left_neg(Zf, nG, h[nL - 1]);                        // spcpl.py:498
        // (the branch-light interp_fields<7> form was measured here too: no gain at 1024 columns and -12 % at
        //  >= 35k columns, because interleaving 7 division chains costs 118 VGPRs and a third of the occupancy)
        const Bracket<T> b = bracket(h, nL, p2L, x);
        T t_i, qt_i, ql_i, qlw_i, qli_i, u_i, v_i;
        if (b.mode == 0) {
            const int j = b.j;
            const T ql0 = s[2 * nL + j], ql1 = s[2 * nL + j + 1], qi0 = s[3 * nL + j], qi1 = s[3 * nL + j + 1];
            const SPC_DIVISOR(T) dx(b.x1 - b.x0);
            t_i = lerp_np(x, b.x0, b.x1, s[j], s[j + 1], dx);                          // spcpl.py:471
            qt_i = lerp_np(x, b.x0, b.x1, s[nL + j], s[nL + j + 1], dx);               // spcpl.py:472
            ql_i = lerp_np(x, b.x0, b.x1, ql0, ql1, dx);                               // spcpl.py:473
            qlw_i = lerp_np(x, b.x0, b.x1, ql0 - qi0, ql1 - qi1, dx);                  // spcpl.py:402,474
            qli_i = lerp_np(x, b.x0, b.x1, qi0, qi1, dx);                              // spcpl.py:475
            u_i = lerp_np(x, b.x0, b.x1, s[4 * nL + j], s[4 * nL + j + 1], dx);        // spcpl.py:476
            v_i = lerp_np(x, b.x0, b.x1, s[5 * nL + j], s[5 * nL + j + 1], dx);        // spcpl.py:477
        } else if (b.mode == 1) {
            const int j = b.j;
            t_i = s[j];
            qt_i = s[nL + j];
            ql_i = s[2 * nL + j];
            qli_i = s[3 * nL + j];
            qlw_i = ql_i - qli_i;
            u_i = s[4 * nL + j];
            v_i = s[5 * nL + j];
        } else {
            t_i = qt_i = ql_i = qlw_i = qli_i = u_i = v_i = x;
        }
        T f_T = ddt.div(p.factor * SPC_MUT(24, in.tt - t_i, t_i - in.tt));               // spcpl.py:518
        T f_SH = ddt.div(p.factor * (SPC_MUT(13, qt_i, qt_i - ql_i) - in.sh));                            // spcpl.py:519
        T f_QL = ddt.div(p.factor * (SPC_MUT(3, ql_i, qlw_i) - in.ql));                                    // spcpl.py:520
        T f_QI = ddt.div(p.factor * (qli_i - in.qi));                                    // spcpl.py:521
        T f_U = ddt.div(p.factor * (SPC_MUT(22, v_i, u_i) - in.u));                      // spcpl.py:524
        T f_V = ddt.div(p.factor * (v_i - in.v));                                        // spcpl.py:525
        T f_A = ddt.div(p.factor * (in.a_d - in.a));                                     // spcpl.py:526
        if (SPC_MUT(4, k <= start_index, k < start_index)) {  // `f[0:start_index] *= 0` (spcpl.py:527-533): -x -> -0, NaN stays NaN
            const T zero = T(0);
            f_T *= zero; f_SH *= zero; f_QL *= zero; f_QI *= zero; f_U *= zero; f_V *= zero; f_A *= zero;
        }
        stg<WT>(&p.f_T[g], f_T);
        stg<WT>(&p.f_SH[g], f_SH);
        stg<WT>(&p.f_QL[g], f_QL);
        stg<WT>(&p.f_QI[g], f_QI);
        stg<WT>(&p.f_U[g], f_U);
        stg<WT>(&p.f_V[g], f_V);
        stg<WT>(&p.f_A[g], f_A);
        if (p.start_index && k == 0) p.start_index[col] = start_index;
    };
    for (int e0 = tid; e0 < n1; e0 += UF * BLK) {
        GcmIn<T> in[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int e = e0 + u * BLK;
            if (e < n1) {
                const int c = e / nG, k = e - c * nG;
                const int64_t cg = (col0 + c) * pitchG;
                in[u] = (PRE && e == tid) ? pre
                                          : load_gcm(p, cg + k, SPC_MUT(9, (col0 + ((c ^ 1) < ncol ? (c ^ 1) : c)) * pitchG, cg) + (nG - 1 - k));
            }
        }
#pragma unroll
        for (int u = 0; u < UF; ++u)
            if (e0 + u * BLK < n1) gcm_item(e0 + u * BLK, in[u]);
    }
    STAMP(4);
    STAMP(5);
}

#include "spc_f32v.hpp"
#include "spc_vnudge.hpp"
#include "spc_vnudge2.hpp"

// =================================================================================================
// K4 backward, conservative coarsening: splib/spcpl.py:479-489 -> sputils.interp_c / integral
// (splib/sputils.py:94-189).  Same tendencies / masking as K3, but each GCM level receives the
// rho-weighted mean of the piecewise-constant LES profile over [Zh[i+1], Zh[i]] instead of a linear
// interpolation.  Kernel: spc_k4.hpp (one thread per (level, field)).
// =================================================================================================
// first k in [1, n-1] with !(z[k] < a), minus 1: the `while z[i+1] < a: i += 1` scan of integral()
// (splib/sputils.py:122-127) for ascending z
template <typename T> __device__ __forceinline__ int scan_cell(const T *z, int n, T a)
{
    int lo = 1, hi = n - 1;   // the scan cannot pass n-2 because a <= z[n-1] was checked
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (z[mid] < a) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

// ---- searches on LDS rows: fixed trip count, no bounds check -------------------------------------------------------
// The three searches of this file -- numpy.interp's bracket (count of xp[i] <= x), numpy.searchsorted (count of entries
// in front of the insertion point) and integral()'s cell scan (count of z[k] < a) -- are prefix counts over an ascending
// row.  A staged row is PADDED WITH NaN up to 2 p2 entries (p2 = the largest power of two <= its length): the greedy
// power-of-two descent of upper_count() then needs no `t <= n` test, because no predicate used here advances on a NaN
// (searchsorted with a NaN key is the one exception and clamps), and with the trip count a template argument (SL =
// log2 p2 + 1) every probe is one ds_read with an immediate offset + compare + select: 3 VALU instructions per step
// instead of 8-12 (round 4: the first K7 generation was bound by VALU issue, 100-150 instructions per output,
// profiles/r04_k7_counters.log).  SL = 0: the same descent with p2 at run time; SL = -1: nothing staged (rows beyond
// the LDS), the operators fall back to the loops on global memory.
__host__ __device__ inline int su_pad(int p2) { return 2 * p2 + 2; }      // entries of a padded row (+2: rows off each other's banks)

// the descent carries the ADDRESS of the first entry not counted (row + count), not the count: a step is then one
// ds_read at [address + immediate], one add, one compare and one select -- no index-to-address shift per probe
template <int SL, typename T, typename Pred> __device__ __forceinline__ const T *su_seek(const T *row, int p2, const Pred &adv)
{
    const T *p = row;
    if constexpr (SL > 0) {
#pragma unroll
        for (int s = 1 << (SL - 1); s > 0; s >>= 1) {
            const T *const nx = p + s;
            p = adv(p[s - 1]) ? nx : p;
        }
    } else {
        for (int s = p2; s > 0; s >>= 1) {
            const T *const nx = p + s;
            p = adv(p[s - 1]) ? nx : p;
        }
    }
    return p;
}

template <int SL, typename T, typename Pred> __device__ __forceinline__ int su_count(const T *row, int p2, const Pred &adv)
{
    return (int)(((unsigned)(size_t)su_seek<SL>(row, p2, adv) - (unsigned)(size_t)row) / (unsigned)sizeof(T));   // 32-bit: LDS addresses
}

#include "spc_k4.hpp"
#include "spc_sputils.hpp"

// =================================================================================================
// K5 diagnostics: splib/spcpl.py:176, 197-198, 214-215 (GCM levels); 402, 408-409 (LES levels)
// LDS per column (only when pf/t requested): Zf reversed | Pf reversed, each [nG].
// =================================================================================================
// Round 5 (round-4 verdict, weak 14): like K1 / K3 the kernel is instantiated for the compile-time geometries (NG / NL != 0:
// contiguous columns, flat-index divisions by constants, unrolled search) and with write-through stores (WT) for launches
// that leave <= 32 MiB behind; the LES-side inputs of an output are loaded BEFORE its search, the interpolation runs the
// branch-light form of K1 (bracket2 / interp_fields), every access goes through ldg / stg.
template <typename T, int NG, int NL, int WT> __global__ __launch_bounds__(BLOCK) void k_diag(const DiagP<T> p)
{
    const DimsP &d = p.d;
    const int nG = NG ? NG : d.nG, nL = NL ? NL : d.nL, cb = d.cb, tid = threadIdx.x;
    const int64_t pitchG = NG ? NG : d.pitchG, pitchGh = NG ? NG + 1 : d.pitchGh, pitchL = NL ? NL : d.pitchL;
    const int p2G = NG ? cfloor_pow2(NG ? NG : 1) : d.p2G;
    const int64_t col0 = (int64_t)slab_index(d.xcd_remap) * cb;
    const int ncol = (int)((d.n_cols - col0) < cb ? (d.n_cols - col0) : cb);
    T *const lds = reinterpret_cast<T *>(spc_smem);
    const T cc = K<T>::rv / K<T>::rd - T(1);                                           // spcpl.py:175
    const bool les = p.zf && (p.pf || p.t || p.ql_water);
    for (int e = tid; e < ncol * nG; e += BLOCK) {
        const int c = e / nG, k = e - c * nG;
        const int64_t col = col0 + c, g = col * pitchG + k;
        const T zs = ldg(&p.Zghalf[col * pitchGh + nG]);
        const T tt = ldg(&p.Tm[g]), sh = ldg(&p.SH[g]), ql = ldg(&p.QL[g]), qi = ldg(&p.QI[g]), pf = ldg(&p.Pf[g]), zg = ldg(&p.Zgfull[g]);
        const T zf_k = div_grav(zg - zs);
        if (les) {
            T *const s = lds + (size_t)c * 2 * nG + (nG - 1 - k);
            s[0] = zf_k;
            s[nG] = pf;
        }
        if (p.Tv) stg<WT>(&p.Tv[g], tt * (T(1) + cc * sh - SPC_MUT(25, -(ql + qi), (ql + qi))));   // spcpl.py:176
        if (p.QT) stg<WT>(&p.QT[g], sh + ql + SPC_MUT(26, T(0), qi));
        if (p.Zf) stg<WT>(&p.Zf[g], zf_k);
        if (p.THL) stg<WT>(&p.THL[g], (tt - div_cp(K<T>::rlv * (ql + qi))) * spc_pow(div_pref0(pf), (-K<T>::rd) / K<T>::cp));
    }
    if (p.Zh) {
        for (int e = tid; e < ncol * (nG + 1); e += BLOCK) {
            const int c = e / (nG + 1), k = e - c * (nG + 1);
            const int64_t gh = (col0 + c) * pitchGh;
            stg<WT>(&p.Zh[gh + k], div_grav(ldg(&p.Zghalf[gh + k]) - ldg(&p.Zghalf[gh + SPC_MUT(27, nG - 1, nG)])));   // spcpl.py:197
        }
    }
    if (!les) return;                                                                  // (uniform: no barrier is skipped by part of a workgroup)
    __syncthreads();
    for (int e = tid; e < ncol * nL; e += BLOCK) {
        const int c = e / nL, l = e - c * nL;
        const int64_t o = (col0 + c) * pitchL + l;
        const T *const s = lds + (size_t)c * 2 * nG;
        const T h = d.shared_grid ? ldg(&p.zf[l]) : ldg(&p.zf[o]);
        const T thl = p.t ? ldg(&p.thl_d[o]) : T(0);
        const T qld = (p.t || p.ql_water) ? ldg(&p.ql_d[o]) : T(0);
        const T qid = p.ql_water ? ldg(&p.ql_ice_d[o]) : T(0);
        const Br<T> b = bracket2(s, nG, p2G, h);
        const T f0[1] = {s[nG + b.j0]}, f1[1] = {s[nG + b.j1]};
        T r[1];
        interp_fields<1>(b, f0, f1, r);
        const T pf = r[0];                                                             // spcpl.py:408
        if (p.pf) stg<WT>(&p.pf[o], pf);
        if (p.t)                                                                       // spcpl.py:409
            stg<WT>(&p.t[o], thl * spc_pow(div_pref0(pf), SPC_MUT(10, -K<T>::rd, K<T>::rd) / K<T>::cp) + div_cp(K<T>::rlv * qld));
        if (p.ql_water) stg<WT>(&p.ql_water[o], qld - qid);                            // spcpl.py:402
    }
}

// spcpl.convert_surface_fluxes for columns WITHOUT an LES (extra output columns, spcpl.py:112-115):
// per-column scalars only.  Ph_s = Phalf[:, nG] (surface pressure), T_s = T[:, nG-1] (lowest level).
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_surface(int64_t n, const T *Ph_s, const T *T_s, const T *QLflux, const T *QIflux,
                                                   const T *SHflux, const T *TSflux, T *wthl, T *wqt)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const T ps = Ph_s[i];
        const T rho = ps / (K<T>::rd * T_s[i]);                                        // spcpl.py:153
        wqt[i] = -(QLflux[i] + SPC_MUT(16, T(0), QIflux[i]) + SHflux[i]) / rho;       // spcpl.py:159
        wthl[i] = -TSflux[i] * spc_pow(div_pref0(ps), (-K<T>::rd) / K<T>::cp) / (K<T>::cp * rho);   // spcpl.py:161
    }
}

// ---- host side --------------------------------------------------------------------------------
int floor_pow2(int n)
{
    int p = 1;
    while (p * 2 <= n) p *= 2;
    return p;
}

int validate(const spc_dims *d)
{
    if (!d) return fail(SPC_ERR_INVALID_ARGUMENT, "%sdims is NULL");
    if (d->n_cols < 0) return fail(SPC_ERR_INVALID_ARGUMENT, "%sn_cols = %lld < 0", "", (long long)d->n_cols);
    if (d->nG < 1 || d->nL < 1)
        return fail(SPC_ERR_INVALID_ARGUMENT, "%slevel counts must be >= 1 (nG=%lld nL=%lld)", "", d->nG, d->nL);
    if (d->pitchG < d->nG || d->pitchGh < d->nG + 1 || d->pitchL < d->nL)
        return fail(SPC_ERR_INVALID_ARGUMENT, "%spitch smaller than the level count");
    if (d->n_cols > (int64_t)INT32_MAX * 8)
        return fail(SPC_ERR_INVALID_ARGUMENT, "%sn_cols too large for one launch");
    if (d->cols_per_block < 0 || d->cols_per_block > 64)
        return fail(SPC_ERR_INVALID_ARGUMENT, "%scols_per_block out of range 0..64");
    return SPC_OK;
}

int geometry_id(const spc_dims *d);

// LDS elements per column / per block for each pass (pass 0 fwd, 1 bwd, 2 idx, 3 diag, 4 conservative bwd)
void lds_elems(const spc_dims *d, int pass, bool with_idx, size_t *per_col, size_t *fixed, size_t esize = 8)
{
    const size_t nG = d->nG, nL = d->nL;
    const bool sh = d->les_grid_shared != 0;
    switch (pass) {
    case 0: *per_col = 6 * nG + ((with_idx && !sh) ? nL : 0); *fixed = (with_idx && sh) ? nL : 0; break;
    case 1: *per_col = 6 * nL + nG + (sh ? 0 : nL); *fixed = sh ? nL : 0; break;
    case 4:
        if (geometry_id(d) != 0) {   // k_backward_cons3: A[8][nL+1] | Zh[nG+1] | cell[nG] | start index; zh rows NaN-padded; dz when shared
            const size_t zrow = (size_t)su_pad(floor_pow2((int)nL - 1));
            *per_col = 8 * (nL + 1) + (nG + 1) + (nG * 4 + esize - 1) / esize + 1 + (sh ? 0 : zrow);
            *fixed = sh ? zrow + (nL - 1) : 0;
            break;
        }
        *per_col = 7 * (nL + 1) + 8 * nG + 2 + (nG * 4 + esize - 1) / esize + 1 + (sh ? 0 : nL); *fixed = sh ? nL : 0; break;   // k_backward_cons2; + zf[nL-1] per column
    case 2: *per_col = sh ? 0 : nL; *fixed = sh ? nL : 0; break;
    default: *per_col = 2 * nG; *fixed = 0; break;
    }
}

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Experiment switch: dynamic LDS of a launch raised to at least SPC_LDS_MIN_KIB_K1 / _K3 KiB (<= 64), which caps the workgroups
// resident per CU (160 KiB / that) without touching the kernel: fewer resident workgroups live shorter at the same chip-wide
// rate, and a launch pays one workgroup lifetime for fill + drain (profiles/r05_residency.log).
size_t lds_floor(size_t smem, const char *name)
{
    const int kib = env_int(name, 0);
    const size_t want = (size_t)(kib > 64 ? 64 : kib) * 1024;
    return want > smem ? want : smem;
}

// Compute units of the CURRENT device (hipDeviceAttributeMultiprocessorCount; cached per device ordinal): what the residency
// rules below count rounds of workgroups against.  An MI355X in SPX mode has 256; a CPX / DPX partition or another SKU
// has fewer, and rule 1 of pick_cb would silently pick the wrong slab there (round-4 verdict, weak 10).  SPC_CUS=<n>
// overrides (tests walk the heuristics at 32 ... 256 CUs without a GPU); without a device: 256.
int device_cus()
{
    const int forced = env_int("SPC_CUS", 0);
    if (forced > 0) return forced;
    thread_local int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 256;
    }
    if (dev >= 0 && dev < 64 && cache[dev]) return cache[dev];
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        n = 256;
    }
    if (dev >= 0 && dev < 64) cache[dev] = n;
    return n;
}

// Resident workgroups per CU for `kernel` with `smem` bytes of dynamic LDS (occupancy API, cached).
// Without a device (CPU-side ABI tests) falls back to min(4, 160 KiB / smem).
template <typename KernelT> int blocks_per_cu(KernelT kernel, size_t smem)
{
    thread_local std::unordered_map<uint64_t, int> cache;
    const uint64_t key = (uint64_t)(uintptr_t)kernel * 1000003u + smem;
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, BLOCK, smem) != hipSuccess || nb <= 0) {
        (void)hipGetLastError();
        const size_t by_lds = smem ? (size_t)(160 * 1024) / smem : 8;
        nb = (int)(by_lds < 4 ? by_lds : 4);
    }
    if (getenv("SPC_DEBUG_OCC")) fprintf(stderr, "spc: occupancy query: %zu B of dynamic LDS -> %d workgroups per CU\n", smem, nb);
    if (nb > 8) nb = 8;
    cache[key] = nb;
    return nb;
}

// Columns per workgroup (CB).
//  1. If some CB in {1,2,4} lets the WHOLE grid be resident at once (n_cols/CB <= CUs of the device x resident
//     workgroups per CU at that CB's LDS footprint), take the smallest such CB: a single round of
//     workgroups, maximum parallelism per column (measured: 2048 columns run 13.3 us at CB=2 but 19-20 us
//     at CB=1, which needs two rounds).
//  2. Otherwise (throughput regime) the CB with the most resident workgroups per CU among those that
//     still give >= 2 full rounds of workgroups (rounds de-synchronise the load / compute / store
//     phases), ties to the larger one (longer coalesced slabs): K1 -> 8, K3 -> 2 at 91<->160.
template <typename KernelT> int pick_cb(const spc_dims *d, int pass, bool with_idx, size_t esize, KernelT kernel)
{
    size_t per_col, fixed;
    lds_elems(d, pass, with_idx, &per_col, &fixed, esize);
    int cb = d->cols_per_block;
    if (cb > 0) {
        while (cb > 1 && (per_col * cb + fixed) * esize > (size_t)MAX_LDS_BYTES) --cb;
        return cb;
    }
    int nb[4] = {0, 0, 0, 0};
    const int64_t cus = device_cus();
    for (int i = 0; i < 4; ++i) {
        cb = 1 << i;
        const size_t smem = (per_col * cb + fixed) * esize;
        if (cb > 1 && smem > (size_t)MAX_LDS_BYTES) continue;
        nb[i] = blocks_per_cu(kernel, smem);
        if (cb <= 4 && (d->n_cols + cb - 1) / cb <= cus * nb[i]) return cb;   // rule 1
    }
    // K4 is bound by dependent LDS reads, not by memory: what counts is resident COLUMNS (2 x 4 workgroups beat 1 x 5
    // by 11 % at config 3, 4 x 2 loses 60 %: profiles/r03_k4_forms.log)
    if (pass == 4 && nb[1] * 2 > nb[0] && (d->n_cols + 1) / 2 >= 2 * cus * nb[1]) return 2;
    int best = 1, best_nb = -1;
    // (K3<float>: slabs of more than two columns lose -- 70.9 us at two, 77.5 at four, 88.5 at eight columns per workgroup at
    //  config 3, profiles/r05_f32_cbs.log -- where the residency tie of the 4-byte footprint would pick four)
    for (int i = (pass == 1 && esize == 4) ? 1 : 3; i >= 0; --i) {                      // rule 2
        cb = 1 << i;
        const int64_t rounds_x_cus = nb[i] ? (d->n_cols + cb - 1) / cb / nb[i] : 0;   // rounds of workgroups x CUs
        const bool enough = rounds_x_cus >= (cb == 8 ? 8 : 2) * cus;   // measured: 8-column slabs pay off from ~8 rounds
        if (nb[i] > best_nb && (enough || i == 0)) { best_nb = nb[i]; best = cb; }
    }
    return best;
}

// Launches that write no more than the aggregate L2 (32 MiB) store write-through: otherwise all of it is
// still dirty when the kernel ends and the end-of-kernel release has to flush it (measured: WT wins up
// to ~4096 columns, loses beyond ~16k).  SPC_FORCE_WT=0/1 overrides (A/B runs).
int wt_forced()      // SPC_FORCE_WT: 0 / 1, else -1
{
    static const int forced = [] { const char *e = getenv("SPC_FORCE_WT"); const int v = e ? atoi(e) : -1; return v == 0 || v == 1 ? v : -1; }();
    return forced;
}

int small_batch(int64_t bytes_written, int limit_mib = 32)
{
    if (wt_forced() >= 0) return wt_forced();
    return bytes_written <= (int64_t)limit_mib * 1024 * 1024 ? 1 : 0;
}

// 0 = generic; 1..3 = compile-time geometries with contiguous columns (see k_forward)
int geometry_id(const spc_dims *d)
{
    if (d->pitchG != d->nG || d->pitchGh != d->nG + 1 || d->pitchL != d->nL) return 0;
    if (d->nG == 91 && d->nL == 160) return 1;
    if (d->nG == 137 && d->nL == 512) return 2;
    if (d->nG == 19 && d->nL == 160) return 3;
    return 0;
}

DimsP make_dims(const spc_dims *d, int cb)
{
    DimsP p;
    p.n_cols = d->n_cols; p.pitchG = d->pitchG; p.pitchGh = d->pitchGh; p.pitchL = d->pitchL;
    p.nG = d->nG; p.nL = d->nL; p.cb = cb; p.p2G = floor_pow2(d->nG); p.p2L = floor_pow2(d->nL);
    p.shared_grid = d->les_grid_shared != 0;
    p.xcd_remap = cb <= 2;   // measured: +4-6 % for 1-2 column slabs (K3), -1.5 % for 8-column slabs (K1)
    return p;
}

// Dynamic LDS above the 64 KiB default needs the per-function opt-in (tall columns: nL > ~1300 in K3).
template <typename KernelT> int ensure_lds(KernelT kernel, size_t smem, const char *what)
{
    if (smem <= (size_t)MAX_LDS_BYTES) return SPC_OK;
    if (smem > (size_t)HARD_LDS_BYTES)
        return fail(SPC_ERR_UNSUPPORTED, "%s needs %lld B of LDS per workgroup (gfx950 has %lld)", what, (long long)smem, HARD_LDS_BYTES);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
        (void)hipGetLastError();
        return fail(SPC_ERR_UNSUPPORTED, "%s: cannot raise the dynamic LDS limit to %lld B", what, (long long)smem);
    }
    return SPC_OK;
}

int launch_status(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return SPC_OK;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return SPC_ERR_LAUNCH;
}

#define REQUIRE(ptr, name) \
    if (!(ptr)) return fail(SPC_ERR_INVALID_ARGUMENT, "required pointer %s is NULL", name)

// Small batches run ONE round of workgroups and are bound by latency, not bandwidth: there fewer, larger workgroups
// win.  2 / 4 columns per workgroup of 512 / 1024 threads (still one work item per thread, so the per-thread chain is
// unchanged) cover <= 1024 columns with <= 256 workgroups -- one per CU -- and the grid is dispatched in a half / a
// quarter of the time.  Measured (tools/ab_blocks.sh, pre-heated, profiles/r02_ab_blocks.log): K1 8.5 -> 7.1 us and
// K3 8.1 -> 7.8 us at 1024 columns, K1 6.4 -> 5.4 us at 512; slower from 1536 columns on.  Returns the columns per
// workgroup (workgroup = 256 x that) or 0 = the 256-thread path.  SPC_SMALL_BLOCK=0 disables it (A/B).
int small_block(const spc_dims *d, int items_per_col)
{
    const int64_t cus = device_cus();          // (MI355X: 256 -> the 257 ... 1024 columns of the measurements above)
    if (d->cols_per_block != 0 || items_per_col > BLOCK || d->n_cols <= cus || d->n_cols > 4 * cus) return 0;
    const int sb = env_int("SPC_SMALL_BLOCK", 1);                    // 0: off; 2 / 4: that many columns per workgroup (A/B)
    if (!sb) return 0;
    if (sb == 2 || sb == 4) return sb;
    return d->n_cols <= 2 * cus ? 2 : 4;
}

// ---- launch choice ---------------------------------------------------------------------------------------------
// WHICH instantiation runs, and in what shape, is decided in ONE place per pass (choose_fwd / choose_bwd); the
// launchers and spc_describe_launch (include/spc.h) both call it, so a test can walk the dispatch table and assert
// that every instantiation it can reach has been bit-checked (tests/test_dispatch_gpu.py).
struct Choice {
    const char *kernel;
    int elem, full, idx, geo, wt, blk, pre, cb;
    int vec = 0;      // the float kernels with 8-byte accesses (spc_f32v.hpp)
    unsigned grid;
    size_t smem;
};

constexpr int GEO_NG[4] = {0, 91, 137, 19}, GEO_NL[4] = {0, 160, 512, 160};

template <typename T> using KLean = void (*)(const FwdP<T, false>);
template <typename T> using KFull = void (*)(const FwdP<T, true>);
template <typename T> using KBwd = void (*)(const BwdP<T>);

#define SPC_FWD_ROW(FULL_, WT_, BLK_, PRE_)                                                                          \
    {k_forward<T, FULL_, 0, 0, WT_, BLK_, PRE_>, k_forward<T, FULL_, 91, 160, WT_, BLK_, PRE_>,                      \
     k_forward<T, FULL_, 137, 512, WT_, BLK_, PRE_>, k_forward<T, FULL_, 19, 160, WT_, BLK_, PRE_>}
// 512- / 1024-thread workgroups (small_block): one round of <= 1024 columns of <= 256 work items each, i.e. always
// write-through and never 137 <-> 512 (649 work items per column): only those instantiations exist
#define SPC_FWD_ROW_BIG(BLK_)                                                                                        \
    {k_forward<T, false, 0, 0, 1, BLK_, true>, k_forward<T, false, 91, 160, 1, BLK_, true>, nullptr,                 \
     k_forward<T, false, 19, 160, 1, BLK_, true>}

// lean forward kernel of (geometry, write-through, workgroup size, prologue prefetch); nullptr = not instantiated
template <typename T> KLean<T> fwd_lean_kernel(int geo, int wt, int blk, int pre)
{
    static const KLean<T> k256[2][2][4] = {{SPC_FWD_ROW(false, 0, BLOCK, false), SPC_FWD_ROW(false, 1, BLOCK, false)},
                                           {SPC_FWD_ROW(false, 0, BLOCK, true), SPC_FWD_ROW(false, 1, BLOCK, true)}};
    static const KLean<T> k512[4] = SPC_FWD_ROW_BIG(512), k1024[4] = SPC_FWD_ROW_BIG(1024);
    if (blk == BLOCK) return k256[pre][wt][geo];
    if (!wt || !pre) return nullptr;
    return blk == 512 ? k512[geo] : (blk == 1024 ? k1024[geo] : nullptr);
}

// the FULL variant (optional outputs, surface coupling: convert_profiles() and cplsurf=True, off the default path of
// splib.py:67) exists with plain stores only: write-through is worth ~5 % on launches of <= 4 k columns and would double
// the number of its instantiations
template <typename T> KFull<T> fwd_full_kernel(int geo, int pre)
{
    static const KFull<T> k[2][4] = {SPC_FWD_ROW(true, 0, BLOCK, false), SPC_FWD_ROW(true, 0, BLOCK, true)};
    return k[pre][geo];
}
#undef SPC_FWD_ROW
#undef SPC_FWD_ROW_BIG

template <typename T> int choose_fwd(const spc_dims *d, bool with_idx, bool full, Choice *c)
{
    c->kernel = "k_forward"; c->elem = (int)sizeof(T); c->full = full; c->idx = with_idx;
    c->geo = geometry_id(d);
    c->wt = full ? 0 : small_batch(d->n_cols * (int64_t)((6 * d->nL + 1) * sizeof(T) + (with_idx ? d->nG * 4 : 0)));
    // (137 <-> 512 never qualifies for small_block: 649 work items per column; nor does a launch whose write-through
    //  stores were switched off for an A/B run)
    const int sb = (full || !c->wt) ? 0 : small_block(d, d->nL + (with_idx ? d->nG : 0));
    // single-round launches keep the prologue prefetch (k_forward's PRE); SPC_K1_PRE=0/1 forces it off / on (A/B)
    const int pre_env = env_int("SPC_K1_PRE", -1);
    c->pre = (sb || (pre_env >= 0 ? pre_env != 0 : d->n_cols <= 4 * (int64_t)device_cus())) ? 1 : 0;   // measured (256 CUs): PRE = false wins from 1100 columns
    c->blk = sb ? BLOCK * sb : BLOCK;
    if (full)
        c->cb = pick_cb(d, 0, with_idx, sizeof(T), fwd_full_kernel<T>(c->geo, c->pre));
    else
        c->cb = sb ? sb : pick_cb(d, 0, with_idx, sizeof(T), fwd_lean_kernel<T>(c->geo, 0, BLOCK, c->pre));
    size_t per_col, fixed;
    lds_elems(d, 0, with_idx, &per_col, &fixed);
    c->smem = (per_col * c->cb + fixed) * sizeof(T);
    c->grid = (unsigned)((d->n_cols + c->cb - 1) / c->cb);
    // float, compile-time geometry, lean, multi-round, an even slab: 8-byte accesses (spc_f32v.hpp; SPC_F32_VEC=0: A/B, tests)
    c->vec = std::is_same<T, float>::value && c->geo != 0 && !full && !sb && !c->pre && c->cb % 2 == 0 && env_int("SPC_F32_VEC", 1);
    return SPC_OK;
}

// K1 of the float variant with 8-byte accesses, by (geometry, write-through)
inline KLean<float> fwd_vec_kernel(int geo, int wt)
{
    static const KLean<float> k[2][4] = {{nullptr, k_forward_f32v<91, 160, 0>, k_forward_f32v<137, 512, 0>, k_forward_f32v<19, 160, 0>},
                                         {nullptr, k_forward_f32v<91, 160, 1>, k_forward_f32v<137, 512, 1>, k_forward_f32v<19, 160, 1>}};
    return k[wt ? 1 : 0][geo];
}
inline bool aligned8(std::initializer_list<const void *> ptrs)
{
    for (const void *q : ptrs) if ((uintptr_t)q & 7u) return false;
    return true;
}

template <typename T> int forward_impl(const spc_dims *d, const spc_forward_args *a, void *stream)
{
    int rc = validate(d);
    if (rc) return rc;
    if (!a) return fail(SPC_ERR_INVALID_ARGUMENT, "%sargs is NULL");
    if (d->n_cols == 0) return SPC_OK;
    REQUIRE(a->U, "U"); REQUIRE(a->V, "V"); REQUIRE(a->T, "T"); REQUIRE(a->SH, "SH"); REQUIRE(a->QL, "QL");
    REQUIRE(a->QI, "QI"); REQUIRE(a->Pf, "Pf"); REQUIRE(a->Ph, "Ph"); REQUIRE(a->Zgfull, "Zgfull");
    REQUIRE(a->Zghalf, "Zghalf"); REQUIRE(a->zf, "zf"); REQUIRE(a->u_d, "u_d"); REQUIRE(a->v_d, "v_d");
    REQUIRE(a->thl_d, "thl_d"); REQUIRE(a->qt_d, "qt_d"); REQUIRE(a->ql_d, "ql_d"); REQUIRE(a->ps_d, "ps_d");
    REQUIRE(a->f_u, "f_u"); REQUIRE(a->f_v, "f_v"); REQUIRE(a->f_thl, "f_thl"); REQUIRE(a->f_qt, "f_qt");
    REQUIRE(a->f_ql, "f_ql"); REQUIRE(a->ql_ref, "ql_ref"); REQUIRE(a->f_ps, "f_ps");
    if (a->idx && !a->zh) return fail(SPC_ERR_INVALID_ARGUMENT, "%sidx requested but zh is NULL");
    if (a->rainrate && (!a->rain || !a->rain_last))
        return fail(SPC_ERR_INVALID_ARGUMENT, "%srainrate requested but rain / rain_last is NULL");
    if (a->wthl || a->wqt) {
        if (!a->wthl || !a->wqt || !a->QLflux || !a->QIflux || !a->SHflux || !a->TSflux)
            return fail(SPC_ERR_INVALID_ARGUMENT, "%ssurface coupling needs wthl, wqt and QLflux,QIflux,SHflux,TSflux");
        if ((a->z0m && !a->Z0M) || (a->z0h && !a->Z0H))
            return fail(SPC_ERR_INVALID_ARGUMENT, "%sz0m/z0h requested but Z0M/Z0H is NULL");
    }
    const bool with_idx = a->idx != nullptr;
    const bool full = a->u || a->v || a->thl || a->qt || a->ps || a->Zf || a->Zh || a->rainrate || a->wthl;
    Choice c;
    if ((rc = choose_fwd<T>(d, with_idx, full, &c))) return rc;
#define CP(f) p.f = (const T *)a->f
#define OP(f) p.f = (T *)a->f
#define COP(f) p.o.f = (const T *)a->f
#define OOP(f) p.o.f = (T *)a->f
    auto fill = [&](auto &p) {
        p.d = make_dims(d, c.cb);
        CP(U); CP(V); p.Tm = (const T *)a->T; CP(SH); CP(QL); CP(QI); CP(Pf); CP(Ph); CP(Zgfull); CP(Zghalf);
        CP(zf); CP(zh); CP(u_d); CP(v_d); CP(thl_d); CP(qt_d); CP(ql_d); CP(ps_d);
        p.factor = (T)a->factor; p.dt = (T)a->dt;
        OP(f_u); OP(f_v); OP(f_thl); OP(f_qt); OP(f_ql); OP(ql_ref); OP(f_ps); p.idx = a->idx;
    };
    if (full) {
        const KFull<T> kern = fwd_full_kernel<T>(c.geo, c.pre);
        FwdP<T, true> p;
        fill(p);
        COP(rain); COP(rain_last); OOP(u); OOP(v); OOP(thl); OOP(qt); OOP(ps); OOP(Zf); OOP(Zh); OOP(rainrate);
        COP(Z0M); COP(Z0H); COP(QLflux); COP(QIflux); COP(SHflux); COP(TSflux); OOP(z0m); OOP(z0h); OOP(wthl); OOP(wqt);
        if ((rc = ensure_lds(kern, c.smem, "forward"))) return rc;
        hipLaunchKernelGGL(kern, dim3(c.grid), dim3(c.blk), c.smem, (hipStream_t)stream, p);
    } else {
        KLean<T> kern = fwd_lean_kernel<T>(c.geo, c.wt, c.blk, c.pre);
        if constexpr (std::is_same<T, float>::value) {
            if (c.vec && aligned8({a->U, a->V, a->T, a->SH, a->QL, a->QI, a->Pf, a->Zgfull, a->zf, a->u_d, a->v_d, a->thl_d, a->qt_d, a->ql_d,
                                   a->f_u, a->f_v, a->f_thl, a->f_qt, a->f_ql, a->ql_ref}))
                kern = fwd_vec_kernel(c.geo, c.wt);
        }
        if (!kern) return fail(SPC_ERR_UNSUPPORTED, "%sforward: no kernel instantiated for this launch choice (internal)");
        FwdP<T, false> p;
        fill(p);
        if ((rc = ensure_lds(kern, c.smem, "forward"))) return rc;
        hipLaunchKernelGGL(kern, dim3(c.grid), dim3(c.blk), lds_floor(c.smem, "SPC_LDS_MIN_KIB_K1"), (hipStream_t)stream, p);
    }
    return launch_status("k_forward");
}

template <typename T>
int cloud_idx_impl(const spc_dims *d, const void *zh, const void *Zh, int32_t *idx, void *stream)
{
    int rc = validate(d);
    if (rc) return rc;
    if (d->n_cols == 0) return SPC_OK;
    REQUIRE(zh, "zh"); REQUIRE(Zh, "Zh"); REQUIRE(idx, "idx");
    const int cb = pick_cb(d, 2, true, sizeof(T), k_cloud_idx<T>);
    size_t per_col, fixed;
    lds_elems(d, 2, true, &per_col, &fixed);
    const size_t smem = (per_col * cb + fixed) * sizeof(T);
    if ((rc = ensure_lds(k_cloud_idx<T>, smem, "cloud_indices"))) return rc;
    const unsigned grid = (unsigned)((d->n_cols + cb - 1) / cb);
    hipLaunchKernelGGL(k_cloud_idx<T>, dim3(grid), dim3(BLOCK), smem, (hipStream_t)stream, make_dims(d, cb),
                       (const T *)zh, (const T *)Zh, idx);
    return launch_status("k_cloud_idx");
}

#define SPC_BWD_ROW(WT_, BLK_, PRE_)                                                                                 \
    {k_backward<T, 0, 0, WT_, BLK_, PRE_>, k_backward<T, 91, 160, WT_, BLK_, PRE_>, k_backward<T, 137, 512, WT_, BLK_, PRE_>, \
     k_backward<T, 19, 160, WT_, BLK_, PRE_>}
#define SPC_BWD_ROW_BIG(BLK_)                                                                                        \
    {k_backward<T, 0, 0, 1, BLK_, true>, k_backward<T, 91, 160, 1, BLK_, true>, nullptr, k_backward<T, 19, 160, 1, BLK_, true>}

// K3 of (geometry, write-through, workgroup size, prologue prefetch); nullptr = not instantiated (see fwd_lean_kernel)
template <typename T> KBwd<T> bwd_kernel(int geo, int wt, int blk, int pre)
{
    static const KBwd<T> k256[2][2][4] = {{SPC_BWD_ROW(0, BLOCK, false), SPC_BWD_ROW(1, BLOCK, false)},
                                          {SPC_BWD_ROW(0, BLOCK, true), SPC_BWD_ROW(1, BLOCK, true)}};
    static const KBwd<T> k512[4] = SPC_BWD_ROW_BIG(512), k1024[4] = SPC_BWD_ROW_BIG(1024);
    if (blk == BLOCK) return k256[pre][wt][geo];
    if (!wt || !pre) return nullptr;
    return blk == 512 ? k512[geo] : (blk == 1024 ? k1024[geo] : nullptr);
}
#undef SPC_BWD_ROW
#undef SPC_BWD_ROW_BIG

// K4 of a geometry; run-time geometry (geo 0): numpy's pairwise recursion unrolled to the depth nL needs (spc_k4.hpp) --
// pd = 1, 2, 3 for LES grids of up to 248 / 488 / 968 levels, else the explicit-stack form
template <typename T> KBwd<T> cons_kernel(int geo, int pd, int cb)
{
    // compile-time geometries: the third form (spc_k4.hpp: products per cell, padded scans, layer means stashed in registers)
    static const KBwd<T> k3[4][2] = {{nullptr, nullptr},
                                     {k_backward_cons3<T, 91, 160, 1>, k_backward_cons3<T, 91, 160, 2>},
                                     {k_backward_cons3<T, 137, 512, 1>, k_backward_cons3<T, 137, 512, 2>},
                                     {k_backward_cons3<T, 19, 160, 1>, k_backward_cons3<T, 19, 160, 2>}};
    if (geo != 0) return k3[geo][cb >= 2 ? 1 : 0];
    if constexpr (std::is_same<T, double>::value) {      // (the float twin, config 5's tolerance sweep, keeps the stack form)
        static const KBwd<T> kd[3] = {k_backward_cons2<T, 0, 0, 1>, k_backward_cons2<T, 0, 0, 2>, k_backward_cons2<T, 0, 0, 3>};
        if (pd >= 1 && pd <= 3) return kd[pd - 1];
    }
    return k_backward_cons2<T, 0, 0, -1>;
}

// columns per workgroup of the third-form K4: one while the whole grid is resident at once (a single round: the most
// parallelism per column), else two when that keeps more COLUMNS resident per CU (K4's rate follows them, spc_k4.hpp)
template <typename T> int pick_cb_cons3(const spc_dims *d, int geo)
{
    size_t per_col, fixed;
    lds_elems(d, 4, false, &per_col, &fixed, sizeof(T));
    const size_t smem1 = (per_col + fixed) * sizeof(T), smem2 = (2 * per_col + fixed) * sizeof(T);
    const bool two_fits = smem2 <= (size_t)MAX_LDS_BYTES;
    if (d->cols_per_block > 0) return (d->cols_per_block >= 2 && two_fits) ? 2 : 1;
    const int nb1 = blocks_per_cu(cons_kernel<T>(geo, 0, 1), smem1);
    if (d->n_cols <= (int64_t)device_cus() * nb1 || !two_fits) return 1;
    const int nb2 = blocks_per_cu(cons_kernel<T>(geo, 0, 2), smem2);
    return nb2 * 2 > nb1 ? 2 : 1;
}

// depth of numpy's pairwise recursion over at most nL elements (<= 8192: one chunk); -1: use the explicit stack
int cons_depth(int nL)
{
    if (nL > 1024) return -1;
    // vn_pw_depth(n) = max over 129 .. n of the (triple-recursive) depth of n: a running maximum, filled ONCE (it was
    // re-evaluated three times per K4 launch: tens of thousands of calls for a grid of ~1000 levels)
    static const struct Tab { signed char d[1025]; Tab() { int m = 0; for (int n = 0; n <= 1024; ++n) { if (n >= 129) { const int dn = vn_pw_depth_of(n); if (dn > m) m = dn; } d[n] = (signed char)m; } } } tab;
    const int d = nL < 0 ? 0 : tab.d[nL];
    return d < 1 ? 1 : (d <= 3 ? d : -1);
}

template <typename T> int choose_bwd(const spc_dims *d, bool cons, Choice *c)
{
    c->kernel = cons ? "k_backward_cons2" : "k_backward"; c->elem = (int)sizeof(T); c->full = cons; c->idx = 0;
    c->geo = geometry_id(d);
    // K3's stores stop gaining from write-through earlier than K1's: at 2 560 columns (13 MB written) it still wins 5-7 %, at
    // 3 072 ... 6 144 it loses 2-5 % (profiles/r04_write_through_sweep.log); K4 loses 10 % at config 3 (182 MB)
    c->wt = cons ? 0 : small_batch(d->n_cols * (int64_t)(7 * d->nG * sizeof(T)), 14);
    const int sb = (cons || !c->wt) ? 0 : small_block(d, d->nL > d->nG ? d->nL : d->nG);
    const int pre_env = env_int("SPC_K3_PRE", -1);        // SPC_K3_PRE=0/1 forces the prologue prefetch off / on (A/B)
    // PRE = false (8 waves per SIMD) pays between one round of workgroups and saturation: 1 025 ... 25 000 columns
    const int64_t cus = device_cus();      // the measured bounds 1 025 ... 25 000 are 4 ... ~98 columns per CU of the 256
    c->pre = (cons || sb || (pre_env >= 0 ? pre_env != 0 : (d->n_cols <= 4 * cus || d->n_cols * 256 > 25000 * cus))) ? 1 : 0;
    c->blk = sb ? BLOCK * sb : BLOCK;
    c->cb = sb ? sb : (cons ? (c->geo ? pick_cb_cons3<T>(d, c->geo) : pick_cb(d, 4, false, sizeof(T), cons_kernel<T>(0, cons_depth(d->nL), 0)))
                            : pick_cb(d, 1, false, sizeof(T), bwd_kernel<T>(c->geo, 0, BLOCK, c->pre)));
    size_t per_col, fixed;
    lds_elems(d, cons ? 4 : 1, false, &per_col, &fixed, sizeof(T));
    c->smem = (per_col * c->cb + fixed) * sizeof(T);
    c->grid = (unsigned)((d->n_cols + c->cb - 1) / c->cb);
    return SPC_OK;
}

template <typename T> int backward_impl(const spc_dims *d, const spc_backward_args *a, void *stream)
{
    int rc = validate(d);
    if (rc) return rc;
    if (!a) return fail(SPC_ERR_INVALID_ARGUMENT, "%sargs is NULL");
    if (d->n_cols == 0) return SPC_OK;
    REQUIRE(a->T, "T"); REQUIRE(a->SH, "SH"); REQUIRE(a->QL, "QL"); REQUIRE(a->QI, "QI"); REQUIRE(a->U, "U");
    REQUIRE(a->V, "V"); REQUIRE(a->A, "A"); REQUIRE(a->zf, "zf"); REQUIRE(a->t_d, "t_d"); REQUIRE(a->qt_d, "qt_d");
    REQUIRE(a->ql_d, "ql_d"); REQUIRE(a->ql_ice_d, "ql_ice_d"); REQUIRE(a->u_d, "u_d"); REQUIRE(a->v_d, "v_d");
    REQUIRE(a->A_prof, "A_prof"); REQUIRE(a->f_T, "f_T"); REQUIRE(a->f_SH, "f_SH"); REQUIRE(a->f_QL, "f_QL");
    REQUIRE(a->f_QI, "f_QI"); REQUIRE(a->f_U, "f_U"); REQUIRE(a->f_V, "f_V"); REQUIRE(a->f_A, "f_A");
    if (!a->Zf && (!a->Zgfull || !a->Zghalf))
        return fail(SPC_ERR_INVALID_ARGUMENT, "%sneither Zf nor (Zgfull, Zghalf) given");
    const bool cons = a->conservative != 0;
    if (cons) {
        REQUIRE(a->zh, "zh (conservative)"); REQUIRE(a->rhobf_d, "rhobf_d (conservative)");
        if (!a->Zh && !a->Zghalf) return fail(SPC_ERR_INVALID_ARGUMENT, "%sconservative: neither Zh nor Zghalf given");
        if (d->nL < 2) return fail(SPC_ERR_INVALID_ARGUMENT, "%sconservative coarsening needs nL >= 2");
    }
    Choice c;
    if ((rc = choose_bwd<T>(d, cons, &c))) return rc;
    const KBwd<T> kern = cons ? cons_kernel<T>(c.geo, cons_depth(d->nL), c.cb) : bwd_kernel<T>(c.geo, c.wt, c.blk, c.pre);
    if (!kern) return fail(SPC_ERR_UNSUPPORTED, "%sbackward: no kernel instantiated for this launch choice (internal)");
    if ((rc = ensure_lds(kern, c.smem, cons ? "backward (conservative)" : "backward"))) return rc;
    BwdP<T> p;
    p.d = make_dims(d, c.cb);
    p.Tm = (const T *)a->T; CP(SH); CP(QL); CP(QI); CP(U); CP(V); CP(A); CP(Zf); CP(Zgfull); CP(Zghalf); CP(zf);
    CP(t_d); CP(qt_d); CP(ql_d); CP(ql_ice_d); CP(u_d); CP(v_d); CP(A_prof); CP(zh); CP(Zh); CP(rhobf_d);
    p.factor = (T)a->factor; p.dt = (T)a->dt;
    OP(f_T); OP(f_SH); OP(f_QL); OP(f_QI); OP(f_U); OP(f_V); OP(f_A); p.start_index = a->start_index;
    hipLaunchKernelGGL(kern, dim3(c.grid), dim3(c.blk), cons ? c.smem : lds_floor(c.smem, "SPC_LDS_MIN_KIB_K3"), (hipStream_t)stream, p);
    return launch_status(cons ? "k_backward_cons" : "k_backward");
}

template <typename T> using KDiag = void (*)(const DiagP<T>);

// K5 of (geometry, write-through)
template <typename T> KDiag<T> diag_kernel(int geo, int wt)
{
    static const KDiag<T> k[2][4] = {{k_diag<T, 0, 0, 0>, k_diag<T, 91, 160, 0>, k_diag<T, 137, 512, 0>, k_diag<T, 19, 160, 0>},
                                     {k_diag<T, 0, 0, 1>, k_diag<T, 91, 160, 1>, k_diag<T, 137, 512, 1>, k_diag<T, 19, 160, 1>}};
    return k[wt ? 1 : 0][geo];
}

// launch choice of K5 (as choose_fwd / choose_bwd: ONE place, also behind spc_describe_launch); `a` may be NULL (describe: every
// output assumed)
template <typename T> int choose_diag(const spc_dims *d, const spc_diagnostics_args *a, Choice *c)
{
    c->kernel = "k_diag"; c->elem = (int)sizeof(T); c->full = c->idx = 0; c->pre = 0; c->blk = BLOCK;
    c->geo = geometry_id(d);
    const int64_t nGw = !a ? 4 : (a->Tv != nullptr) + (a->THL != nullptr) + (a->QT != nullptr) + (a->Zf != nullptr);
    const int64_t nLw = !a ? 3 : (a->pf != nullptr) + (a->t != nullptr) + (a->ql_water != nullptr);
    const int64_t elems = nGw * d->nG + ((!a || a->Zh) ? d->nG + 1 : 0) + ((!a || a->zf) ? nLw * d->nL : 0);
    c->wt = small_batch(d->n_cols * elems * (int64_t)sizeof(T));
    c->cb = pick_cb(d, 3, false, sizeof(T), diag_kernel<T>(c->geo, 0));
    size_t per_col, fixed;
    lds_elems(d, 3, false, &per_col, &fixed);
    c->smem = (per_col * c->cb + fixed) * sizeof(T);
    c->grid = (unsigned)((d->n_cols + c->cb - 1) / c->cb);
    return SPC_OK;
}

template <typename T> int diag_impl(const spc_dims *d, const spc_diagnostics_args *a, void *stream)
{
    int rc = validate(d);
    if (rc) return rc;
    if (!a) return fail(SPC_ERR_INVALID_ARGUMENT, "%sargs is NULL");
    if (d->n_cols == 0) return SPC_OK;
    REQUIRE(a->T, "T"); REQUIRE(a->SH, "SH"); REQUIRE(a->QL, "QL"); REQUIRE(a->QI, "QI"); REQUIRE(a->Pf, "Pf");
    REQUIRE(a->Zgfull, "Zgfull"); REQUIRE(a->Zghalf, "Zghalf");
    if ((a->pf || a->t || a->ql_water) && !a->zf) return fail(SPC_ERR_INVALID_ARGUMENT, "%sLES diagnostics need zf");
    if (a->t && (!a->thl_d || !a->ql_d)) return fail(SPC_ERR_INVALID_ARGUMENT, "%st needs thl_d and ql_d");
    if (a->ql_water && (!a->ql_d || !a->ql_ice_d)) return fail(SPC_ERR_INVALID_ARGUMENT, "%sql_water needs ql_d and ql_ice_d");
    Choice c;
    if ((rc = choose_diag<T>(d, a, &c))) return rc;
    const KDiag<T> kern = diag_kernel<T>(c.geo, c.wt);
    const int cb = c.cb;
    const size_t smem = c.smem;
    if ((rc = ensure_lds(kern, smem, "diagnostics"))) return rc;
    DiagP<T> p;
    p.d = make_dims(d, cb);
    p.Tm = (const T *)a->T; CP(SH); CP(QL); CP(QI); CP(Pf); CP(Zgfull); CP(Zghalf); CP(zf); CP(thl_d); CP(ql_d); CP(ql_ice_d);
    OP(Tv); OP(THL); OP(QT); OP(Zf); OP(Zh); OP(pf); OP(t); OP(ql_water);
    const unsigned grid = (unsigned)((d->n_cols + cb - 1) / cb);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), smem, (hipStream_t)stream, p);
    return launch_status("k_diag");
}
#undef CP
#undef OP
#undef COP
#undef OOP

template <typename T>
int surface_impl(int64_t n, const void *Ph_s, const void *T_s, const void *QLflux, const void *QIflux,
                        const void *SHflux, const void *TSflux, void *wthl, void *wqt, void *stream)
{
    if (n < 0) return fail(SPC_ERR_INVALID_ARGUMENT, "%ssurface_fluxes: n < 0");
    if (n == 0) return SPC_OK;
    if (!Ph_s || !T_s || !QLflux || !QIflux || !SHflux || !TSflux || !wthl || !wqt)
        return fail(SPC_ERR_INVALID_ARGUMENT, "%ssurface_fluxes: NULL pointer");
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK < 2048 ? (n + BLOCK - 1) / BLOCK : 2048);
    hipLaunchKernelGGL(k_surface<T>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, n, (const T *)Ph_s, (const T *)T_s,
                       (const T *)QLflux, (const T *)QIflux, (const T *)SHflux, (const T *)TSflux, (T *)wthl, (T *)wqt);
    return launch_status("k_surface");
}

template <typename T> int describe_impl(const spc_dims *d, int pass, int flags, Choice *c)
{
    switch (pass) {
    case 0: return choose_fwd<T>(d, (flags & 1) != 0, (flags & 2) != 0, c);
    case 1: return choose_bwd<T>(d, false, c);
    case 4: return choose_bwd<T>(d, true, c);
    case 3: return choose_diag<T>(d, nullptr, c);
    case 2: {
        c->kernel = "k_cloud_idx"; c->elem = (int)sizeof(T); c->full = c->idx = c->geo = c->wt = c->pre = 0;
        c->blk = BLOCK;
        c->cb = pick_cb(d, 2, true, sizeof(T), k_cloud_idx<T>);
        size_t per_col, fixed;
        lds_elems(d, pass, true, &per_col, &fixed);
        c->smem = (per_col * c->cb + fixed) * sizeof(T);
        c->grid = (unsigned)((d->n_cols + c->cb - 1) / c->cb);
        return SPC_OK;
    }
    default: return fail(SPC_ERR_INVALID_ARGUMENT, "%spass must be 0..4");
    }
}

#include "spc_sputils_host.hpp"

}  // namespace

extern "C" {

int spc_forward_f64(const spc_dims *d, const spc_forward_args *a, void *s) { return forward_impl<double>(d, a, s); }
int spc_forward_f32(const spc_dims *d, const spc_forward_args *a, void *s) { return forward_impl<float>(d, a, s); }
int spc_cloud_indices_f64(const spc_dims *d, const void *zh, const void *Zh, int32_t *idx, void *s)
{
    return cloud_idx_impl<double>(d, zh, Zh, idx, s);
}
int spc_cloud_indices_f32(const spc_dims *d, const void *zh, const void *Zh, int32_t *idx, void *s)
{
    return cloud_idx_impl<float>(d, zh, Zh, idx, s);
}
int spc_backward_f64(const spc_dims *d, const spc_backward_args *a, void *s) { return backward_impl<double>(d, a, s); }
int spc_backward_f32(const spc_dims *d, const spc_backward_args *a, void *s) { return backward_impl<float>(d, a, s); }
int spc_diagnostics_f64(const spc_dims *d, const spc_diagnostics_args *a, void *s) { return diag_impl<double>(d, a, s); }
int spc_diagnostics_f32(const spc_dims *d, const spc_diagnostics_args *a, void *s) { return diag_impl<float>(d, a, s); }

#ifdef SPC_STAMPS
int spc_debug_set_stamps(void *buf)  // diagnostic build only
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : SPC_ERR_LAUNCH;
}
#endif

int spc_exner_f64(int64_t n, const void *p, void *out, int32_t inv, void *s) { return exner_impl<double>(n, p, out, inv, s); }
int spc_exner_f32(int64_t n, const void *p, void *out, int32_t inv, void *s) { return exner_impl<float>(n, p, out, inv, s); }
int spc_interp_f64(const spc_interp_args *a, void *s) { return interp_impl<double>(a, s); }
int spc_interp_f32(const spc_interp_args *a, void *s) { return interp_impl<float>(a, s); }
int spc_searchsorted_f64(const spc_searchsorted_args *a, void *s) { return searchsorted_impl<double>(a, s); }
int spc_searchsorted_f32(const spc_searchsorted_args *a, void *s) { return searchsorted_impl<float>(a, s); }
int spc_interp_c_f64(const spc_interp_c_args *a, void *s) { return interp_c_impl<double>(a, s); }
int spc_interp_c_f32(const spc_interp_c_args *a, void *s) { return interp_c_impl<float>(a, s); }
int spc_rms_f64(int64_t nr, int64_t n, int64_t pitch, const void *a, void *out, void *s) { return rms_impl<double>(nr, n, pitch, a, out, s); }
int spc_rms_f32(int64_t nr, int64_t n, int64_t pitch, const void *a, void *out, void *s) { return rms_impl<float>(nr, n, pitch, a, out, s); }

int spc_surface_fluxes_f64(int64_t n, const void *Ph_s, const void *T_s, const void *QLflux, const void *QIflux,
                           const void *SHflux, const void *TSflux, void *wthl, void *wqt, void *stream)
{
    return surface_impl<double>(n, Ph_s, T_s, QLflux, QIflux, SHflux, TSflux, wthl, wqt, stream);
}

int spc_surface_fluxes_f32(int64_t n, const void *Ph_s, const void *T_s, const void *QLflux, const void *QIflux,
                           const void *SHflux, const void *TSflux, void *wthl, void *wqt, void *stream)
{
    return surface_impl<float>(n, Ph_s, T_s, QLflux, QIflux, SHflux, TSflux, wthl, wqt, stream);
}

// leaves of numpy's pairwise recursion over n elements (n <= 8192): the host-side twin of vn_build_tree's count
static int vn_count_leaves(int n)
{
    if (n <= 128) return 1;
    int n2 = n / 2;
    n2 -= n2 % 8;
    return vn_count_leaves(n2) + vn_count_leaves(n - n2);
}

// LDS bytes of the plane-resident solver (spc_vnudge2.hpp) with `t` levels per workgroup
static size_t vn_lds_need(int nij, int nleaf_max, int t)
{
    return (size_t)t * vn2_plane(nij) * 16 + (size_t)t * nleaf_max * 8 + VN2_THREADS * 12;
}

// Planes that fit the LDS (KT levels x nij x 16 B <= 150 KiB, KT a power of two <= 16; 64 x 64 planes: KT = 2) are
// solved there; larger planes (> ~9 000 points) are streamed from the transposed workspace.  Returns whether the LDS
// path applies, the levels per workgroup and the leaf count of numpy's pairwise tree.
static bool vn_lds_fit(int nij, int *kt_, int *log2_kt_, int *nleaf_max_)
{
    int kt = 16, log2_kt = 4;
    const int cn = nij < 8192 ? nij : 8192, nleaf_max = nij > 8192 ? VN_MAXLEAF : vn_count_leaves(cn);
    while (kt > 1 && vn_lds_need(nij, nleaf_max, kt) > (size_t)VN2_MAX_LDS) { kt >>= 1; --log2_kt; }
    *kt_ = kt; *log2_kt_ = log2_kt; *nleaf_max_ = nleaf_max;
    return vn_lds_need(nij, nleaf_max, kt) <= (size_t)VN2_MAX_LDS;
}

int64_t spc_vnudge_workspace_bytes(int64_t n_cols, int32_t itot, int32_t jtot, int32_t ktot)
{
    if (n_cols < 0 || itot < 1 || jtot < 1 || ktot < 1 || (int64_t)itot * jtot > INT32_MAX / 2)
        return fail(SPC_ERR_INVALID_ARGUMENT, "%svariability_nudge: bad extents");
    return n_cols * 2 * (int64_t)itot * jtot * ktot * 8;
}

int spc_variability_nudge_f64(const spc_vnudge_args *a, void *stream)
{
    if (!a) return fail(SPC_ERR_INVALID_ARGUMENT, "%sargs is NULL");
    if (a->n_cols < 0 || a->itot < 1 || a->jtot < 1 || a->ktot < 1 || (int64_t)a->itot * a->jtot > INT32_MAX / 2)
        return fail(SPC_ERR_INVALID_ARGUMENT, "%svariability_nudge: bad extents");
    if (a->n_cols == 0) return SPC_OK;
    REQUIRE(a->qt, "qt"); REQUIRE(a->qsat, "qsat"); REQUIRE(a->R, "R"); REQUIRE(a->ql_av, "ql_av"); REQUIRE(a->qt_av, "qt_av");
    REQUIRE(a->ql_ref, "ql_ref"); REQUIRE(a->beta, "beta"); REQUIRE(a->a_add, "a_add"); REQUIRE(a->qt_std, "qt_std");
    REQUIRE(a->status, "status");
    if (a->constantT) { REQUIRE(a->thl, "thl (constantT)"); REQUIRE(a->ql, "ql (constantT)"); REQUIRE(a->presf, "presf (constantT)"); }
    if (a->n_cols > 32767) return fail(SPC_ERR_UNSUPPORTED, "%svariability_nudge: more than 32767 columns per launch");
    VnP p;
    p.n_cols = a->n_cols; p.nij = a->itot * a->jtot; p.ktot = a->ktot; p.constantT = a->constantT; p.pad = 0;
    p.qsat = (const double *)a->qsat; p.R = (const double *)a->R; p.ql_av = (const double *)a->ql_av; p.qt_av = (const double *)a->qt_av;
    p.presf = (const double *)a->presf; p.ql_ref = (const double *)a->ql_ref; p.ql = (const double *)a->ql;
    p.qt = (double *)a->qt; p.thl = (double *)a->thl; p.beta = (double *)a->beta; p.a_add = (double *)a->a_add;
    p.qt_std = (double *)a->qt_std; p.status = a->status;
    // Where the planes live while the root finder runs: in the CU's LDS when KT levels' planes fit (KT x nij x 16 B <= 150
    // KiB: up to ~9 000 points; 64 x 64 planes: KT = 2), else -- 128 x 128 and up -- in the caller's transposed workspace,
    // one workgroup per level streaming its contiguous planes (k_vnudge_solve<true>; SPC_VN_GLOBAL=1 forces it: A/B, tests).
    int kt, log2_kt, nleaf_max;
    const bool fits = vn_lds_fit(p.nij, &kt, &log2_kt, &nleaf_max);
    const int64_t work_need = a->n_cols * 2 * (int64_t)p.nij * a->ktot * 8;
    const bool have_work = a->work && a->work_bytes >= work_need && env_int("SPC_VN_TRANSPOSE", 1);
    const bool global = have_work && (!fits || env_int("SPC_VN_GLOBAL", 0));
    if (!fits && !have_work)
        return fail(SPC_ERR_INVALID_ARGUMENT, "%svariability_nudge: planes of %lld points do not fit the LDS: `work` of "
                    "spc_vnudge_workspace_bytes() bytes is required", "", (long long)p.nij);
    {
        auto lds_need = [&](int t) { return global ? (size_t)t * nleaf_max * 8 + VN2_THREADS * 12 : vn_lds_need(p.nij, nleaf_max, t); };
        // many workgroups (more than two rounds of one per CU): half the levels and half the threads per workgroup where
        // that lets TWO workgroups share a CU's LDS -- one's barriers and serial steps overlap the other's sums
        int nthreads = VN2_THREADS;
        if (global) { kt = 1; log2_kt = 0; }
        const int pair = env_int("SPC_VN_PAIR", 1);       // 0 never, 1 by workgroup count, 2 always (tests)
        if (!global && kt > 1 && lds_need(kt / 2) <= (size_t)(78 * 1024) &&
            (pair == 2 || (pair == 1 && a->n_cols * (int64_t)((a->ktot + kt - 1) / kt) > 512))) {
            kt >>= 1; --log2_kt; nthreads = VN2_THREADS / 2;
        }
        Vn2P q = {};
        q.p = p; q.kt = kt; q.log2_kt = log2_kt; q.nleaf_max = nleaf_max;
        for (int shape = 0; shape < 2; ++shape) {
            unsigned char ready[VN_MAXLEAF];
            vn_build_tree(shape == 0 ? 8192 : (p.nij % 8192 ? p.nij % 8192 : 8192), q.tab.lo[shape], q.tab.n[shape], q.tab.pl[shape],
                          q.tab.pr[shape], &q.tab.nleaf[shape]);
            q.tab.nround[shape] = vn_build_rounds(q.tab.nleaf[shape], q.tab.pl[shape], q.tab.pr[shape], q.tab.rnd[shape], ready);
            q.tab.balanced[shape] = env_int("SPC_VN_TREE_SHFL", 1) ? vn_tree_balanced(q.tab.nleaf[shape], q.tab.pl[shape], q.tab.pr[shape], q.tab.rnd[shape]) : 0;
        }
        q.work = nullptr;
        if (have_work) {
            hipLaunchKernelGGL(k_vnudge_transpose, dim3((unsigned)((p.nij + 63) / 64), (unsigned)((a->ktot + 15) / 16), (unsigned)(a->n_cols * 2)),
                               dim3(256), 0, (hipStream_t)stream, p, (double *)a->work);
            int rct = launch_status("k_vnudge_transpose");
            if (rct) return rct;
            q.work = (const double *)a->work;
        }
        q.tiles = (a->ktot + kt - 1) / kt;
        q.tg = 16 / kt;                                       // tiles that share the 128-B lines of 16 levels
        q.gpc = (q.tiles + q.tg - 1) / q.tg;
        q.groups = a->n_cols * q.gpc;
        const size_t smem = lds_need(kt);
        // the noise plane in registers (k_vnudge_solve<false, true>): planes of one chunk with one leaf per 8-lane group
        const bool rcache = !global && p.nij <= 8192 && q.tab.nleaf[1] <= ((nthreads >> log2_kt) >> 3) && env_int("SPC_VN_RCACHE", 1);
        int rc = global ? ensure_lds(k_vnudge_solve<true>, smem, "variability_nudge")
                        : (rcache ? ensure_lds(k_vnudge_solve<false, true>, smem, "variability_nudge") : ensure_lds(k_vnudge_solve<false>, smem, "variability_nudge"));
        if (rc) return rc;
        const int64_t nblk = (q.groups + 7) / 8 * 8 * q.tg;
        if (nblk > INT32_MAX) return fail(SPC_ERR_UNSUPPORTED, "%svariability_nudge: too many workgroups");
        if (global)
            hipLaunchKernelGGL(k_vnudge_solve<true>, dim3((unsigned)nblk), dim3(nthreads), smem, (hipStream_t)stream, q);
        else if (rcache)
            hipLaunchKernelGGL((k_vnudge_solve<false, true>), dim3((unsigned)nblk), dim3(nthreads), smem, (hipStream_t)stream, q);
        else
            hipLaunchKernelGGL(k_vnudge_solve<false>, dim3((unsigned)nblk), dim3(nthreads), smem, (hipStream_t)stream, q);
        rc = launch_status("k_vnudge_solve");
        if (rc) return rc;
        // the update (elementwise, wide) and qt.std (ordered sums, 16 levels per workgroup)
        const size_t usmem = (size_t)a->ktot * (3 * sizeof(double) + sizeof(int));
        if ((rc = ensure_lds(k_vnudge_update, usmem, "variability_nudge (update)"))) return rc;
        hipLaunchKernelGGL(k_vnudge_update, dim3((unsigned)((p.nij + VU_ROWS - 1) / VU_ROWS), (unsigned)a->n_cols), dim3(256), usmem,
                           (hipStream_t)stream, p);
        if ((rc = launch_status("k_vnudge_update"))) return rc;
        const dim3 sgrid((unsigned)((a->ktot + 15) / 16), (unsigned)a->n_cols);
        p.pad = env_int("SPC_VN_STD_DEBUG", 0);
        if ((int64_t)sgrid.x * sgrid.y <= 256 && env_int("SPC_VN_STD_ROWS", 512) == 512) {
            const size_t ssmem = (size_t)2 * 512 * 16 * sizeof(double);
            if ((rc = ensure_lds(k_vnudge_std<512>, ssmem, "variability_nudge (std)"))) return rc;
            hipLaunchKernelGGL(k_vnudge_std<512>, sgrid, dim3(VS_THREADS), ssmem, (hipStream_t)stream, p);
        } else {
            const size_t ssmem = (size_t)2 * 256 * 16 * sizeof(double);
            if ((rc = ensure_lds(k_vnudge_std<256>, ssmem, "variability_nudge (std)"))) return rc;
            hipLaunchKernelGGL(k_vnudge_std<256>, sgrid, dim3(VS_THREADS), ssmem, (hipStream_t)stream, p);
        }
        return launch_status("k_vnudge_std");
    }
}

int spc_abi_version(void) { return SPC_ABI_VERSION; }
const char *spc_last_error(void) { return g_err; }

int spc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int spc_describe_launch(const spc_dims *d, int pass, int flags, int elem_size, char *buf, int buflen)
{
    int rc = validate(d);
    if (rc) return rc;
    if (!buf || buflen < 1) return fail(SPC_ERR_INVALID_ARGUMENT, "%sdescribe_launch: no buffer");
    if (elem_size != 8 && elem_size != 4) return fail(SPC_ERR_INVALID_ARGUMENT, "%sdescribe_launch: elem_size must be 8 or 4");
    Choice c = {};
    rc = elem_size == 8 ? describe_impl<double>(d, pass, flags, &c) : describe_impl<float>(d, pass, flags, &c);
    if (rc) return rc;
    const char *ty = elem_size == 8 ? "f64" : "f32";
    char name[160];
    if (pass == 0 && c.vec)
        snprintf(name, sizeof(name), "k_forward_f32v<%d,%d,wt=%d>", GEO_NG[c.geo], GEO_NL[c.geo], c.wt);
    else if (pass == 0)
        snprintf(name, sizeof(name), "k_forward<%s,%s,%d,%d,wt=%d,blk=%d,pre=%d>", ty, c.full ? "full" : "lean", GEO_NG[c.geo], GEO_NL[c.geo],
                 c.wt, c.blk, c.pre);
    else if (pass == 1)
        snprintf(name, sizeof(name), "k_backward<%s,%d,%d,wt=%d,blk=%d,pre=%d>", ty, GEO_NG[c.geo], GEO_NL[c.geo], c.wt, c.blk, c.pre);
    else if (pass == 4)
        if (c.geo == 0)
            snprintf(name, sizeof(name), "k_backward_cons2<%s,0,0,pd=%d>", ty, elem_size == 8 ? cons_depth(d->nL) : -1);
        else
            snprintf(name, sizeof(name), "k_backward_cons3<%s,%d,%d,cb=%d>", ty, GEO_NG[c.geo], GEO_NL[c.geo], c.cb >= 2 ? 2 : 1);
    else if (pass == 3)
        snprintf(name, sizeof(name), "k_diag<%s,%d,%d,wt=%d>", ty, GEO_NG[c.geo], GEO_NL[c.geo], c.wt);
    else
        snprintf(name, sizeof(name), "%s<%s>", c.kernel, ty);
    return snprintf(buf, (size_t)buflen, "%s cb=%d grid=%u block=%d lds=%lld cus=%d", name, c.cb, c.grid, c.blk, (long long)c.smem, device_cus());
}

int spc_pick_cols_per_block(const spc_dims *d, int pass)
{
    int rc = validate(d);
    if (rc) return rc;
    Choice c = {};
    rc = describe_impl<double>(d, pass, 1, &c);     // forward: lean, with the fused index map
    return rc ? rc : c.cb;
}

}  // extern "C"
