// spc_hip.hip -- gfx950 (MI355X / CDNA4) kernels and C ABI of the batched SP coupling step.
//
// Replaces the serial per-column Python loops of the reference (splib/splib.py:317-323, 330-332)
// and the NumPy helpers they call (splib/spcpl.py:171-246, 299-385, 388-555, 761-764;
// splib/sputils.py:28-34, 82-91) with three launches over ALL columns:
//   K1 k_forward   GCM state -> LES-level profiles + nudging forcings (+ fused K2 index map,
//                  surface fluxes, rain rate)
//   K2 k_cloud_idx cloud-fraction level-index map (standalone form)
//   K3 k_backward  LES slab means -> GCM tendencies, masked above the LES top
//   K5 k_diag      spifs.nc diagnostics
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
#include <string.h>

#include "spc.h"

#pragma clang fp contract(off)

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_LDS_BYTES = 64 * 1024;  // default dynamic-LDS limit; keeps >= 2 workgroups per CU

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

__device__ __forceinline__ double spc_pow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float spc_pow(float x, float y) { return powf(x, y); }

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

// Count of xp[i] <= x for ascending xp (== numpy.interp's j + 1), fixed trip count: `p2` is the
// largest power of two <= n, so every lane runs the same floor(log2 n)+1 steps (no divergence).
template <typename T> __device__ __forceinline__ int upper_count(const T *xp, int n, int p2, T x)
{
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

// ---- kernel parameter blocks (typed copies of the C structs) ------------------------------------
struct DimsP {
    int64_t n_cols, pitchG, pitchGh, pitchL;
    int nG, nL, cb, p2G, p2L, shared_grid;
};

template <typename T> struct FwdP {
    DimsP d;
    const T *U, *V, *Tm, *SH, *QL, *QI, *Pf, *Ph, *Zgfull, *Zghalf, *zf, *zh;
    const T *u_d, *v_d, *thl_d, *qt_d, *ql_d, *ps_d, *rain, *rain_last;
    T factor, dt;
    T *f_u, *f_v, *f_thl, *f_qt, *f_ql, *ql_ref, *f_ps, *u, *v, *thl, *qt, *ps, *Zf, *Zh, *rainrate;
    int32_t *idx;
    const T *Z0M, *Z0H, *QLflux, *QIflux, *SHflux, *TSflux;
    T *z0m, *z0h, *wthl, *wqt;
};

template <typename T> struct BwdP {
    DimsP d;
    const T *Tm, *SH, *QL, *QI, *U, *V, *A, *Zf, *Zgfull, *Zghalf, *zf;
    const T *t_d, *qt_d, *ql_d, *ql_ice_d, *u_d, *v_d, *A_prof;
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

// =================================================================================================
// K1 forward: splib/spcpl.py:171-246 (convert_profiles) + 299-385 (set_les_forcings) for CB columns
// per workgroup; optional fused K2 (spcpl.py:764) and surface fluxes (spcpl.py:136-167).
// LDS per column: xp=Zf reversed | thl_ | qt_ | QL | U | V, each [nG] in ascending-height order;
// then (idx only) zh: [nL] when the LES grid is shared, else [CB x nL].
// =================================================================================================
template <typename T> __global__ __launch_bounds__(BLOCK) void k_forward(const FwdP<T> p)
{
    const DimsP &d = p.d;
    const int nG = d.nG, nL = d.nL, cb = d.cb;
    const int tid = threadIdx.x;
    const int64_t col0 = (int64_t)blockIdx.x * cb;
    const int ncol = (int)((d.n_cols - col0) < cb ? (d.n_cols - col0) : cb);
    T *const lds = reinterpret_cast<T *>(spc_smem);
    T *const lzh = lds + (size_t)cb * 6 * nG;

    // ---- phase 1: load GCM levels (flat over the [ncol x nG] slab), convert, stage reversed ----
    for (int e = tid; e < ncol * nG; e += BLOCK) {
        const int c = e / nG, k = e - c * nG;
        const int64_t col = col0 + c, g = col * d.pitchG + k;
        const T zsurf = p.Zghalf[col * d.pitchGh + nG];
        const T tt = p.Tm[g], sh = p.SH[g], ql = p.QL[g], qi = p.QI[g];
        const T zf_k = (p.Zgfull[g] - zsurf) / K<T>::grav;                            // spcpl.py:198
        const T iex = spc_pow(p.Pf[g] / K<T>::pref0, (-K<T>::rd) / K<T>::cp);         // sputils.py:34
        const T thl_ = (tt - (K<T>::rlv * (ql + qi)) / K<T>::cp) * iex;               // spcpl.py:214
        const T qt_ = sh + ql + qi;                                                   // spcpl.py:215
        T *const s = lds + (size_t)c * 6 * nG + (nG - 1 - k);                         // [::-1], spcpl.py:224
        s[0] = zf_k;
        s[nG] = thl_;
        s[2 * nG] = qt_;
        s[3 * nG] = ql;
        s[4 * nG] = p.U[g];
        s[5 * nG] = p.V[g];
        if (p.Zf) p.Zf[g] = zf_k;                                                     // spcpl.py:200
    }
    if (p.idx) {  // stage the LES half levels for the fused index map
        const int nz = d.shared_grid ? nL : ncol * nL;
        for (int e = tid; e < nz; e += BLOCK) {
            const int c = e / nL, l = e - c * nL;
            lzh[e] = d.shared_grid ? p.zh[e] : p.zh[(col0 + c) * d.pitchL + l];
        }
    }
    __syncthreads();

    // ---- phase 2: every LES level of the slab: interpolate 5 fields, form the forcings ----------
    for (int e = tid; e < ncol * nL; e += BLOCK) {
        const int c = e / nL, l = e - c * nL;
        const int64_t col = col0 + c, o = col * d.pitchL + l;
        const T *const s = lds + (size_t)c * 6 * nG;
        const T h = d.shared_grid ? p.zf[l] : p.zf[o];                                // spcpl.py:222
        const T ud = p.u_d[o], vd = p.v_d[o], thld = p.thl_d[o], qtd = p.qt_d[o], qld = p.ql_d[o];
        const Bracket<T> b = bracket(s, nG, d.p2G, h);
        const T thl = interp_at(b, s + nG);                                           // spcpl.py:224
        const T qt = interp_at(b, s + 2 * nG);                                        // spcpl.py:225
        const T ql = interp_at(b, s + 3 * nG);                                        // spcpl.py:226
        const T u = interp_at(b, s + 4 * nG);                                         // spcpl.py:227
        const T v = interp_at(b, s + 5 * nG);                                         // spcpl.py:228
        p.f_u[o] = p.factor * (u - ud) / p.dt;                                        // spcpl.py:328
        p.f_v[o] = p.factor * (v - vd) / p.dt;                                        // spcpl.py:329
        p.f_thl[o] = p.factor * (thl - thld) / p.dt;                                  // spcpl.py:330
        p.f_qt[o] = p.factor * (qt - qtd) / p.dt;                                     // spcpl.py:331
        p.f_ql[o] = p.factor * (ql - qld) / p.dt;                                     // spcpl.py:333
        p.ql_ref[o] = ql;                                                             // spcpl.py:347-348
        if (p.u) p.u[o] = u;
        if (p.v) p.v[o] = v;
        if (p.thl) p.thl[o] = thl;
        if (p.qt) p.qt[o] = qt;
    }

    // ---- per-column scalars ------------------------------------------------------------------
    if (tid < ncol) {
        const int64_t col = col0 + tid;
        const T ps = p.Ph[col * d.pitchGh + nG];                                      // spcpl.py:246
        p.f_ps[col] = p.factor * (ps - p.ps_d[col]) / p.dt;                           // spcpl.py:332
        if (p.ps) p.ps[col] = ps;
        if (p.rainrate) p.rainrate[col] = (p.rain[col] - p.rain_last[col]) / p.dt;    // spcpl.py:325
        if (p.wthl) {                                                                 // spcpl.py:136-167
            const T rho = ps / (K<T>::rd * p.Tm[col * d.pitchG + nG - 1]);            // spcpl.py:153
            p.wqt[col] = -(p.QLflux[col] + p.QIflux[col] + p.SHflux[col]) / rho;      // spcpl.py:159
            p.wthl[col] = -p.TSflux[col] * spc_pow(ps / K<T>::pref0, (-K<T>::rd) / K<T>::cp)
                          / (K<T>::cp * rho);                                         // spcpl.py:161
            if (p.z0m) p.z0m[col] = p.Z0M[col];
            if (p.z0h) p.z0h[col] = p.Z0H[col];
        }
    }

    // ---- half-level heights and fused index map (K2): spcpl.py:197, 764 -------------------------
    if (p.Zh) {
        for (int e = tid; e < ncol * (nG + 1); e += BLOCK) {
            const int c = e / (nG + 1), k = e - c * (nG + 1);
            const int64_t gh = (col0 + c) * d.pitchGh;
            p.Zh[gh + k] = (p.Zghalf[gh + k] - p.Zghalf[gh + nG]) / K<T>::grav;
        }
    }
    if (p.idx) {
        for (int e = tid; e < ncol * nG; e += BLOCK) {
            const int c = e / nG, m = e - c * nG;
            const int64_t col = col0 + c, gh = col * d.pitchGh;
            const T Zh_k = (p.Zghalf[gh + (nG - 1 - m)] - p.Zghalf[gh + nG]) / K<T>::grav;
            const T *const zh = d.shared_grid ? lzh : lzh + (size_t)c * nL;
            p.idx[col * d.pitchG + m] = ss_right(zh, nL, Zh_k);
        }
    }
}

// =================================================================================================
// K2 standalone: splib/spcpl.py:26 / 764
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_cloud_idx(const DimsP d, const T *zh_, const T *Zh_, int32_t *idx)
{
    const int nG = d.nG, nL = d.nL, cb = d.cb, tid = threadIdx.x;
    const int64_t col0 = (int64_t)blockIdx.x * cb;
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
template <typename T> __global__ __launch_bounds__(BLOCK) void k_backward(const BwdP<T> p)
{
    const DimsP &d = p.d;
    const int nG = d.nG, nL = d.nL, cb = d.cb, tid = threadIdx.x;
    const int64_t col0 = (int64_t)blockIdx.x * cb;
    const int ncol = (int)((d.n_cols - col0) < cb ? (d.n_cols - col0) : cb);
    const size_t per_col = (size_t)6 * nL + nG;
    T *const lds = reinterpret_cast<T *>(spc_smem);
    T *const lh = lds + (size_t)cb * per_col;

    for (int e = tid; e < ncol * nL; e += BLOCK) {
        const int c = e / nL, l = e - c * nL;
        const int64_t o = (col0 + c) * d.pitchL + l;
        T *const s = lds + (size_t)c * per_col + l;
        s[0] = p.t_d[o];
        s[nL] = p.qt_d[o];
        s[2 * nL] = p.ql_d[o];
        s[3 * nL] = p.ql_ice_d[o];
        s[4 * nL] = p.u_d[o];
        s[5 * nL] = p.v_d[o];
        if (!d.shared_grid) lh[e] = p.zf[o];
    }
    if (d.shared_grid)
        for (int e = tid; e < nL; e += BLOCK) lh[e] = p.zf[e];
    for (int e = tid; e < ncol * nG; e += BLOCK) {
        const int c = e / nG, k = e - c * nG;
        const int64_t col = col0 + c, g = col * d.pitchG + k;
        const T zf_k = p.Zf ? p.Zf[g]
                            : (p.Zgfull[g] - p.Zghalf[col * d.pitchGh + nG]) / K<T>::grav;  // spcpl.py:198
        lds[(size_t)c * per_col + 6 * nL + k] = zf_k;
    }
    __syncthreads();

    for (int e = tid; e < ncol * nG; e += BLOCK) {
        const int c = e / nG, k = e - c * nG;
        const int64_t col = col0 + c, g = col * d.pitchG + k;
        const T *const s = lds + (size_t)c * per_col;
        const T *const h = d.shared_grid ? lh : lh + (size_t)c * nL;
        const T *const Zf = s + 6 * nL;
        const T tt = p.Tm[g], sh = p.SH[g], qlg = p.QL[g], qig = p.QI[g], ug = p.U[g], vg = p.V[g], ag = p.A[g];
        const T a_d = p.A_prof[col * d.pitchG + (nG - 1 - k)];                         // spcpl.py:404
        const T x = Zf[k];
        const int start_index = ss_left_neg(Zf, nG, h[nL - 1]);                        // spcpl.py:498
        const Bracket<T> b = bracket(h, nL, d.p2L, x);
        T t_i, qt_i, ql_i, qlw_i, qli_i, u_i, v_i;
        if (b.mode == 0) {
            const int j = b.j;
            const T ql0 = s[2 * nL + j], ql1 = s[2 * nL + j + 1], qi0 = s[3 * nL + j], qi1 = s[3 * nL + j + 1];
            t_i = lerp_np(x, b.x0, b.x1, s[j], s[j + 1]);                              // spcpl.py:471
            qt_i = lerp_np(x, b.x0, b.x1, s[nL + j], s[nL + j + 1]);                   // spcpl.py:472
            ql_i = lerp_np(x, b.x0, b.x1, ql0, ql1);                                   // spcpl.py:473
            qlw_i = lerp_np(x, b.x0, b.x1, ql0 - qi0, ql1 - qi1);                      // spcpl.py:402,474
            qli_i = lerp_np(x, b.x0, b.x1, qi0, qi1);                                  // spcpl.py:475
            u_i = lerp_np(x, b.x0, b.x1, s[4 * nL + j], s[4 * nL + j + 1]);            // spcpl.py:476
            v_i = lerp_np(x, b.x0, b.x1, s[5 * nL + j], s[5 * nL + j + 1]);            // spcpl.py:477
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
        T f_T = p.factor * (t_i - tt) / p.dt;                                          // spcpl.py:518
        T f_SH = p.factor * ((qt_i - ql_i) - sh) / p.dt;                               // spcpl.py:519
        T f_QL = p.factor * (qlw_i - qlg) / p.dt;                                      // spcpl.py:520
        T f_QI = p.factor * (qli_i - qig) / p.dt;                                      // spcpl.py:521
        T f_U = p.factor * (u_i - ug) / p.dt;                                          // spcpl.py:524
        T f_V = p.factor * (v_i - vg) / p.dt;                                          // spcpl.py:525
        T f_A = p.factor * (a_d - ag) / p.dt;                                          // spcpl.py:526
        if (k < start_index) {  // `f[0:start_index] *= 0` (spcpl.py:527-533): -x -> -0, NaN stays NaN
            const T zero = T(0);
            f_T *= zero; f_SH *= zero; f_QL *= zero; f_QI *= zero; f_U *= zero; f_V *= zero; f_A *= zero;
        }
        p.f_T[g] = f_T;
        p.f_SH[g] = f_SH;
        p.f_QL[g] = f_QL;
        p.f_QI[g] = f_QI;
        p.f_U[g] = f_U;
        p.f_V[g] = f_V;
        p.f_A[g] = f_A;
        if (p.start_index && k == 0) p.start_index[col] = start_index;
    }
}

// =================================================================================================
// K5 diagnostics: splib/spcpl.py:176, 197-198, 214-215 (GCM levels); 402, 408-409 (LES levels)
// LDS per column (only when pf/t requested): Zf reversed | Pf reversed, each [nG].
// =================================================================================================
template <typename T> __global__ __launch_bounds__(BLOCK) void k_diag(const DiagP<T> p)
{
    const DimsP &d = p.d;
    const int nG = d.nG, nL = d.nL, cb = d.cb, tid = threadIdx.x;
    const int64_t col0 = (int64_t)blockIdx.x * cb;
    const int ncol = (int)((d.n_cols - col0) < cb ? (d.n_cols - col0) : cb);
    T *const lds = reinterpret_cast<T *>(spc_smem);
    const T cc = K<T>::rv / K<T>::rd - T(1);                                           // spcpl.py:175
    for (int e = tid; e < ncol * nG; e += BLOCK) {
        const int c = e / nG, k = e - c * nG;
        const int64_t col = col0 + c, g = col * d.pitchG + k;
        const T tt = p.Tm[g], sh = p.SH[g], ql = p.QL[g], qi = p.QI[g], pf = p.Pf[g];
        const T zf_k = (p.Zgfull[g] - p.Zghalf[col * d.pitchGh + nG]) / K<T>::grav;
        if (p.Tv) p.Tv[g] = tt * (T(1) + cc * sh - (ql + qi));                          // spcpl.py:176
        if (p.THL) p.THL[g] = (tt - (K<T>::rlv * (ql + qi)) / K<T>::cp) * spc_pow(pf / K<T>::pref0, (-K<T>::rd) / K<T>::cp);
        if (p.QT) p.QT[g] = sh + ql + qi;
        if (p.Zf) p.Zf[g] = zf_k;
        T *const s = lds + (size_t)c * 2 * nG + (nG - 1 - k);
        s[0] = zf_k;
        s[nG] = pf;
    }
    if (p.Zh) {
        for (int e = tid; e < ncol * (nG + 1); e += BLOCK) {
            const int c = e / (nG + 1), k = e - c * (nG + 1);
            const int64_t gh = (col0 + c) * d.pitchGh;
            p.Zh[gh + k] = (p.Zghalf[gh + k] - p.Zghalf[gh + nG]) / K<T>::grav;         // spcpl.py:197
        }
    }
    __syncthreads();
    if (p.zf && (p.pf || p.t || p.ql_water)) {
        for (int e = tid; e < ncol * nL; e += BLOCK) {
            const int c = e / nL, l = e - c * nL;
            const int64_t o = (col0 + c) * d.pitchL + l;
            const T *const s = lds + (size_t)c * 2 * nG;
            const T h = d.shared_grid ? p.zf[l] : p.zf[o];
            const Bracket<T> b = bracket(s, nG, d.p2G, h);
            const T pf = interp_at(b, s + nG);                                         // spcpl.py:408
            if (p.pf) p.pf[o] = pf;
            if (p.t)                                                                   // spcpl.py:409
                p.t[o] = p.thl_d[o] * spc_pow(pf / K<T>::pref0, K<T>::rd / K<T>::cp) + K<T>::rlv * p.ql_d[o] / K<T>::cp;
            if (p.ql_water) p.ql_water[o] = p.ql_d[o] - p.ql_ice_d[o];                  // spcpl.py:402
        }
    }
}

// 16 B/lane streaming copy: the measured-bandwidth yardstick reported beside the roofline.
__global__ __launch_bounds__(BLOCK) void k_copy16(uint4 *dst, const uint4 *src, int64_t n16)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += (int64_t)gridDim.x * BLOCK)
        dst[i] = src[i];
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

// LDS elements per column / per block for each pass (pass 0 fwd, 1 bwd, 2 idx, 3 diag)
void lds_elems(const spc_dims *d, int pass, bool with_idx, size_t *per_col, size_t *fixed)
{
    const size_t nG = d->nG, nL = d->nL;
    const bool sh = d->les_grid_shared != 0;
    switch (pass) {
    case 0: *per_col = 6 * nG + ((with_idx && !sh) ? nL : 0); *fixed = (with_idx && sh) ? nL : 0; break;
    case 1: *per_col = 6 * nL + nG + (sh ? 0 : nL); *fixed = sh ? nL : 0; break;
    case 2: *per_col = sh ? 0 : nL; *fixed = sh ? nL : 0; break;
    default: *per_col = 2 * nG; *fixed = 0; break;
    }
}

// Columns per workgroup: as many as keeps >= 8 workgroups per CU in flight (256 CUs), within LDS.
int pick_cb(const spc_dims *d, int pass, bool with_idx, size_t esize)
{
    size_t per_col, fixed;
    lds_elems(d, pass, with_idx, &per_col, &fixed);
    int cb = d->cols_per_block;
    if (cb <= 0) {
        cb = 8;
        while (cb > 1 && d->n_cols / cb < 2048) cb >>= 1;
    }
    while (cb > 1 && (per_col * cb + fixed) * esize > (size_t)MAX_LDS_BYTES) --cb;
    return cb;
}

DimsP make_dims(const spc_dims *d, int cb)
{
    DimsP p;
    p.n_cols = d->n_cols; p.pitchG = d->pitchG; p.pitchGh = d->pitchGh; p.pitchL = d->pitchL;
    p.nG = d->nG; p.nL = d->nL; p.cb = cb; p.p2G = floor_pow2(d->nG); p.p2L = floor_pow2(d->nL);
    p.shared_grid = d->les_grid_shared != 0;
    return p;
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
    const int cb = pick_cb(d, 0, with_idx, sizeof(T));
    size_t per_col, fixed;
    lds_elems(d, 0, with_idx, &per_col, &fixed);
    const size_t smem = (per_col * cb + fixed) * sizeof(T);
    if (smem > (size_t)MAX_LDS_BYTES)
        return fail(SPC_ERR_UNSUPPORTED, "%sforward needs %lld B of LDS per workgroup (max %lld)", "", (long long)smem, MAX_LDS_BYTES);
    FwdP<T> p;
    p.d = make_dims(d, cb);
#define CP(f) p.f = (const T *)a->f
#define OP(f) p.f = (T *)a->f
    CP(U); CP(V); p.Tm = (const T *)a->T; CP(SH); CP(QL); CP(QI); CP(Pf); CP(Ph); CP(Zgfull); CP(Zghalf); CP(zf); CP(zh);
    CP(u_d); CP(v_d); CP(thl_d); CP(qt_d); CP(ql_d); CP(ps_d); CP(rain); CP(rain_last);
    p.factor = (T)a->factor; p.dt = (T)a->dt;
    OP(f_u); OP(f_v); OP(f_thl); OP(f_qt); OP(f_ql); OP(ql_ref); OP(f_ps); OP(u); OP(v); OP(thl); OP(qt); OP(ps);
    OP(Zf); OP(Zh); OP(rainrate); p.idx = a->idx;
    CP(Z0M); CP(Z0H); CP(QLflux); CP(QIflux); CP(SHflux); CP(TSflux); OP(z0m); OP(z0h); OP(wthl); OP(wqt);
    const unsigned grid = (unsigned)((d->n_cols + cb - 1) / cb);
    hipLaunchKernelGGL(k_forward<T>, dim3(grid), dim3(BLOCK), smem, (hipStream_t)stream, p);
    return launch_status("k_forward");
}

template <typename T>
int cloud_idx_impl(const spc_dims *d, const void *zh, const void *Zh, int32_t *idx, void *stream)
{
    int rc = validate(d);
    if (rc) return rc;
    if (d->n_cols == 0) return SPC_OK;
    REQUIRE(zh, "zh"); REQUIRE(Zh, "Zh"); REQUIRE(idx, "idx");
    const int cb = pick_cb(d, 2, true, sizeof(T));
    size_t per_col, fixed;
    lds_elems(d, 2, true, &per_col, &fixed);
    const size_t smem = (per_col * cb + fixed) * sizeof(T);
    if (smem > (size_t)MAX_LDS_BYTES)
        return fail(SPC_ERR_UNSUPPORTED, "%scloud_indices needs %lld B of LDS (max %lld)", "", (long long)smem, MAX_LDS_BYTES);
    const unsigned grid = (unsigned)((d->n_cols + cb - 1) / cb);
    hipLaunchKernelGGL(k_cloud_idx<T>, dim3(grid), dim3(BLOCK), smem, (hipStream_t)stream, make_dims(d, cb),
                       (const T *)zh, (const T *)Zh, idx);
    return launch_status("k_cloud_idx");
}

template <typename T> int backward_impl(const spc_dims *d, const spc_backward_args *a, void *stream)
{
    int rc = validate(d);
    if (rc) return rc;
    if (!a) return fail(SPC_ERR_INVALID_ARGUMENT, "%sargs is NULL");
    if (a->conservative)
        return fail(SPC_ERR_UNSUPPORTED, "%sconservative coarsening (sputils.interp_c) is not built yet");
    if (d->n_cols == 0) return SPC_OK;
    REQUIRE(a->T, "T"); REQUIRE(a->SH, "SH"); REQUIRE(a->QL, "QL"); REQUIRE(a->QI, "QI"); REQUIRE(a->U, "U");
    REQUIRE(a->V, "V"); REQUIRE(a->A, "A"); REQUIRE(a->zf, "zf"); REQUIRE(a->t_d, "t_d"); REQUIRE(a->qt_d, "qt_d");
    REQUIRE(a->ql_d, "ql_d"); REQUIRE(a->ql_ice_d, "ql_ice_d"); REQUIRE(a->u_d, "u_d"); REQUIRE(a->v_d, "v_d");
    REQUIRE(a->A_prof, "A_prof"); REQUIRE(a->f_T, "f_T"); REQUIRE(a->f_SH, "f_SH"); REQUIRE(a->f_QL, "f_QL");
    REQUIRE(a->f_QI, "f_QI"); REQUIRE(a->f_U, "f_U"); REQUIRE(a->f_V, "f_V"); REQUIRE(a->f_A, "f_A");
    if (!a->Zf && (!a->Zgfull || !a->Zghalf))
        return fail(SPC_ERR_INVALID_ARGUMENT, "%sneither Zf nor (Zgfull, Zghalf) given");
    const int cb = pick_cb(d, 1, false, sizeof(T));
    size_t per_col, fixed;
    lds_elems(d, 1, false, &per_col, &fixed);
    const size_t smem = (per_col * cb + fixed) * sizeof(T);
    if (smem > (size_t)MAX_LDS_BYTES)
        return fail(SPC_ERR_UNSUPPORTED, "%sbackward needs %lld B of LDS per workgroup (max %lld)", "", (long long)smem, MAX_LDS_BYTES);
    BwdP<T> p;
    p.d = make_dims(d, cb);
    p.Tm = (const T *)a->T; CP(SH); CP(QL); CP(QI); CP(U); CP(V); CP(A); CP(Zf); CP(Zgfull); CP(Zghalf); CP(zf);
    CP(t_d); CP(qt_d); CP(ql_d); CP(ql_ice_d); CP(u_d); CP(v_d); CP(A_prof);
    p.factor = (T)a->factor; p.dt = (T)a->dt;
    OP(f_T); OP(f_SH); OP(f_QL); OP(f_QI); OP(f_U); OP(f_V); OP(f_A); p.start_index = a->start_index;
    const unsigned grid = (unsigned)((d->n_cols + cb - 1) / cb);
    hipLaunchKernelGGL(k_backward<T>, dim3(grid), dim3(BLOCK), smem, (hipStream_t)stream, p);
    return launch_status("k_backward");
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
    const int cb = pick_cb(d, 3, false, sizeof(T));
    size_t per_col, fixed;
    lds_elems(d, 3, false, &per_col, &fixed);
    const size_t smem = (per_col * cb + fixed) * sizeof(T);
    if (smem > (size_t)MAX_LDS_BYTES)
        return fail(SPC_ERR_UNSUPPORTED, "%sdiagnostics needs %lld B of LDS (max %lld)", "", (long long)smem, MAX_LDS_BYTES);
    DiagP<T> p;
    p.d = make_dims(d, cb);
    p.Tm = (const T *)a->T; CP(SH); CP(QL); CP(QI); CP(Pf); CP(Zgfull); CP(Zghalf); CP(zf); CP(thl_d); CP(ql_d); CP(ql_ice_d);
    OP(Tv); OP(THL); OP(QT); OP(Zf); OP(Zh); OP(pf); OP(t); OP(ql_water);
    const unsigned grid = (unsigned)((d->n_cols + cb - 1) / cb);
    hipLaunchKernelGGL(k_diag<T>, dim3(grid), dim3(BLOCK), smem, (hipStream_t)stream, p);
    return launch_status("k_diag");
}
#undef CP
#undef OP

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

int spc_pick_cols_per_block(const spc_dims *d, int pass)
{
    int rc = validate(d);
    if (rc) return rc;
    if (pass < 0 || pass > 3) return fail(SPC_ERR_INVALID_ARGUMENT, "%spass must be 0..3");
    return pick_cb(d, pass, pass == 0 || pass == 2, sizeof(double));
}

int spc_stream_copy(void *dst, const void *src, int64_t bytes, void *stream)
{
    if (bytes < 0 || (bytes & 15) || !dst || !src) return fail(SPC_ERR_INVALID_ARGUMENT, "%sstream_copy: bytes must be a multiple of 16, pointers non-NULL");
    if (bytes == 0) return SPC_OK;
    hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(BLOCK), 0, (hipStream_t)stream, (uint4 *)dst, (const uint4 *)src, bytes / 16);
    return launch_status("k_copy16");
}

}  // extern "C"
