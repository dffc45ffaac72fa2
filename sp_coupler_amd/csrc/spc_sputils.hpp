// spc_sputils.hpp -- K7: the helpers of splib/sputils.py as standalone batched operators (included by spc_hip.hip).
//   exner / iexner      splib/sputils.py:28-34
//   interp              splib/sputils.py:82-86   (numpy.interp per row)
//   searchsorted        splib/sputils.py:88-91   (numpy.searchsorted per row)
//   integral / interp_c / interp_rho   splib/sputils.py:94-161, 173-189, 191-197
//   rms                 splib/sputils.py:23-24
// The fused kernels K1 / K3 / K4 contain the same arithmetic (and share the device functions: spc_pow, bracket /
// interp_at, ss_right, scan_cell, vn_npsum); these entry points serve callers that use a helper on its own -- the
// commented-out alternatives of spcpl.py:435-466, diagnostics, tests written against sputils -- for ALL rows (columns) at
// once.  A "row" is one independent 1-D problem; arrays are [n_rows x n] with an element pitch between rows, pitch 0 =
// one row shared by all.  One workgroup serves RB rows: their sample arrays are staged in LDS when they fit (64 KiB),
// otherwise read from global memory through the same (flat) pointers.
#pragma once

constexpr int SU_THREADS = 256;
constexpr size_t SU_MAX_LDS = 64 * 1024;

template <typename T> __global__ __launch_bounds__(SU_THREADS) void k_exner(int64_t n, const T *p, T *out, int inverse)
{
    const T y = inverse ? (-K<T>::rd) / K<T>::cp : K<T>::rd / K<T>::cp;                  // sputils.py:34 / 29
    for (int64_t i = (int64_t)blockIdx.x * SU_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SU_THREADS)
        out[i] = spc_pow(div_pref0(p[i]), y);
}

struct SuInterpP {
    int64_t n_rows, pitch_x, pitch_xp, pitch_fp, pitch_out;
    int n_x, n_xp, p2, rb, stage;      // rb rows per workgroup; stage: sample arrays go through LDS
    const void *x, *xp, *fp;
    void *out;
};

template <typename T> __global__ __launch_bounds__(SU_THREADS) void k_interp(const SuInterpP q)
{
    T *const lds = reinterpret_cast<T *>(spc_smem);
    const int64_t row0 = (int64_t)blockIdx.x * q.rb;
    const int nrow = (int)((q.n_rows - row0) < q.rb ? (q.n_rows - row0) : q.rb);
    const T *const x = (const T *)q.x, *const xp = (const T *)q.xp, *const fp = (const T *)q.fp;
    T *const out = (T *)q.out;
    const int n_xp = q.n_xp, tid = threadIdx.x;
    // LDS: fp[rb][n_xp] | xp[rb][n_xp] (or xp[n_xp] when shared)
    T *const lfp = lds, *const lxp = lds + (size_t)q.rb * n_xp;
    if (q.stage) {
        for (int e = tid; e < nrow * n_xp; e += SU_THREADS) {
            const int r = e / n_xp, j = e - r * n_xp;
            lfp[e] = fp[(row0 + r) * q.pitch_fp + j];
            if (q.pitch_xp) lxp[e] = xp[(row0 + r) * q.pitch_xp + j];
        }
        if (!q.pitch_xp)
            for (int e = tid; e < n_xp; e += SU_THREADS) lxp[e] = xp[e];
        __syncthreads();
    }
    for (int e = tid; e < nrow * q.n_x; e += SU_THREADS) {
        const int r = e / q.n_x, i = e - r * q.n_x;
        const int64_t row = row0 + r;
        const T *const xpr = q.stage ? (q.pitch_xp ? lxp + (size_t)r * n_xp : lxp) : xp + row * q.pitch_xp;
        const T *const fpr = q.stage ? lfp + (size_t)r * n_xp : fp + row * q.pitch_fp;
        const Bracket<T> b = bracket(xpr, n_xp, q.p2, x[row * q.pitch_x + i]);
        out[row * q.pitch_out + i] = interp_at(b, fpr);
    }
}

// numpy.searchsorted(a, v, side='left'): first i with !(a[i] < v)   (splib/sputils.py:88-91)
template <typename T> __device__ __forceinline__ int ss_left(const T *a, int n, T key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (np_lt(a[mid], key)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct SuSearchP {
    int64_t n_rows, pitch_a, pitch_v, pitch_out;
    int n_a, n_v, rb, stage, right;
    const void *a, *v;
    int64_t *out;
};

template <typename T> __global__ __launch_bounds__(SU_THREADS) void k_searchsorted(const SuSearchP q)
{
    T *const lds = reinterpret_cast<T *>(spc_smem);
    const int64_t row0 = (int64_t)blockIdx.x * q.rb;
    const int nrow = (int)((q.n_rows - row0) < q.rb ? (q.n_rows - row0) : q.rb);
    const T *const a = (const T *)q.a, *const v = (const T *)q.v;
    const int n_a = q.n_a, tid = threadIdx.x;
    if (q.stage) {
        if (q.pitch_a) {
            for (int e = tid; e < nrow * n_a; e += SU_THREADS) {
                const int r = e / n_a, j = e - r * n_a;
                lds[e] = a[(row0 + r) * q.pitch_a + j];
            }
        } else {
            for (int e = tid; e < n_a; e += SU_THREADS) lds[e] = a[e];
        }
        __syncthreads();
    }
    for (int e = tid; e < nrow * q.n_v; e += SU_THREADS) {
        const int r = e / q.n_v, i = e - r * q.n_v;
        const int64_t row = row0 + r;
        const T *const ar = q.stage ? (q.pitch_a ? lds + (size_t)r * n_a : lds) : a + row * q.pitch_a;
        const T key = v[row * q.pitch_v + i];
        q.out[row * q.pitch_out + i] = q.right ? ss_right(ar, n_a, key) : ss_left(ar, n_a, key);
    }
}

enum { SU_INTERP_C = 0, SU_INTERP_RHO = 1, SU_INTEGRAL = 2 };

struct SuCoarseP {
    int64_t n_rows, pitch_Zh, pitch_zh, pitch_q, pitch_out;
    int nG, nL, mode, stage, rb;
    const void *Zh, *zh, *q, *rho;
    void *out;
};

// integral() of splib/sputils.py:94-161 over [a, b] of the piecewise-constant q on the cells of z (n points), optional
// weights w.  *none: an end point lies outside z (the reference prints a message and returns None).  The (up to two)
// sums of a call -- sum w q dz and sum w dz -- run through ONE instance of the summation code (`pass`), and numpy's
// pairwise recursion is unrolled to the depth PD the host derived from n (as in K4; PD = -1: explicit stack).
template <typename T, int PD> __device__ __forceinline__ T su_integral(T a, T b, const T *z, int n, const T *qv, const T *w, bool weighted, bool *none)
{
    *none = false;
    if (a < z[0] || a > z[n - 1] || b < z[0] || b > z[n - 1]) { *none = true; return T(0); }     // sputils.py:113-115
    T sign = T(1);
    if (a > b) { sign = T(-1); const T t = a; a = b; b = t; }                                      // sputils.py:117-120
    const int ia = scan_cell(z, n, a);                                                             // sputils.py:122-124
    int ib = scan_cell(z, n, b);                                                                   // sputils.py:125-127
    if (ib < ia) ib = ia;
    const int cnt = ib - ia + 1;
    const T da = a - z[ia], db = z[ib + 1] - b;
    T num = T(0), den = T(1);
    const int npass = weighted ? 2 : 1;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
        // pass 0: q dz (sputils.py:146) or (w q) dz (:154); pass 1: w dz (:159)
        auto val = [&](int i) { return pass ? w[i] : (weighted ? w[i] * qv[i] : qv[i]); };
        auto term = [&](int i) { return val(ia + i) * (z[ia + i + 1] - z[ia + i]); };
        T S;
        if constexpr (PD >= 0) S = T(0) + vn_pw<PD>(term, 0, cnt);
        else S = cnt <= 128 ? T(0) + vn_leaf(term, 0, cnt) : vn_npsum(term, cnt);
        const T v = (S - val(ia) * da) - val(ib) * db;                                             // sputils.py:149-152, 156-162
        if (pass) den = v; else num = v;
    }
    return weighted ? num / den * sign : num * sign;
}

// RB rows per workgroup; thread = (row, layer k): the layer [Zh[k+1], Zh[k]].  STAGE: the rows' zh, q, rho go through LDS
// (a compile-time switch: mixing LDS and global addresses in one pointer makes every access a 64-bit flat one).
template <typename T, int PD, bool STAGE> __global__ __launch_bounds__(SU_THREADS) void k_interp_c(const SuCoarseP p)
{
    T *const lds = reinterpret_cast<T *>(spc_smem);
    const int64_t row0 = (int64_t)blockIdx.x * p.rb;
    const int nrow = (int)((p.n_rows - row0) < p.rb ? (p.n_rows - row0) : p.rb);
    const int nL = p.nL, nG = p.nG, tid = threadIdx.x;
    const T *const zg = (const T *)p.zh, *const qg = (const T *)p.q, *const wg = (const T *)p.rho;
    // LDS: q[rb][nL] | rho[rb][nL] | z[rb][nL] (or z[nL] when shared)
    T *const lq = lds, *const lw = lds + (size_t)p.rb * nL, *const lz = lw + (size_t)p.rb * nL;
    if constexpr (STAGE) {
        for (int e = tid; e < nrow * nL; e += SU_THREADS) {
            const int r = e / nL, l = e - r * nL;
            const int64_t o = (row0 + r) * p.pitch_q + l;
            lq[e] = l < nL - 1 ? qg[o] : T(0);                // nL points bound nL - 1 cells: the last element is never used
            if (wg) lw[e] = l < nL - 1 ? wg[o] : T(0);
            if (p.pitch_zh) lz[e] = zg[(row0 + r) * p.pitch_zh + l];
        }
        if (!p.pitch_zh)
            for (int e = tid; e < nL; e += SU_THREADS) lz[e] = zg[e];
        __syncthreads();
    }
    for (int e = tid; e < nrow * nG; e += SU_THREADS) {
        const int r = e / nG, k = e - r * nG;
        const int64_t row = row0 + r;
        const T *const z = STAGE ? lz + (p.pitch_zh ? (size_t)r * nL : 0) : zg + row * p.pitch_zh;
        const T *const qv = STAGE ? lq + (size_t)r * nL : qg + row * p.pitch_q;
        const T *const wr = STAGE ? lw + (size_t)r * nL : wg + row * p.pitch_q;
        const bool weighted = wg != nullptr;
        const T *const Zh = (const T *)p.Zh + row * p.pitch_Zh;
        const T top = Zh[k], bot = Zh[k + 1];
        T res = T(0);                                                                  // Q = zeros / RHO = zeros
        if (p.mode == SU_INTEGRAL || top < z[nL - 1]) {                                // sputils.py:187 / 195
            bool none;
            res = su_integral<T, PD>(bot, top, z, nL, qv, wr, weighted, &none);
            if (none) res = T(0) / T(0);                                               // Q[i] = None stores NaN (numpy 2.x)
            else if (p.mode == SU_INTERP_RHO) res = res / (top - bot);                  // sputils.py:196
        }
        ((T *)p.out)[row * p.pitch_out + k] = res;
    }
}

// sqrt(mean(a**2)) per row: numpy's mean = add.reduce (pairwise, 8192-element chunks) / n; one thread per row
template <typename T> __global__ __launch_bounds__(64) void k_rms(int64_t n_rows, int n, int64_t pitch, const T *a, T *out)
{
    const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= n_rows) return;
    const T *const ar = a + row * pitch;
    auto term = [&](int i) { return ar[i] * ar[i]; };
    const T S = vn_npsum(term, n);
    out[row] = sqrt(S / (T)n);
}

