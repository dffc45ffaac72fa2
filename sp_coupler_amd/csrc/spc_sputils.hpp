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

// integral() of splib/sputils.py:94-161 over [a, b] of the piecewise-constant q on the cells of z (n points), optional
// weights w.  *none: an end point lies outside z (the reference prints a message and returns None).
template <typename T> __device__ T su_integral(T a, T b, const T *z, int n, const T *qv, const T *w, bool *none)
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
    if (!w) {
        auto term = [&](int i) { return qv[ia + i] * (z[ia + i + 1] - z[ia + i]); };               // sputils.py:146
        const T S = cnt <= 128 ? T(0) + vn_leaf(term, 0, cnt) : vn_npsum(term, cnt);
        return ((S - qv[ia] * da) - qv[ib] * db) * sign;                                           // sputils.py:149-152
    }
    auto term = [&](int i) { return (w[ia + i] * qv[ia + i]) * (z[ia + i + 1] - z[ia + i]); };      // sputils.py:154
    auto termw = [&](int i) { return w[ia + i] * (z[ia + i + 1] - z[ia + i]); };                   // sputils.py:159
    const T S = cnt <= 128 ? T(0) + vn_leaf(term, 0, cnt) : vn_npsum(term, cnt);
    const T Sw = cnt <= 128 ? T(0) + vn_leaf(termw, 0, cnt) : vn_npsum(termw, cnt);
    const T num = (S - (w[ia] * qv[ia]) * da) - (w[ib] * qv[ib]) * db;                             // sputils.py:156-157
    const T den = (Sw - w[ia] * da) - w[ib] * db;                                                  // sputils.py:161-162
    return num / den * sign;
}

enum { SU_INTERP_C = 0, SU_INTERP_RHO = 1, SU_INTEGRAL = 2 };

struct SuCoarseP {
    int64_t n_rows, pitch_Zh, pitch_zh, pitch_q, pitch_out;
    int nG, nL, mode, stage;
    const void *Zh, *zh, *q, *rho;
    void *out;
};

// one workgroup per row; thread k: the layer [Zh[k+1], Zh[k]]
template <typename T> __global__ __launch_bounds__(SU_THREADS) void k_interp_c(const SuCoarseP p)
{
    T *const lds = reinterpret_cast<T *>(spc_smem);
    const int64_t row = blockIdx.x;
    const int nL = p.nL, tid = threadIdx.x;
    const T *z = (const T *)p.zh + row * p.pitch_zh, *qv = (const T *)p.q + row * p.pitch_q;
    const T *w = p.rho ? (const T *)p.rho + row * p.pitch_q : nullptr;
    const T *const Zh = (const T *)p.Zh + row * p.pitch_Zh;
    if (p.stage) {                                  // z[nL] | q[nL] | rho[nL]
        for (int e = tid; e < nL; e += SU_THREADS) {
            lds[e] = z[e];
            lds[nL + e] = e < nL - 1 ? qv[e] : T(0);          // nL points bound nL - 1 cells: the last element is never used
            if (w) lds[2 * nL + e] = e < nL - 1 ? w[e] : T(0);
        }
        __syncthreads();
        z = lds; qv = lds + nL; if (w) w = lds + 2 * nL;
    }
    for (int k = tid; k < p.nG; k += SU_THREADS) {
        const T top = Zh[k], bot = Zh[k + 1];
        T r = T(0);                                                                    // Q = zeros / RHO = zeros
        if (p.mode == SU_INTEGRAL || top < z[nL - 1]) {                                // sputils.py:187 / 195
            bool none;
            r = su_integral(bot, top, z, nL, qv, p.mode == SU_INTERP_RHO ? (const T *)nullptr : w, &none);
            if (none) r = T(0) / T(0);                                                 // Q[i] = None stores NaN (numpy 2.x)
            else if (p.mode == SU_INTERP_RHO) r = r / (top - bot);                      // sputils.py:196
        }
        ((T *)p.out)[row * p.pitch_out + k] = r;
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

