// spc_sputils.hpp -- K7: the helpers of splib/sputils.py as standalone batched operators (included by spc_hip.hip).
//   exner / iexner      splib/sputils.py:28-34
//   interp              splib/sputils.py:82-86   (numpy.interp per row)
//   searchsorted        splib/sputils.py:88-91   (numpy.searchsorted per row)
//   integral / interp_c / interp_rho   splib/sputils.py:94-161, 173-189, 191-197
//   rms                 splib/sputils.py:23-24
// The fused kernels K1 / K3 / K4 contain the same arithmetic (and share the device functions: spc_pow, bracket /
// interp_at, ss_right, scan_cell, vn_leaf / vn_pw); these entry points serve callers that use a helper on its own -- the
// commented-out alternatives of spcpl.py:435-466, diagnostics, tests written against sputils -- for ALL rows (columns) at
// once.  A "row" is one independent 1-D problem; arrays are [n_rows x n] with an element pitch between rows, pitch 0 =
// one row shared by all.
//
// Round 4 (second generation; counters of the first in profiles/r04_k7_baseline.log): the first generation moved exactly its
// algorithmic bytes (PMC traffic 1.00-1.06 x) but slowly -- its LDS staging went through ONE pointer that could be LDS or
// global, so every access was a FLAT one (3 LDS instructions per wave, 30 flat loads: interp); interp_c summed w q dz with
// 3-4 LDS reads and 2 multiplications per term in two passes (1 279 VALU instructions per wave); rms gave a whole row to
// one thread (559 waves on 1 024 SIMDs); exner looped 3 x over a capped grid with one pow in flight per thread.  Now:
//   * staging is a compile-time switch (STAGE) everywhere: LDS pointers are LDS pointers; rows that do not fit read global;
//   * a workgroup owns a SLAB of rows (su_rows: ~1 000-1 300 outputs, as K1's 8-column slabs) and loads it with flat,
//     coalesced accesses, all of a thread's loads issued before the first LDS store;
//   * launches that write <= 32 MiB store write-through (WT, as K1 / K3: nothing left dirty in L2 at the end);
//   * interp_c forms the per-CELL terms (w q) dz and w dz once while staging; a layer's two sums are then pure additions
//     of LDS values, run side by side in ONE numpy-ordered pass (Pair2);
//   * rms: 8 lanes per row = the 8 accumulators of numpy's leaf, combined by shuffles in numpy's order (64-B segments,
//     no LDS, 8 x the parallelism per row);
//   * exner: 4 elements per thread, loads up front, 4 independent pow chains, one pass over an uncapped grid.
#pragma once

constexpr int SU_THREADS = 256;
constexpr size_t SU_MAX_LDS = 64 * 1024;

// ---- staging: cnt elements, element e from at(e), into LDS dst[e]; U loads of a thread in flight before its first store --
template <int U, typename T, typename At> __device__ __forceinline__ void su_stage(T *dst, int cnt, int tid, const At &at)
{
    for (int e0 = tid; e0 < cnt; e0 += SU_THREADS * U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * SU_THREADS;
            v[u] = e < cnt ? at(e) : T(0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * SU_THREADS;
            if (e < cnt) dst[e] = v[u];
        }
    }
}

// element e of the [nrow x n] slab that starts at row0: flat when the rows are contiguous (pitch == n), else by (row, j)
template <typename T> struct SuSlab {
    const T *base;     // row0's first element
    int64_t pitch;
    int n;
    bool flat;
    __device__ __forceinline__ T operator()(int e) const
    {
        if (flat) return ldg(base + e);
        const int r = e / n, j = e - r * n;
        return ldg(base + (int64_t)r * pitch + j);
    }
};
template <typename T> __device__ __forceinline__ SuSlab<T> su_slab(const void *p, int64_t row0, int64_t pitch, int n)
{
    return SuSlab<T>{(const T *)p + row0 * pitch, pitch, n, pitch == (int64_t)n};
}

// ---- exner ---------------------------------------------------------------------------------------------------------
template <typename T, int WT> __global__ __launch_bounds__(SU_THREADS) void k_exner(int64_t n, const T *p, T *out, int inverse)
{
    const T y = inverse ? (-K<T>::rd) / K<T>::cp : K<T>::rd / K<T>::cp;                  // sputils.py:34 / 29
    const int64_t i0 = (int64_t)blockIdx.x * (SU_THREADS * 4) + threadIdx.x;
    T v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = i0 + u * SU_THREADS;
        v[u] = i < n ? ldg(p + i) : T(1);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = spc_pow(div_pref0(v[u]), y);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = i0 + u * SU_THREADS;
        if (i < n) stg<WT>(out + i, v[u]);
    }
}

// ---- interp --------------------------------------------------------------------------------------------------------
struct SuInterpP {
    int64_t n_rows, pitch_x, pitch_xp, pitch_fp, pitch_out;
    int n_x, n_xp, p2, rb;             // rb rows per workgroup
    const void *x, *xp, *fp;
    void *out;
};

// STAGE: the slab's sample arrays go through LDS: fp[rb][n_xp] | xp[rb][n_xp] (or xp[n_xp] when shared)
template <typename T, bool STAGE, int WT> __global__ __launch_bounds__(SU_THREADS) void k_interp(const SuInterpP q)
{
    T *const lds = reinterpret_cast<T *>(spc_smem);
    const int64_t row0 = (int64_t)blockIdx.x * q.rb;
    const int nrow = (int)((q.n_rows - row0) < q.rb ? (q.n_rows - row0) : q.rb);
    const int n_xp = q.n_xp, n_x = q.n_x, tid = threadIdx.x;
    const T *const xg = (const T *)q.x, *const xpg = (const T *)q.xp, *const fpg = (const T *)q.fp;
    T *const out = (T *)q.out;
    T *const lfp = lds, *const lxp = lds + (size_t)q.rb * n_xp;
    if constexpr (STAGE) {
        su_stage<3>(lfp, nrow * n_xp, tid, su_slab<T>(q.fp, row0, q.pitch_fp, n_xp));
        if (q.pitch_xp) su_stage<3>(lxp, nrow * n_xp, tid, su_slab<T>(q.xp, row0, q.pitch_xp, n_xp));
        else su_stage<1>(lxp, n_xp, tid, [&](int e) { return ldg(xpg + e); });
        __syncthreads();
    }
    const bool flat_x = q.pitch_x == (int64_t)n_x, flat_o = q.pitch_out == (int64_t)n_x;
    const int cnt = nrow * n_x;
    int r = tid / n_x, i = tid - r * n_x;                     // (row, point) of this thread's first output; then stepped
    for (int e = tid; e < cnt; e += SU_THREADS) {
        const int64_t row = row0 + r;
        const T xv = q.pitch_x ? (flat_x ? ldg(xg + row0 * n_x + e) : ldg(xg + row * q.pitch_x + i)) : ldg(xg + i);
        T res;
        if constexpr (STAGE) {
            const T *const xpr = q.pitch_xp ? lxp + (size_t)r * n_xp : lxp;
            res = interp_at(bracket(xpr, n_xp, q.p2, xv), lfp + (size_t)r * n_xp);
        } else {
            res = interp_at(bracket(xpg + row * q.pitch_xp, n_xp, q.p2, xv), fpg + row * q.pitch_fp);
        }
        stg<WT>(flat_o ? out + row0 * n_x + e : out + row * q.pitch_out + i, res);
        i += SU_THREADS;
        while (i >= n_x) { i -= n_x; ++r; }
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

// ---- searchsorted --------------------------------------------------------------------------------------------------
struct SuSearchP {
    int64_t n_rows, pitch_a, pitch_v, pitch_out;
    int n_a, n_v, rb, right;
    const void *a, *v;
    int64_t *out;
};

template <typename T, bool STAGE, int WT> __global__ __launch_bounds__(SU_THREADS) void k_searchsorted(const SuSearchP q)
{
    T *const lds = reinterpret_cast<T *>(spc_smem);
    const int64_t row0 = (int64_t)blockIdx.x * q.rb;
    const int nrow = (int)((q.n_rows - row0) < q.rb ? (q.n_rows - row0) : q.rb);
    const T *const ag = (const T *)q.a, *const vg = (const T *)q.v;
    const int n_a = q.n_a, n_v = q.n_v, tid = threadIdx.x;
    if constexpr (STAGE) {
        if (q.pitch_a) su_stage<3>(lds, nrow * n_a, tid, su_slab<T>(q.a, row0, q.pitch_a, n_a));
        else su_stage<1>(lds, n_a, tid, [&](int e) { return ldg(ag + e); });
        __syncthreads();
    }
    const bool flat_v = q.pitch_v == (int64_t)n_v, flat_o = q.pitch_out == (int64_t)n_v;
    const int cnt = nrow * n_v;
    int r = tid / n_v, i = tid - r * n_v;
    for (int e = tid; e < cnt; e += SU_THREADS) {
        const int64_t row = row0 + r;
        const T key = q.pitch_v ? (flat_v ? ldg(vg + row0 * n_v + e) : ldg(vg + row * q.pitch_v + i)) : ldg(vg + i);
        int idx;
        if constexpr (STAGE) {
            const T *const ar = q.pitch_a ? lds + (size_t)r * n_a : lds;
            idx = q.right ? ss_right(ar, n_a, key) : ss_left(ar, n_a, key);
        } else {
            const T *const ar = ag + row * q.pitch_a;
            idx = q.right ? ss_right(ar, n_a, key) : ss_left(ar, n_a, key);
        }
        stg<WT>(flat_o ? q.out + row0 * n_v + e : q.out + row * q.pitch_out + i, (int64_t)idx);
        i += SU_THREADS;
        while (i >= n_v) { i -= n_v; ++r; }
    }
}

// ---- integral / interp_c / interp_rho --------------------------------------------------------------------------------
enum { SU_INTERP_C = 0, SU_INTERP_RHO = 1, SU_INTEGRAL = 2 };

struct SuCoarseP {
    int64_t n_rows, pitch_Zh, pitch_zh, pitch_q, pitch_out;
    int nG, nL, mode, rb;
    const void *Zh, *zh, *q, *rho;
    void *out;
};

// two numpy-ordered sums run side by side: vn_leaf / vn_pw are generic over the term's type
template <typename T> struct Pair2 {
    T a, b;
    __device__ __forceinline__ Pair2() {}
    __device__ __forceinline__ explicit Pair2(int) : a(T(0)), b(T(0)) {}
    __device__ __forceinline__ Pair2(T a_, T b_) : a(a_), b(b_) {}
    __device__ __forceinline__ Pair2 &operator+=(const Pair2 &o) { a += o.a; b += o.b; return *this; }
    friend __device__ __forceinline__ Pair2 operator+(Pair2 x, const Pair2 &y) { x.a += y.a; x.b += y.b; return x; }
};

template <int PD, typename F> __device__ __forceinline__ auto su_npsum(const F &term, int cnt) -> decltype(term(0))
{
    using V = decltype(term(0));
    if constexpr (PD >= 0) return V(0) + vn_pw<PD>(term, 0, cnt);
    else return cnt <= 128 ? V(0) + vn_leaf(term, 0, cnt) : vn_npsum(term, cnt);
}

// integral() of splib/sputils.py:94-161 over the layers [Zh[k+1], Zh[k]] of RB rows.  While staging, every CELL i of a row
// gets its terms once: tn[i] = (w[i] q[i]) (z[i+1] - z[i]) (sputils.py:154; q[i] dz without weights, :146) and
// td[i] = w[i] dz (:159) -- the very products the reference's temporaries hold -- so that a layer's sums are additions of
// LDS values in ndarray.sum() order (pairwise recursion unrolled to the depth PD the host derived from nL, as K4; PD = -1:
// explicit stack).  The two edge pieces read q / w of the first and last cell from global memory (lines this workgroup has
// just loaded).  STAGE = false (rows beyond the LDS): terms formed on the fly from global memory.
template <typename T, int PD, bool STAGE, bool WEIGHTED, int WT> __global__ __launch_bounds__(SU_THREADS) void k_interp_c(const SuCoarseP p)
{
    T *const lds = reinterpret_cast<T *>(spc_smem);
    const int64_t row0 = (int64_t)blockIdx.x * p.rb;
    const int nrow = (int)((p.n_rows - row0) < p.rb ? (p.n_rows - row0) : p.rb);
    const int nL = p.nL, nG = p.nG, nc = nL - 1, tid = threadIdx.x;                    // nL points bound nc cells
    const T *const zg = (const T *)p.zh, *const qg = (const T *)p.q, *const wg = (const T *)p.rho;
    // LDS: tn[rb][nc] | td[rb][nc] (WEIGHTED) | z[rb][nL] (or z[nL] when shared)
    T *const ltn = lds, *const ltd = lds + (size_t)p.rb * nc, *const lz = ltd + (WEIGHTED ? (size_t)p.rb * nc : 0);
    if constexpr (STAGE) {
        const int total = nrow * nc;
        int r0 = tid / nc, l0 = tid - r0 * nc;                 // (row, cell) of element e0; stepped, not divided
        for (int e0 = tid; e0 < total; e0 += SU_THREADS * 2) {
            T qv[2], wv[2], z0[2], z1[2];
            int rr = r0, ll = l0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool in = e0 + u * SU_THREADS < total;
                const int r = in ? rr : 0, l = in ? ll : 0;
                const int64_t o = (row0 + r) * p.pitch_q + l;
                const int64_t oz = p.pitch_zh ? (row0 + r) * p.pitch_zh + l : (int64_t)l;
                qv[u] = ldg(qg + o);
                wv[u] = WEIGHTED ? ldg(wg + o) : T(1);
                z0[u] = ldg(zg + oz);
                z1[u] = ldg(zg + oz + 1);
                ll += SU_THREADS;
                while (ll >= nc) { ll -= nc; ++rr; }
            }
            r0 = rr; l0 = ll;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = e0 + u * SU_THREADS;
                if (e < total) {
                    const T dz = z1[u] - z0[u];
                    ltn[e] = WEIGHTED ? (wv[u] * qv[u]) * dz : qv[u] * dz;             // sputils.py:154 / 146
                    if constexpr (WEIGHTED) ltd[e] = wv[u] * dz;                       // sputils.py:159
                }
            }
        }
        if (p.pitch_zh) su_stage<2>(lz, nrow * nL, tid, su_slab<T>(p.zh, row0, p.pitch_zh, nL));
        else su_stage<1>(lz, nL, tid, [&](int e) { return ldg(zg + e); });
        __syncthreads();
    }
    const int cnt_out = nrow * nG;
    int r = tid / nG, k = tid - r * nG;
    for (int e = tid; e < cnt_out; e += SU_THREADS) {
        const int64_t row = row0 + r;
        const T *const z = STAGE ? lz + (p.pitch_zh ? (size_t)r * nL : 0) : zg + row * p.pitch_zh;
        const T *const Zh = (const T *)p.Zh + row * p.pitch_Zh;
        const T top = ldg(Zh + k), bot = ldg(Zh + k + 1);
        T res = T(0);                                                                  // Q = zeros / RHO = zeros
        if (p.mode == SU_INTEGRAL || top < z[nL - 1]) {                                // sputils.py:187 / 195
            T a = bot, b = top;                                                        // integral(ZZ[i+1], ZZ[i], ...)
            if (a < z[0] || a > z[nL - 1] || b < z[0] || b > z[nL - 1]) {              // sputils.py:113-115: None
                res = T(0) / T(0);                                                     // Q[i] = None stores NaN (numpy 2.x)
            } else {
                T sign = T(1);
                if (a > b) { sign = T(-1); const T t = a; a = b; b = t; }              // sputils.py:117-120
                const int ia = scan_cell(z, nL, a);                                    // sputils.py:122-124
                int ib = scan_cell(z, nL, b);                                          // sputils.py:125-127
                if (ib < ia) ib = ia;
                const int cnt = ib - ia + 1;
                const T da = a - z[ia], db = z[ib + 1] - b;
                const T *const qr = qg + row * p.pitch_q, *const wr = wg + row * p.pitch_q;
                const T qa = ldg(qr + ia), qb = ldg(qr + ib);
                T num, den = T(1);
                if constexpr (WEIGHTED) {
                    const T wa = ldg(wr + ia), wb = ldg(wr + ib);
                    Pair2<T> S;
                    if constexpr (STAGE) {
                        const T *const tn = ltn + (size_t)r * nc + ia, *const td = ltd + (size_t)r * nc + ia;
                        S = su_npsum<PD>([&](int i) { return Pair2<T>(tn[i], td[i]); }, cnt);
                    } else {
                        S = su_npsum<PD>([&](int i) {
                            const T dz = z[ia + i + 1] - z[ia + i];
                            return Pair2<T>((wr[ia + i] * qr[ia + i]) * dz, wr[ia + i] * dz); }, cnt);
                    }
                    num = (S.a - (wa * qa) * da) - (wb * qb) * db;                     // sputils.py:156-157
                    den = (S.b - wa * da) - wb * db;                                   // sputils.py:160-161
                    res = num / den * sign;                                            // sputils.py:162
                } else {
                    T S;
                    if constexpr (STAGE) {
                        const T *const tn = ltn + (size_t)r * nc + ia;
                        S = su_npsum<PD>([&](int i) { return tn[i]; }, cnt);
                    } else {
                        S = su_npsum<PD>([&](int i) { return qr[ia + i] * (z[ia + i + 1] - z[ia + i]); }, cnt);
                    }
                    num = (S - qa * da) - qb * db;                                     // sputils.py:149-152
                    res = num * sign;
                }
                if (p.mode == SU_INTERP_RHO) res = res / (top - bot);                   // sputils.py:196
            }
        }
        stg<WT>((T *)p.out + row * p.pitch_out + k, res);
        k += SU_THREADS;
        while (k >= nG) { k -= nG; ++r; }
    }
}

// ---- rms -----------------------------------------------------------------------------------------------------------
// sqrt(mean(a**2)) per row; numpy's mean = add.reduce (pairwise: blocks of <= 128 elements with 8 accumulators, halves
// split at multiples of 8, 8192-element chunks) / n.  EIGHT LANES PER ROW: lane j of a row's group carries accumulator j of
// the current leaf -- r[j] += a[i + j]**2 for i = 8, 16, ... (numpy's unrolled loop) -- then ((r0+r1)+(r2+r3))+((r4+r5)+
// (r6+r7)) by shuffles, the leaf's tail and the tree of leaf sums on every lane alike (same operands, same order: the
// group stays converged).  A wave reads eight 64-byte segments per load instruction; no LDS.
template <typename T> __device__ __forceinline__ T su_leaf8(const T *a, int lo, int n, int j)
{
    if (n < 8) {                                             // numpy: plain loop
        T res = T(0);
        for (int i = 0; i < n; ++i) { const T v = ldg(a + lo + i); res += v * v; }
        return res;
    }
    const int n8 = n - (n % 8);
    T v0 = ldg(a + lo + j);
    T rj = v0 * v0;
    int i = 8;
    for (; i + 24 < n8; i += 32) {                           // four loads in flight per lane
        const T x0 = ldg(a + lo + i + j), x1 = ldg(a + lo + i + 8 + j), x2 = ldg(a + lo + i + 16 + j), x3 = ldg(a + lo + i + 24 + j);
        rj += x0 * x0; rj += x1 * x1; rj += x2 * x2; rj += x3 * x3;
    }
    for (; i < n8; i += 8) { const T x = ldg(a + lo + i + j); rj += x * x; }
    // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)): lane j ^ 1, then ^ 2, then ^ 4 (IEEE addition commutes)
    T s = rj + __shfl_xor(rj, 1);
    s = s + __shfl_xor(s, 2);
    s = s + __shfl_xor(s, 4);
    for (i = n8; i < n; ++i) { const T x = ldg(a + lo + i); s += x * x; }
    return s;
}

// numpy's pairwise recursion over a chunk with its depth fixed at compile time (as vn_pw), leaves evaluated by the 8 lanes
template <int D, typename T> __device__ __forceinline__ T su_pw8(const T *a, int lo, int n, int j)
{
    if constexpr (D == 0) {
        return su_leaf8(a, lo, n, j);
    } else {
        if (n <= 128) return su_leaf8(a, lo, n, j);
        int n2 = n / 2;
        n2 -= n2 % 8;
        return su_pw8<D - 1>(a, lo, n2, j) + su_pw8<D - 1>(a, lo + n2, n - n2, j);
    }
}

// PD: depth of the recursion for rows of n <= 8192 elements as the host derived it (0: n <= 128; 1, 2, 3: up to 248 / 488 /
// 968); PD = -1: any n, explicit stack (vn_npsum's walk) and 8192-element chunks
template <typename T, int PD, int WT> __global__ __launch_bounds__(SU_THREADS) void k_rms(int64_t n_rows, int n, int64_t pitch, const T *a, T *out)
{
    const int64_t row = (int64_t)blockIdx.x * (SU_THREADS / 8) + (threadIdx.x >> 3);
    const int j = threadIdx.x & 7;
    const bool live = row < n_rows;                          // dead groups walk along on row 0 (shuffles need every lane)
    const T *const ar = a + (live ? row : 0) * pitch;
    T total = T(0);
    if constexpr (PD >= 0) {
        total += su_pw8<PD>(ar, 0, n, j);                    // ndarray.sum(): 0.0 + the one chunk
    } else {
        for (int c0 = 0; c0 < n; c0 += 8192) {               // 0.0 + chunk sums
            int cur_lo = c0, cur_n = (n - c0) < 8192 ? (n - c0) : 8192;
            int r_lo[10], r_n[10], depth = 0;
            T left[10];
            bool has_left[10];
            T v;
            for (;;) {
                while (cur_n > 128) {                        // descend into the left halves (vn_npsum's walk)
                    int n2 = cur_n / 2;
                    n2 -= n2 % 8;
                    r_lo[depth] = cur_lo + n2; r_n[depth] = cur_n - n2; has_left[depth] = false;
                    ++depth;
                    cur_n = n2;
                }
                v = su_leaf8(ar, cur_lo, cur_n, j);
                while (depth > 0 && has_left[depth - 1]) { v = left[depth - 1] + v; --depth; }
                if (depth == 0) break;
                left[depth - 1] = v; has_left[depth - 1] = true;
                cur_lo = r_lo[depth - 1]; cur_n = r_n[depth - 1];
            }
            total += v;
        }
    }
    if (live && j == 0) stg<WT>(out + row, sqrt(total / (T)n));
}
