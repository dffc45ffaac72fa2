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
// Round 4 (second generation; counters of the first and of this one in profiles/r04_k7_counters.log): the first generation moved exactly its
// algorithmic bytes (PMC traffic 1.00-1.06 x) but slowly -- its LDS staging went through ONE pointer that could be LDS or
// global, so every access was a FLAT one (3 LDS instructions per wave, 30 flat loads: interp); interp_c summed w q dz with
// 3-4 LDS reads and 2 multiplications per term in two passes (1 279 VALU instructions per wave); rms gave a whole row to
// one thread (559 waves on 1 024 SIMDs); exner looped 3 x over a capped grid with one pow in flight per thread.  Now:
//   * staging is a compile-time switch (STAGE) everywhere: LDS pointers are LDS pointers; rows that do not fit read global;
//   * a workgroup owns a SLAB of rows (su_rows: ~1 000-1 300 outputs, as K1's 8-column slabs) and loads it with flat,
//     coalesced accesses, all of a thread's loads issued before the first LDS store;
//   * launches that write <= 64 MiB store write-through (WT; su_write_through in spc_sputils_host.hpp -- K1 draws that line
//     at 32 MiB, K3 at 14: nothing left dirty in L2 at the end);
//   * interp_c forms the per-CELL terms (w q) dz and w dz once while staging; a layer's two sums are then pure additions
//     of LDS values, run side by side in ONE numpy-ordered pass (Pair2);
//   * rms: 8 lanes per row = the 8 accumulators of numpy's leaf, combined by shuffles in numpy's order (64-B segments,
//     no LDS, 8 x the parallelism per row);
//   * exner: 4 elements per thread, loads up front, 4 independent pow chains, one pass over an uncapped grid; the pow's 21
//     polynomial coefficients come from a __constant__ table, i.e. scalar registers (spc_exner_pow: a 64-bit literal costs
//     two v_mov per use, 26 % of the instructions of a pow; the operator is bound by VALU issue);
//   * the searches run on NaN-padded LDS rows with a fixed trip count (su_count, spc_hip.hip), addresses are a uniform base + a
//     32-bit byte offset (su_at).
#pragma once

constexpr int SU_THREADS = 256;
constexpr size_t SU_MAX_LDS = 64 * 1024;

// element `off` (a 32-bit count off a wave-uniform base) addressed as base + zero-extended BYTE offset: the form the
// backend turns into one global_load / global_store with a scalar base and a 32-bit VGPR offset (no 64-bit VALU arithmetic)
template <typename T> __device__ __forceinline__ const T *su_at(const T *base, int off)
{
    return (const T *)((const char *)base + (unsigned)(off * (int)sizeof(T)));
}
template <typename T> __device__ __forceinline__ T *su_at(T *base, int off)
{
    return (T *)((char *)base + (unsigned)(off * (int)sizeof(T)));
}

// r * pitch for a slab row r < 64 and a pitch below 2^24 elements (su_pitch_ok): the full-rate 24-bit multiply -- a
// v_mul_lo_u32 occupies the SIMD four times as long, and the standalone operators are bound by VALU issue
__device__ __forceinline__ int su_mul(int r, int pitch) { return (int)__umul24((unsigned)r, (unsigned)pitch); }

// The (row, entry) of a thread's outputs: output e = tid + k * SU_THREADS of a slab of rows with n entries each.  The step
// between two outputs of a thread is the same for all threads -- (dr, di) = divmod(SU_THREADS, n) plus one carry -- so no
// division and no loop per output.
struct SuWalk {
    int r, i, n, dr, di;
    __device__ __forceinline__ SuWalk(int tid, int n_) : n(n_)
    {
        // tid / n for tid < 256 without a vector division: (tid * ceil(2^16 / n)) >> 16 is exact while 255 n < 2^16
        const int m = n < SU_THREADS ? (65536 + n - 1) / n : 0;  // uniform
        r = n < SU_THREADS ? (int)(__umul24((unsigned)tid, (unsigned)m) >> 16) : (tid >= n ? 1 : 0);
        i = tid - su_mul(r, n);
        dr = SU_THREADS / n; di = SU_THREADS - dr * n;
    }
    __device__ __forceinline__ void next()
    {
        i += di; r += dr;
        if (i >= n) { i -= n; ++r; }
    }
};

// ---- staging: the [nrow x n] slab at `base` (rows `pitch` elements apart) into LDS dst[r * n + j]; U loads of a thread in
// flight before its first store; flat when the rows are contiguous, else (row, entry) stepped
template <int U, typename T> __device__ __forceinline__ void su_stage_slab(T *dst, int nrow, int n, int tid, const T *base, int pitch)
{
    const int cnt = nrow * n;
    const bool flat = pitch == n;                             // uniform
    SuWalk w(tid, n);
    for (int e0 = tid; e0 < cnt; e0 += SU_THREADS * U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * SU_THREADS;
            v[u] = e < cnt ? ldg(su_at(base, flat ? e : su_mul(w.r, pitch) + w.i)) : T(0);
            w.next();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * SU_THREADS;
            if (e < cnt) dst[e] = v[u];
        }
    }
}

// (su_pad / su_seek / su_count -- the fixed-trip searches on NaN-padded LDS rows -- live in spc_hip.hip: K4 uses them too)

// nrow rows of n elements (row r at src + r * pitch) into LDS rows of `stride` entries, the tail of every row NaN;
// U entries of a thread in flight before its first LDS store; (row, entry) stepped, never divided
template <int U, typename T> __device__ __forceinline__ void su_stage_rows(T *dst, int nrow, int n, int stride, int tid, const T *src, int pitch)
{
    const int total = nrow * stride;
    const T nan = T(0) / T(0);
    SuWalk w(tid, stride);                                    // (row, entry) of LDS element t0 + u * SU_THREADS
    for (int t0 = tid; t0 < total; t0 += SU_THREADS * U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = (t0 + u * SU_THREADS < total && w.i < n) ? ldg(su_at(src, su_mul(w.r, pitch) + w.i)) : nan;   // uniform base + 32-bit offset
            w.next();
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (t0 + u * SU_THREADS < total) dst[t0 + u * SU_THREADS] = v[u];
    }
}

// numpy.interp at one point x on a staged row: xp NaN-padded for su_seek, fp alongside.  The standalone interp is bound
// by VALU issue (K1 / K3, which are not, keep bracket() / lerp_np as they are), so the common case -- x inside the
// table -- runs straight through: clamp the cell, read its four corners, one IEEE division, and SELECT the end values
// (x < xp[0] -> fp[0]; x >= xp[n-1] -> fp[n-1]; x == xp[j] -> fp[j]: numpy's own order of tests); NaN (a NaN x, or numpy's
// fallbacks for a non-finite product) is a rare branch.
template <int SL, typename T> __device__ __forceinline__ T su_interp_pad(const T *xp, const T *fp, int n, int p2, T x)
{
    if (n == 1) return fp[0];                                                       // numpy lenxp == 1: fp[0], NaN x included
    const T *const e = su_seek<SL>(xp, p2, [&](T v) { return v <= x; });           // first entry > x (or the NaN padding)
    const int cnt = (int)(((unsigned)(size_t)e - (unsigned)(size_t)xp) / (unsigned)sizeof(T));   // entries <= x
    int j = cnt - 1;
    j = j < 0 ? 0 : j;
    j = j > n - 2 ? n - 2 : j;
    const T x0 = xp[j], x1 = xp[j + 1], f0 = fp[j], f1 = fp[j + 1];
    const T slope = (f1 - f0) / (x1 - x0);
    T r = slope * (x - x0) + f0;
    const bool low = cnt == 0 || x0 == x, high = cnt >= n;
    if (__builtin_expect(r != r && !low && !high, 0)) {                              // numpy's fallbacks, in its order
        r = slope * (x - x1) + f1;
        if (r != r && f0 == f1) r = f0;
    }
    r = low ? f0 : r;
    r = high ? f1 : r;
    if (__builtin_expect(x != x, 0)) r = x;                                          // a NaN x counts nothing: numpy returns it
    return r;
}

// ---- exner ---------------------------------------------------------------------------------------------------------
#ifndef SU_EX_PER        // elements per thread: 2 -> 11.1-11.3 us at 35 718 x 91 elements, 4 -> 11.4-11.5, 6 -> 12.2 (profiles/r04_k7_slab_sweep.log)
#define SU_EX_PER 2
#endif
template <typename T, int WT> __global__ __launch_bounds__(SU_THREADS) void k_exner(int64_t n, const T *p, T *out, int inverse)
{
    constexpr int PER = SU_EX_PER;
    const T y = inverse ? (-K<T>::rd) / K<T>::cp : K<T>::rd / K<T>::cp;                  // sputils.py:34 / 29
    const int64_t b0 = (int64_t)blockIdx.x * (SU_THREADS * PER);                         // the workgroup's first element: uniform
    const int left = (int)((n - b0) < SU_THREADS * PER ? (n - b0) : SU_THREADS * PER);
    const T *const pb = p + b0;                                                          // uniform bases + 32-bit offsets
    T *const ob = out + b0;
    T v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = threadIdx.x + u * SU_THREADS;
        v[u] = i < left ? ldg(su_at(pb, i)) : T(1);
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) v[u] = spc_exner_pow(v[u], y);      // Markstein quotient, coefficients as scalar operands
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = threadIdx.x + u * SU_THREADS;
        if (i < left) stg<WT>(su_at(ob, i), v[u]);
    }
}

// ---- interp --------------------------------------------------------------------------------------------------------
struct SuInterpP {
    int64_t n_rows, pitch_x, pitch_xp, pitch_fp, pitch_out;
    int n_x, n_xp, p2, rb;             // rb rows per workgroup
    const void *x, *xp, *fp;
    void *out;
};

// SL >= 0: the slab's sample arrays go through LDS: fp[rb][n_xp] | xp rows padded for su_count ([rb] of them, or one when
// xp is shared)
template <typename T, int SL, int WT> __global__ __launch_bounds__(SU_THREADS) void k_interp(const SuInterpP q)
{
    constexpr bool STAGE = SL >= 0;
    T *const lds = reinterpret_cast<T *>(spc_smem);
    const int64_t row0 = (int64_t)blockIdx.x * q.rb;
    const int nrow = (int)((q.n_rows - row0) < q.rb ? (q.n_rows - row0) : q.rb);
    const int n_xp = q.n_xp, n_x = q.n_x, tid = threadIdx.x, stride = su_pad(q.p2);
    const T *const xg = (const T *)q.x, *const xpg = (const T *)q.xp, *const fpg = (const T *)q.fp;
    T *const out = (T *)q.out;
    T *const lfp = lds, *const lxp = lds + (size_t)q.rb * n_xp;
    // addresses: the slab's (uniform) base + a 32-bit element offset r * pitch + i (the host keeps rb * pitch below 2^31)
    const T *const xb = xg + row0 * q.pitch_x;
    T *const ob = out + row0 * q.pitch_out;
    const int px = (int)q.pitch_x, po = (int)q.pitch_out;
    const int cnt = nrow * n_x;
    SuWalk w(tid, n_x);                                       // (row, point) of this thread's outputs
    // a thread's x values are loaded one output AHEAD (the first before the staging): the load's latency sits behind the
    // evaluation of the previous output instead of in front of every search
    T xn = tid < cnt ? ldg(su_at(xb, su_mul(w.r, px) + w.i)) : T(0);
    if constexpr (STAGE) {
        su_stage_slab<3>(lfp, nrow, n_xp, tid, fpg + row0 * q.pitch_fp, (int)q.pitch_fp);
        if (q.pitch_xp) su_stage_rows<3>(lxp, nrow, n_xp, stride, tid, xpg + row0 * q.pitch_xp, (int)q.pitch_xp);
        else su_stage_rows<2>(lxp, 1, n_xp, stride, tid, xpg, 0);
        __syncthreads();
    }
    for (int e = tid; e < cnt; e += SU_THREADS) {
        const int r = w.r, i = w.i;
        const T xv = xn;
        w.next();
        if (e + SU_THREADS < cnt) xn = ldg(su_at(xb, su_mul(w.r, px) + w.i));
        T res;
        if constexpr (STAGE) {
            const T *const xpr = q.pitch_xp ? lxp + su_mul(r, stride) : lxp;
            res = su_interp_pad<SL>(xpr, lfp + su_mul(r, n_xp), n_xp, q.p2, xv);
        } else {
            const int64_t row = row0 + r;
            res = interp_at(bracket(xpg + row * q.pitch_xp, n_xp, q.p2, xv), fpg + row * q.pitch_fp);
        }
        stg<WT>(su_at(ob, su_mul(r, po) + i), res);
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
    int n_a, n_v, rb, right, p2;
    const void *a, *v;
    int64_t *out;
};

template <typename T, int SL, int WT> __global__ __launch_bounds__(SU_THREADS) void k_searchsorted(const SuSearchP q)
{
    constexpr bool STAGE = SL >= 0;
    T *const lds = reinterpret_cast<T *>(spc_smem);
    const int64_t row0 = (int64_t)blockIdx.x * q.rb;
    const int nrow = (int)((q.n_rows - row0) < q.rb ? (q.n_rows - row0) : q.rb);
    const T *const ag = (const T *)q.a, *const vg = (const T *)q.v;
    const int n_a = q.n_a, n_v = q.n_v, tid = threadIdx.x, stride = su_pad(q.p2);
    const T *const vb = vg + row0 * q.pitch_v;
    int64_t *const ob = q.out + row0 * q.pitch_out;
    const int pv = (int)q.pitch_v, po = (int)q.pitch_out;
    const int cnt = nrow * n_v;
    SuWalk w(tid, n_v);
    T kn = tid < cnt ? ldg(su_at(vb, su_mul(w.r, pv) + w.i)) : T(0);      // keys one output ahead, as k_interp's x
    if constexpr (STAGE) {
        if (q.pitch_a) su_stage_rows<3>(lds, nrow, n_a, stride, tid, ag + row0 * q.pitch_a, (int)q.pitch_a);
        else su_stage_rows<2>(lds, 1, n_a, stride, tid, ag, 0);
        __syncthreads();
    }
    for (int e = tid; e < cnt; e += SU_THREADS) {
        const int r = w.r, i = w.i;
        const T key = kn;
        w.next();
        if (e + SU_THREADS < cnt) kn = ldg(su_at(vb, su_mul(w.r, pv) + w.i));
        int idx;
        if constexpr (STAGE) {
            // the insertion point of a sorted row = the number of entries in front of it: side='right' those with
            // !(key < a[i]), side='left' those with a[i] < key, in numpy's NaN-last order (np_lt); a NaN key passes the
            // NaN padding too, hence the clamp
            const T *const ar = q.pitch_a ? lds + su_mul(r, stride) : lds;
            idx = q.right ? su_count<SL>(ar, q.p2, [&](T a) { return !np_lt(key, a); }) : su_count<SL>(ar, q.p2, [&](T a) { return np_lt(a, key); });
            idx = idx < n_a ? idx : n_a;
        } else {
            const T *const ar = ag + (row0 + r) * q.pitch_a;
            idx = q.right ? ss_right(ar, n_a, key) : ss_left(ar, n_a, key);
        }
        stg<WT>(su_at(ob, su_mul(r, po) + i), (int64_t)idx);
    }
}

// ---- integral / interp_c / interp_rho --------------------------------------------------------------------------------
enum { SU_INTERP_C = 0, SU_INTERP_RHO = 1, SU_INTEGRAL = 2 };

struct SuCoarseP {
    int64_t n_rows, pitch_Zh, pitch_zh, pitch_q, pitch_out;
    int nG, nL, mode, rb, p2;          // p2: largest power of two <= nL - 1 (the rows z[1:] the cell scans count over)
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
#ifndef SU_IC_WAVES      // waves per SIMD the register allocator is asked to fit: 5 = 96 VGPRs (15 spilled) runs 34 us at 35 718 rows,
#define SU_IC_WAVES 5    // 4 (116 VGPRs, no spills) 36 us, 6 (80 VGPRs, 37 spilled) 38 us (profiles/r04_k7_slab_sweep.log)
#endif
template <typename T, int PD, int SL, bool WEIGHTED, int WT> __global__ __launch_bounds__(SU_THREADS, SU_IC_WAVES) void k_interp_c(const SuCoarseP p)
{
    constexpr bool STAGE = SL >= 0;
    T *const lds = reinterpret_cast<T *>(spc_smem);
    const int64_t row0 = (int64_t)blockIdx.x * p.rb;
    const int nrow = (int)((p.n_rows - row0) < p.rb ? (p.n_rows - row0) : p.rb);
    const int nL = p.nL, nG = p.nG, nc = nL - 1, tid = threadIdx.x, zstride = su_pad(p.p2);     // nL points bound nc cells
    const T *const zg = (const T *)p.zh, *const qg = (const T *)p.q, *const wg = (const T *)p.rho;
    // LDS: tn[rb][nc] | td[rb][nc] (WEIGHTED) | Zh[rb][nG+1] | out[rb][nG] | z rows padded for su_count ([rb] of them, or one
    // when zh is shared)
    T *const ltn = lds, *const ltd = lds + (size_t)p.rb * nc, *const lZ = ltd + (WEIGHTED ? (size_t)p.rb * nc : 0);
    T *const lout = lZ + (size_t)p.rb * (nG + 1), *const lz = lout + (size_t)p.rb * nG;
    const T *const qb = qg + row0 * p.pitch_q, *const wb = wg + row0 * p.pitch_q, *const zb = zg + row0 * p.pitch_zh;    // slab bases
    const T *const Zb = (const T *)p.Zh + row0 * p.pitch_Zh;
    T *const ob = (T *)p.out + row0 * p.pitch_out;
    const int cnt_out = nrow * nG;
    // Outputs are walked LAYER-major -- output e is layer e / nrow of row e % nrow -- so that a wave holds a few consecutive layers
    // of all the slab's rows: the layers above the fine grid's top (most of them for interp_c on a column whose LES is shallow:
    // sputils.py:187 leaves them zero) then fill WHOLE waves, which skip the integral, instead of idling in every wave.  The
    // coarse levels Zh and the results go through LDS in their memory order (row-major, coalesced both ways).
    SuWalk w(tid, nrow);                                      // w.r: the layer k, w.i: the row
    if constexpr (STAGE) {
        const int total = nrow * nc;
        SuWalk sw(tid, nc);                                    // (row, cell) of element e0 + u * SU_THREADS
        for (int e0 = tid; e0 < total; e0 += SU_THREADS * 2) {
            T qv[2], wv[2], z0[2], z1[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool in = e0 + u * SU_THREADS < total;
                const int r = in ? sw.r : 0, l = in ? sw.i : 0;
                const int o = su_mul(r, (int)p.pitch_q) + l, oz = su_mul(r, (int)p.pitch_zh) + l;      // 32-bit offsets off the slab's bases
                qv[u] = ldg(su_at(qb, o));
                wv[u] = WEIGHTED ? ldg(su_at(wb, o)) : T(1);
                z0[u] = ldg(su_at(zb, oz));
                z1[u] = ldg(su_at(zb, oz + 1));
                sw.next();
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = e0 + u * SU_THREADS;
                if (e < total) {
                    const T dz = z1[u] - z0[u];
                    ltn[e] = (WEIGHTED && SPC_MUTANT != 7) ? (wv[u] * qv[u]) * dz : qv[u] * dz;             // sputils.py:154 / 146
                    if constexpr (WEIGHTED) ltd[e] = wv[u] * dz;                       // sputils.py:159
                }
            }
        }
        su_stage_slab<2>(lZ, nrow, nG + 1, tid, Zb, (int)p.pitch_Zh);
        if (p.pitch_zh) su_stage_rows<2>(lz, nrow, nL, zstride, tid, zb, (int)p.pitch_zh);
        else su_stage_rows<2>(lz, 1, nL, zstride, tid, zg, 0);
        __syncthreads();
    }
    for (int e = tid; e < cnt_out; e += SU_THREADS, w.next()) {
        const int r = w.i, k = w.r;
        T top, bot;
        if constexpr (STAGE) { const T *const Zr = lZ + su_mul(r, nG + 1) + k; top = Zr[0]; bot = Zr[1]; }
        else { const int oZ = su_mul(r, (int)p.pitch_Zh) + k; top = ldg(su_at(Zb, oZ)); bot = ldg(su_at(Zb, oZ + 1)); }
        const T *const z = STAGE ? lz + (p.pitch_zh ? su_mul(r, zstride) : 0) : zg + (row0 + r) * p.pitch_zh;
        T res = T(0);                                                                  // Q = zeros / RHO = zeros
        if (p.mode == SU_INTEGRAL || top < z[nL - 1]) {                                // sputils.py:187 / 195
            T a = bot, b = top;                                                        // integral(ZZ[i+1], ZZ[i], ...)
            if (a < z[0] || a > z[nL - 1] || b < z[0] || b > z[nL - 1]) {              // sputils.py:113-115: None
                res = T(0) / T(0);                                                     // Q[i] = None stores NaN (numpy 2.x)
            } else {
                T sign = T(1);
                if (a > b) { sign = T(-1); const T t = a; a = b; b = t; }              // sputils.py:117-120
                // the scans `while z[i+1] < a: i += 1` (sputils.py:122-127) = the number of k >= 1 with z[k] < a; z[nL-1] < a
                // is false (a <= z[nL-1] was checked), so the count stops at nL - 2 at the latest
                int ia, ib;
                if constexpr (STAGE) {
                    ia = su_count<SL>(z + 1, p.p2, [&](T zk) { return zk < a; });
                    ib = su_count<SL>(z + 1, p.p2, [&](T zk) { return zk < b; });
                } else {
                    ia = scan_cell(z, nL, a);
                    ib = scan_cell(z, nL, b);
                }
                if (ib < ia) ib = ia;
                const int cnt = ib - ia + 1;
                const T da = a - z[ia], db = z[ib + 1] - b;
                const T *const qr = qb + su_mul(r, (int)p.pitch_q), *const wr = wb + su_mul(r, (int)p.pitch_q);
                const T qa = ldg(qr + ia), qe = ldg(qr + ib);
                T num, den = T(1);
                if constexpr (WEIGHTED) {
                    const T wa = ldg(wr + ia), we = ldg(wr + ib);
                    Pair2<T> S;
                    if constexpr (STAGE) {
                        const T *const tn = ltn + su_mul(r, nc) + ia, *const td = ltd + su_mul(r, nc) + ia;
                        S = su_npsum<PD>([&](int i) { return Pair2<T>(tn[i], td[i]); }, cnt);
                    } else {
                        S = su_npsum<PD>([&](int i) {
                            const T dz = z[ia + i + 1] - z[ia + i];
                            return Pair2<T>((wr[ia + i] * qr[ia + i]) * dz, wr[ia + i] * dz); }, cnt);
                    }
                    num = (S.a - (wa * qa) * da) - (we * qe) * db;                     // sputils.py:156-157
                    den = (S.b - wa * da) - we * db;                                   // sputils.py:160-161
                    res = num / den * sign;                                            // sputils.py:162
                } else {
                    T S;
                    if constexpr (STAGE) {
                        const T *const tn = ltn + su_mul(r, nc) + ia;
                        S = su_npsum<PD>([&](int i) { return tn[i]; }, cnt);
                    } else {
                        S = su_npsum<PD>([&](int i) { return qr[ia + i] * (z[ia + i + 1] - z[ia + i]); }, cnt);
                    }
                    num = (S - qa * da) - qe * db;                                     // sputils.py:149-152
                    res = num * sign;
                }
                if (p.mode == SU_INTERP_RHO) res = res / (top - bot);                   // sputils.py:196
            }
        }
        if constexpr (STAGE) lout[su_mul(r, nG) + k] = res;
        else stg<WT>(su_at(ob, su_mul(r, (int)p.pitch_out) + k), res);
    }
    if constexpr (STAGE) {                                    // the slab's results, row-major: whole lines per store instruction
        __syncthreads();
        SuWalk ow(tid, nG);
        const bool flat = p.pitch_out == nG;
        for (int e = tid; e < cnt_out; e += SU_THREADS, ow.next())
            stg<WT>(su_at(ob, flat ? e : su_mul(ow.r, (int)p.pitch_out) + ow.i), lout[e]);
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
