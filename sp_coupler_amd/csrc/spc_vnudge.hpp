// spc_vnudge.hpp -- variability nudge (qt_forcing == 'variance'): splib/spcpl.py:613-744.  Shared pieces of K6.
//
// For every LES level k of every column the reference solves, with scipy.optimize.brentq,
//     mean_ij( max(beta (qt - qt_av) + qt_av - qsat, 0) ) = ql_ref[k]          (multiplicative, spcpl.py:646-648)
// or  mean_ij( max(qt + a R - qsat, 0) ) = ql_ref[k]                           (additive noise,  spcpl.py:653-656)
// over the horizontal plane (itot x jtot) of the LES's 3-D fields, then rescales / perturbs qt (and, with
// constantT, corrects thl).  The plane sum reproduces numpy's ndarray.sum() (pairwise blocks of 128 with 8
// accumulators, halves split at multiples of 8, chunks of 8192, result = 0.0 + chunk sums) and the root finder is
// scipy's brentq.c restated statement by statement (oracle/vnudge_oracle.py holds the same restatements and checks
// them against scipy / numpy bit for bit), so beta, a and the updated qt are BIT-identical to the NumPy/SciPy
// evaluation; thl (through exner's pow) agrees to a few ulp.  This file: the numpy-ordered sums (vn_leaf, vn_npsum --
// also used by K4), brentq as a resumable step function, and the host-side flattening of numpy's pairwise tree; the
// kernels are in spc_vnudge2.hpp.  (Rounds 1-2 also had a kernel that swept the planes from memory in every evaluation;
// k_vnudge_solve<true> replaced it: 7-9 x faster on 128 x 128 and 256 x 256 planes, profiles/r03_k6.log.)
// Included by spc_hip.hip.
#pragma once

#ifndef VN_LEAF_UNROLL
#define VN_LEAF_UNROLL 3      // K6: x 8 terms whose loads are issued together (its plane sweeps are load-latency bound)
#endif

struct VnP {
    int64_t n_cols;
    int nij, ktot, constantT, pad;
    const double *qsat, *R, *ql_av, *qt_av, *presf, *ql_ref, *ql;
    double *qt, *thl, *beta, *a_add, *qt_std;
    int32_t *status;
};

enum { VN_NONE = 0, VN_MULT = 1, VN_UNSAT = 2, VN_ADD = 4, VN_ADD_SKIPPED = 8, VN_NO_BRACKET = 16,
       VN_ERR_SIGN = 256, VN_ERR_CONV = 512 };

// numpy pairwise_sum over elements [lo, lo+n) of term(ij), n <= 128: 8 accumulators, then the tail
template <int UNROLL = 1, typename F> __device__ __forceinline__ auto vn_leaf(const F &term, int lo, int n) -> decltype(term(0))
{
    using T = decltype(term(0));
    if (n < 8) {
        T res = T(0);
        for (int i = 0; i < n; ++i) res += term(lo + i);
        return res;
    }
    T r0 = term(lo), r1 = term(lo + 1), r2 = term(lo + 2), r3 = term(lo + 3), r4 = term(lo + 4), r5 = term(lo + 5),
      r6 = term(lo + 6), r7 = term(lo + 7);
    int i = 8;
#pragma unroll UNROLL
    for (; i < n - (n % 8); i += 8) {
        r0 += term(lo + i); r1 += term(lo + i + 1); r2 += term(lo + i + 2); r3 += term(lo + i + 3);
        r4 += term(lo + i + 4); r5 += term(lo + i + 5); r6 += term(lo + i + 6); r7 += term(lo + i + 7);
    }
    T res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += term(lo + i);
    return res;
}

// ndarray.sum() of term(0..n-1): 0.0 + sum over 8192-element chunks of the pairwise recursion
// (pairwise_sum(a, n): n <= 128 -> leaf; else n2 = n/2 - (n/2)%8, pairwise(a, n2) + pairwise(a+n2, n-n2)),
// evaluated depth first with an explicit stack (depth <= 7 inside a chunk: 8191 -> 4103 -> 2055 -> 1031 -> 519 -> 263 -> 135 -> 71).
template <typename F> __device__ __forceinline__ auto vn_npsum(const F &term, int n) -> decltype(term(0))
{
    using T = decltype(term(0));
    T total = T(0);
    for (int c0 = 0; c0 < n; c0 += 8192) {
        int cur_lo = c0, cur_n = (n - c0) < 8192 ? (n - c0) : 8192;
        int r_lo[10], r_n[10], depth = 0;
        T left[10];
        bool has_left[10];
        T v;
        for (;;) {
            while (cur_n > 128) {                       // descend into the left halves
                int n2 = cur_n / 2;
                n2 -= n2 % 8;
                r_lo[depth] = cur_lo + n2; r_n[depth] = cur_n - n2; has_left[depth] = false;
                ++depth;
                cur_n = n2;
            }
            v = vn_leaf(term, cur_lo, cur_n);
            while (depth > 0 && has_left[depth - 1]) {  // right child done: node = left + right
                v = left[depth - 1] + v;
                --depth;
            }
            if (depth == 0) break;
            left[depth - 1] = v; has_left[depth - 1] = true;       // left child done: go right
            cur_lo = r_lo[depth - 1]; cur_n = r_n[depth - 1];
        }
        total += v;
    }
    return total;
}

// the same, kept out of line: callers whose sums almost never exceed one 128-element block (K4: cells per GCM layer)
// take vn_leaf directly and pay the recursion's registers and scratch only in the rare case
template <typename F> __device__ __attribute__((noinline)) auto vn_npsum_outlined(const F &term, int n) -> decltype(term(0))
{
    return vn_npsum(term, n);
}

constexpr int VN_MAXLEAF = 128;      // leaves of one 8192-element chunk (every leaf of a split node has >= 64 elements)

// scipy brentq.c as a resumable step function.  vn_brent_next() runs the top of one loop iteration: it either
// finishes (returns 1 converged / 2 sign error / 3 no convergence, *root set) or leaves the next abscissa in b.xcur
// (returns 0); the caller evaluates f there, stores it in b.fcur and calls again.
struct VnBrent {
    double xpre, xcur, xblk, fpre, fcur, fblk, spre, scur;
    int it;
};

__device__ __forceinline__ int vn_brent_next(VnBrent &b, double *root)
{
    const double xtol = 2e-12, rtol = 8.881784197001252e-16;
    if (b.it >= 100) { *root = b.xcur; return 3; }
    if (b.fpre != 0 && b.fcur != 0 && (signbit(b.fpre) != signbit(b.fcur))) {
        b.xblk = b.xpre; b.fblk = b.fpre; b.spre = b.scur = b.xcur - b.xpre;
    }
    if (fabs(b.fblk) < fabs(b.fcur)) {
        b.xpre = b.xcur; b.xcur = b.xblk; b.xblk = b.xpre;
        b.fpre = b.fcur; b.fcur = b.fblk; b.fblk = b.fpre;
    }
    const double delta = (xtol + rtol * fabs(b.xcur)) / 2;
    const double sbis = (b.xblk - b.xcur) / 2;
    if (b.fcur == 0 || fabs(sbis) < delta) { *root = b.xcur; return 1; }
    if (fabs(b.spre) > delta && fabs(b.fcur) < fabs(b.fpre)) {
        double stry;
        if (b.xpre == b.xblk) {
            stry = -b.fcur * (b.xcur - b.xpre) / (b.fcur - b.fpre);
        } else {
            const double dpre = (b.fpre - b.fcur) / (b.xpre - b.xcur), dblk = (b.fblk - b.fcur) / (b.xblk - b.xcur);
            stry = -b.fcur * (b.fblk * dblk - b.fpre * dpre) / (dblk * dpre * (b.fblk - b.fpre));
        }
        const double lim = fmin(fabs(b.spre), 3 * fabs(sbis) - delta);
        if (2 * fabs(stry) < lim) { b.spre = b.scur; b.scur = stry; }
        else { b.spre = sbis; b.scur = sbis; }
    } else {
        b.spre = sbis; b.scur = sbis;
    }
    b.xpre = b.xcur; b.fpre = b.fcur;
    if (fabs(b.scur) > delta) b.xcur += b.scur;
    else b.xcur += (sbis > 0 ? delta : -delta);
    ++b.it;
    return 0;
}

__device__ __forceinline__ int vn_brent_start(VnBrent &b, double xa, double xb, double fa, double fb, double *root)
{
    b.xpre = xa; b.xcur = xb; b.xblk = 0.0; b.fpre = fa; b.fcur = fb; b.fblk = 0.0; b.spre = 0.0; b.scur = 0.0; b.it = 0;
    if (fa == 0) { *root = xa; return 1; }
    if (fb == 0) { *root = xb; return 1; }
    if (signbit(fa) == signbit(fb)) { *root = 0.0; return 2; }
    return vn_brent_next(b, root);
}

// Flattens numpy's pairwise recursion over a chunk of cn <= 8192 elements: leaves (lo, n) in order, and the combine
// program in post-order -- step t: slot[pl[t]] = slot[pl[t]] + slot[pr[t]] (node = left + right; a subtree's value
// lives in the slot of its first leaf) -- so that the sum ends in slot 0.
__host__ __device__ inline void vn_build_tree(int cn, unsigned short *lo, unsigned short *n, unsigned short *pl, unsigned short *pr, int *nleaf)
{
    int r_n[10], lslot[10], rslot[10], depth = 0, cur_lo = 0, cur_n = cn, li = 0, np = 0;
    bool has_left[10];
    for (;;) {
        while (cur_n > 128) {
            int n2 = cur_n / 2;
            n2 -= n2 % 8;
            r_n[depth] = cur_n - n2; has_left[depth] = false; lslot[depth] = li; rslot[depth] = 0;
            ++depth;
            cur_n = n2;
        }
        lo[li] = (unsigned short)cur_lo; n[li] = (unsigned short)cur_n;
        cur_lo += cur_n;
        ++li;
        while (depth > 0 && has_left[depth - 1]) {
            --depth;
            pl[np] = (unsigned short)lslot[depth]; pr[np] = (unsigned short)rslot[depth];
            ++np;
        }
        if (depth == 0) break;
        has_left[depth - 1] = true; rslot[depth - 1] = li;
        cur_n = r_n[depth - 1];
    }
    *nleaf = li;
}

enum { VS_DONE = 0, VS_M0, VS_M1, VS_MB, VS_A0, VS_A1, VS_AB };
