// spc_vnudge.hpp -- variability nudge (qt_forcing == 'variance'): splib/spcpl.py:613-744.
//
// For every LES level k of every column the reference solves, with scipy.optimize.brentq,
//     mean_ij( max(beta (qt - qt_av) + qt_av - qsat, 0) ) = ql_ref[k]          (multiplicative, spcpl.py:646-648)
// or  mean_ij( max(qt + a R - qsat, 0) ) = ql_ref[k]                           (additive noise,  spcpl.py:653-656)
// over the horizontal plane (itot x jtot) of the LES's 3-D fields, then rescales / perturbs qt (and, with
// constantT, corrects thl).  One thread owns one (column, level): lanes run along k, the fastest index of the
// reference's [itot, jtot, k] field layout, so every plane sweep is a coalesced 8-B-per-lane stream; the plane
// sum reproduces numpy's ndarray.sum() (pairwise blocks of 128 with 8 accumulators, halves split at multiples of
// 8, chunks of 8192, result = 0.0 + chunk sums) and the root finder is scipy's brentq.c restated statement by
// statement (oracle/vnudge_oracle.py holds the same restatements and checks them against scipy / numpy bit for
// bit), so beta, a and the updated qt are BIT-identical to the NumPy/SciPy evaluation; thl (through exner's
// pow) agrees to a few ulp.  Included by spc_hip.hip.
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

// plane access of one (column, level): element ij of a [nij x ktot] slab, lanes along k
struct VnPlane {
    const double *qt, *qsat, *R;
    int64_t stride;      // ktot
    double qt_av;
    int nij;
};

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

// get_ql_diff(beta) (spcpl.py:646-648) / get_ql_diff_additive(a) (spcpl.py:653-656)
template <bool ADD> __device__ __forceinline__ double vn_ql_diff(const VnPlane &pl, double x, double ql_ref)
{
    const double s = vn_npsum([&](int ij) {
        const double q = pl.qt[(int64_t)ij * pl.stride], qs = pl.qsat[(int64_t)ij * pl.stride];
        const double t = ADD ? (q + (x * pl.R[ij])) - qs : ((x * (q - pl.qt_av)) + pl.qt_av) - qs;
        return (t >= 0.0 || t != t) ? t : 0.0;                           // numpy.maximum(t, 0): NaN propagates
    }, pl.nij);
    return s / (double)pl.nij - ql_ref;
}

// scipy/optimize/Zeros/brentq.c (scipy 1.15), xtol = 2e-12, rtol = 4 eps, maxiter = 100; err: 0 ok, 1 sign, 2 conv
template <bool ADD>
__device__ __forceinline__ double vn_brentq(const VnPlane &pl, double ql_ref, double xa, double xb, double fa, double fb, int *err)
{
    const double xtol = 2e-12, rtol = 8.881784197001252e-16;
    double xpre = xa, xcur = xb, xblk = 0.0, fpre = fa, fcur = fb, fblk = 0.0, spre = 0.0, scur = 0.0;
    *err = 0;
    if (fpre == 0) return xpre;
    if (fcur == 0) return xcur;
    if (signbit(fpre) == signbit(fcur)) { *err = 1; return 0.0; }
    for (int i = 0; i < 100; ++i) {
        if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre; fblk = fpre; spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        const double delta = (xtol + rtol * fabs(xcur)) / 2;
        const double sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            double stry;
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);                       // interpolate
            } else {
                const double dpre = (fpre - fcur) / (xpre - xcur), dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));   // extrapolate
            }
            const double lim = fmin(fabs(spre), 3 * fabs(sbis) - delta);
            if (2 * fabs(stry) < lim) { spre = scur; scur = stry; }                 // good short step
            else { spre = sbis; scur = sbis; }                                      // bisect
        } else {
            spre = sbis; scur = sbis;
        }
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0 ? delta : -delta);
        fcur = vn_ql_diff<ADD>(pl, xcur, ql_ref);
    }
    *err = 2;
    return xcur;
}

// ---- cooperative form: 32 slices (16-lane groups of 8 waves) share one (column, 16-level tile) ---------------------
// Lanes run along k (16 levels = one 128-B line of a plane row), and the plane's pairwise-sum LEAVES (<= 128 elements
// each, the unit numpy sums with 8 accumulators) are dealt round-robin to the 32 slices; each slice leaves its leaf sums in
// LDS and slice 0 combines them in numpy's tree order (the recursion is flattened once per workgroup into a leaf
// table and a post-order combine program, vn_build_tree), so the result is bit-identical to the serial evaluation while 8 x as many loads are in flight.  Wave 0 also owns the
// per-level state machines (bracket test, brentq restated as a resumable step function, additive fallback); every
// round evaluates ONE point per level, each level at its own abscissa and in its own mode.
constexpr int VN_WAVES = 8;
constexpr int VN_KT = 16;                        // levels per workgroup tile (128 contiguous bytes of a plane row)
constexpr int VN_NSL = VN_WAVES * 64 / VN_KT;    // slices: 16-lane groups, each summing its share of the leaves
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

__global__ __launch_bounds__(64 * VN_WAVES) void k_vnudge(const VnP p)
{
    __shared__ double s_leaf[VN_MAXLEAF][VN_KT];    // leaf sums of the current chunk / scratch of the argmax merge
    // numpy's pairwise tree of a chunk, flattened ONCE per workgroup (the recursion needs a stack, i.e. scratch
    // memory: walking it in every evaluation cost more than the sums): leaves (lo, n) and the combine program
    // "slot[pl] += slot[pr]" in post-order; shape 0 = a full 8192-element chunk, shape 1 = the last (or only) chunk
    __shared__ unsigned short s_lo[2][VN_MAXLEAF], s_n[2][VN_MAXLEAF], s_pl[2][VN_MAXLEAF], s_pr[2][VN_MAXLEAF];
    __shared__ int s_nleaf[2];
    __shared__ double s_x[VN_KT], s_coef[VN_KT], s_tc[VN_KT];
    __shared__ int s_mode[VN_KT], s_apply[VN_KT], s_touched[VN_KT], s_flag[2];
    // thread = (level kl of the tile, slice sl): 16 consecutive lanes read 128 contiguous bytes of a plane row, the
    // four 16-lane groups of a wave work on different leaves / rows; slice 0 owns the per-level state machines
    const int lane = threadIdx.x & (VN_KT - 1), sl = threadIdx.x / VN_KT;
    const bool own = sl == 0;
    const int k = blockIdx.x * VN_KT + lane;
    const int64_t col = blockIdx.y;
    const bool valid = k < p.ktot;
    const int kk = valid ? k : p.ktot - 1;                                   // in-bounds addresses for idle lanes
    const int64_t lev = col * p.ktot + kk, base = col * (int64_t)p.nij * p.ktot + kk, ks = p.ktot;
    double *const qt = p.qt + base;
    const double *const qsat = p.qsat + base, *const R = p.R + col * (int64_t)p.nij;
    const double qt_av = p.qt_av[lev], ql_ref = p.ql_ref[lev], ql_av = p.ql_av[lev];
    const int nij = p.nij;
    if (threadIdx.x < 2) {                                    // two lanes flatten the two chunk shapes
        const int shape = threadIdx.x;
        vn_build_tree(shape == 0 ? 8192 : (nij % 8192 ? nij % 8192 : 8192), s_lo[shape], s_n[shape], s_pl[shape], s_pr[shape],
                      &s_nleaf[shape]);
    }

    // ---- wave 0: per-level state ------------------------------------------------------------------------------
    int stage = VS_DONE, st = VN_NONE, err = 0, apply = 0;       // apply: 0 none, 1 multiplicative, 2 additive
    bool touched = false, want_argmax = false;
    double beta = 1.0, a = 0.0, f_lo = 0.0;
    VnBrent br = {};
    if (own) {
        if (valid) {
            if (ql_ref > 1e-9) { stage = VS_M0; touched = true; }                    // spcpl.py:665
            else if (ql_av > ql_ref) { want_argmax = true; touched = true; }         // spcpl.py:679
        }
        s_mode[lane] = want_argmax ? 3 : 0;
        if (lane == 0) s_flag[0] = 0;
    }
    __syncthreads();
    if (own && want_argmax) atomicOr(&s_flag[0], 1);
    __syncthreads();

    // ---- "barely unsaturated" branch (spcpl.py:679-695): numpy.argmax(qt - qsat) over the plane, first maximum,
    //      a NaN wins.  Each wave scans one contiguous segment, wave 0 merges the segments in order.
    if (s_flag[0]) {
        const int seg = (nij + VN_NSL - 1) / VN_NSL, lo = sl * seg, hi = (lo + seg) < nij ? (lo + seg) : nij;
        double bv = 0.0;
        int bi = -1;
        if (s_mode[lane] == 3 && lo < hi) {
            bi = lo; bv = qt[(int64_t)lo * ks] - qsat[(int64_t)lo * ks];
            for (int ij = lo + 1; ij < hi && !(bv != bv); ++ij) {
                const double v = qt[(int64_t)ij * ks] - qsat[(int64_t)ij * ks];
                if (v > bv || v != v) { bv = v; bi = ij; }
            }
        }
        s_leaf[sl][lane] = bv;
        s_leaf[VN_NSL + sl][lane] = (double)bi;
        __syncthreads();
        if (own && want_argmax) {
            double best = s_leaf[0][lane];
            int idx = (int)s_leaf[VN_NSL][lane];
            for (int q = 1; q < VN_NSL && !(best != best); ++q) {
                const int qi = (int)s_leaf[VN_NSL + q][lane];
                const double v = s_leaf[q][lane];
                if (qi >= 0 && (v > best || v != v)) { best = v; idx = qi; }
            }
            beta = (qsat[(int64_t)idx * ks] - qt_av) / (qt[(int64_t)idx * ks] - qt_av);      // spcpl.py:683
            if (beta < 0) beta = 1.0;                                                        // spcpl.py:692-695
            st = VN_UNSAT;
        }
        __syncthreads();
    }

    // decide what follows a known beta (spcpl.py:703-725); wave 0 only
    auto after_beta = [&]() {
        if (beta >= 5.0) {
            if (ql_ref > ql_av) { stage = VS_A0; }
            else { st |= VN_ADD_SKIPPED; beta = 1.0; stage = VS_DONE; }
        } else {
            apply = 1; stage = VS_DONE;
        }
    };
    if (own && want_argmax) after_beta();

    // ---- root-finding rounds: one evaluation per level and round ---------------------------------------------------
    for (;;) {
        if (own) {
            double x = 0.0;
            int mode = 0;
            switch (stage) {
            case VS_M0: x = 0.0; mode = 1; break;
            case VS_M1: x = 5.0; mode = 1; break;
            case VS_MB: x = br.xcur; mode = 1; break;
            case VS_A0: x = 0.0; mode = 2; break;
            case VS_A1: x = 5.0; mode = 2; break;
            case VS_AB: x = br.xcur; mode = 2; break;
            default: break;
            }
            s_x[lane] = x; s_mode[lane] = mode;
            if (lane == 0) s_flag[1] = 0;
        }
        __syncthreads();
        if (own && stage != VS_DONE) atomicOr(&s_flag[1], 1);
        __syncthreads();
        if (!s_flag[1]) break;
        const double x = s_x[lane];
        const int mode = s_mode[lane];
        double total = 0.0;
        for (int c0 = 0; c0 < nij; c0 += 8192) {                    // ndarray.sum(): 0.0 + chunk sums
            const int cn = (nij - c0) < 8192 ? (nij - c0) : 8192;
            const int shape = (c0 + cn < nij) ? 0 : 1;                  // the last (or only) chunk has its own shape
            if (mode != 0) {
                auto term = [&](int ij) {
                    const double q = qt[(int64_t)(c0 + ij) * ks], qs = qsat[(int64_t)(c0 + ij) * ks];
                    const double t = mode == 2 ? (q + (x * R[c0 + ij])) - qs : ((x * (q - qt_av)) + qt_av) - qs;
                    return (t >= 0.0 || t != t) ? t : 0.0;                           // numpy.maximum(t, 0)
                };
                for (int li = sl; li < s_nleaf[shape]; li += VN_NSL)
                    s_leaf[li][lane] = vn_leaf<VN_LEAF_UNROLL>(term, (int)s_lo[shape][li], (int)s_n[shape][li]);
            }
            __syncthreads();
            if (own && mode != 0) {
                for (int t = 0; t + 1 < s_nleaf[shape]; ++t) {
                    const int l = s_pl[shape][t], r = s_pr[shape][t];
                    s_leaf[l][lane] = s_leaf[l][lane] + s_leaf[r][lane];
                }
                total += s_leaf[0][lane];
            }
            __syncthreads();
        }
        if (own && stage != VS_DONE) {
            const double f = total / (double)nij - ql_ref;                           // spcpl.py:646-648 / 653-656
            double root = 0.0;
            int rc = 0;
            switch (stage) {
            case VS_M0: f_lo = f; stage = VS_M1; break;
            case VS_M1:
                if (f_lo > 0 || f < 0) { beta = 5.0; st = VN_NO_BRACKET; after_beta(); }          // spcpl.py:669-673
                else {
                    st = VN_MULT;
                    rc = vn_brent_start(br, 0.0, 5.0, f_lo, f, &root);
                    if (rc == 0) stage = VS_MB;
                    else { beta = root; err |= rc == 2 ? 1 : (rc == 3 ? 2 : 0); after_beta(); }
                }
                break;
            case VS_MB:
                br.fcur = f;
                rc = vn_brent_next(br, &root);
                if (rc != 0) { beta = root; err |= rc == 2 ? 1 : (rc == 3 ? 2 : 0); after_beta(); }
                break;
            case VS_A0: f_lo = f; stage = VS_A1; break;
            case VS_A1:
                st |= VN_ADD;
                rc = vn_brent_start(br, 0.0, 5.0, f_lo, f, &root);
                if (rc == 0) { stage = VS_AB; break; }
                [[fallthrough]];
            case VS_AB:
                if (stage == VS_AB) { br.fcur = f; rc = vn_brent_next(br, &root); if (rc == 0) break; }
                a = root; beta = 1.0; stage = VS_DONE;                               // spcpl.py:713-722
                if (rc == 1) apply = 2; else err |= rc == 2 ? 1 : 2;
                break;
            default: break;
            }
        }
    }

    // ---- apply: qt (and thl with constantT) of every touched level, all waves -----------------------------------------
    if (own) {
        s_apply[lane] = apply; s_touched[lane] = touched ? 1 : 0;
        s_coef[lane] = apply == 1 ? beta - 1 : a;
        double tc = 0.0;
        if (touched && p.constantT)
            tc = (-K<double>::rlv) / (K<double>::cp * spc_pow(div_pref0(p.presf[lev]), K<double>::rd / K<double>::cp));   // spcpl.py:731
        s_tc[lane] = tc;
    }
    __syncthreads();
    {
        const int ap = s_apply[lane], tch = s_touched[lane];
        const double coef = s_coef[lane], tc = s_tc[lane];
        const int seg = (nij + VN_NSL - 1) / VN_NSL, lo = sl * seg, hi = (lo + seg) < nij ? (lo + seg) : nij;
        if (valid && tch) {
            double *const thl = p.constantT ? p.thl + base : nullptr;
            const double *const ql = p.constantT ? p.ql + base : nullptr;
#pragma unroll 4
            for (int ij = lo; ij < hi; ++ij) {
                double q = qt[(int64_t)ij * ks];
                if (ap == 1) { q = q + coef * (q - qt_av); qt[(int64_t)ij * ks] = q; }        // spcpl.py:724-725
                else if (ap == 2) { q = q + coef * R[ij]; qt[(int64_t)ij * ks] = q; }         // spcpl.py:716-719
                if (thl) {                                                                   // spcpl.py:726-733
                    const double t = q - qsat[(int64_t)ij * ks];
                    const double ql_target = (t >= 0.0 || t != t) ? t : 0.0;
                    thl[(int64_t)ij * ks] += tc * (ql_target - ql[(int64_t)ij * ks]);
                }
            }
        }
    }
    __syncthreads();      // workgroup scope: the updated plane is visible to slice 0 (one CU, one L1)

    // ---- qt.std(axis=(0, 1)) (spcpl.py:741): numpy reduces over (i, j) with k as the inner loop, i.e. plain SEQUENTIAL
    //      sums in C order (mean = sum/N, then sum((x - mean)^2)/N, sqrt): slice 0, loads issued 16 at a time
    if (own && valid) {
        double s = 0.0;
        int ij = 0;
        for (; ij + 16 <= nij; ij += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = qt[(int64_t)(ij + u) * ks];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; ij < nij; ++ij) s += qt[(int64_t)ij * ks];
        const double mean = s / (double)nij;
        double var = 0.0;
        for (ij = 0; ij + 16 <= nij; ij += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = qt[(int64_t)(ij + u) * ks];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const double dlt = v[u] - mean; var += dlt * dlt; }
        }
        for (; ij < nij; ++ij) { const double dlt = qt[(int64_t)ij * ks] - mean; var += dlt * dlt; }
        p.qt_std[lev] = sqrt(var / (double)nij);
        p.beta[lev] = beta;
        p.a_add[lev] = a;
        p.status[lev] = st | ((err & 1) ? VN_ERR_SIGN : 0) | ((err & 2) ? VN_ERR_CONV : 0);
    }
}
