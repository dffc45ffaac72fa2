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
template <typename F> __device__ __forceinline__ auto vn_leaf(const F &term, int lo, int n) -> decltype(term(0))
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
#pragma unroll 4      // 4 x 8 terms: their loads are issued together (the sweep is load-latency bound: few waves)
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

__global__ __launch_bounds__(64) void k_vnudge(const VnP p)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    const int64_t col = blockIdx.y;
    if (k >= p.ktot || col >= p.n_cols) return;
    const int64_t lev = col * p.ktot + k, base = col * (int64_t)p.nij * p.ktot + k;
    double *const qt = p.qt + base;
    VnPlane pl;
    pl.qt = qt; pl.qsat = p.qsat + base; pl.R = p.R + col * (int64_t)p.nij; pl.stride = p.ktot;
    pl.qt_av = p.qt_av[lev]; pl.nij = p.nij;
    const double ql_ref = p.ql_ref[lev], ql_av = p.ql_av[lev];
    const double beta_min = 0.0, beta_max = 5.0;
    double beta = 1.0, a = 0.0;
    int st = VN_NONE, err = 0;
    bool touched = true;
    if (ql_ref > 1e-9) {                                                           // spcpl.py:665
        const double q_min = vn_ql_diff<false>(pl, beta_min, ql_ref), q_max = vn_ql_diff<false>(pl, beta_max, ql_ref);
        if (q_min > 0 || q_max < 0) { beta = beta_max; st = VN_NO_BRACKET; }       // spcpl.py:669-673
        else { beta = vn_brentq<false>(pl, ql_ref, beta_min, beta_max, q_min, q_max, &err); st = VN_MULT; }
    } else if (ql_av > ql_ref) {                                                   // spcpl.py:679-695
        // numpy.argmax(qt - qsat) over the plane in C order: first maximum, a NaN wins
        int best = 0;
        double bv = pl.qt[0] - pl.qsat[0];
#pragma unroll 8
        for (int ij = 1; ij < p.nij && !(bv != bv); ++ij) {
            const double v = pl.qt[(int64_t)ij * pl.stride] - pl.qsat[(int64_t)ij * pl.stride];
            if (v > bv || v != v) { bv = v; best = ij; }
        }
        beta = (pl.qsat[(int64_t)best * pl.stride] - pl.qt_av) / (pl.qt[(int64_t)best * pl.stride] - pl.qt_av);
        if (beta < 0) beta = 1.0;
        st = VN_UNSAT;
    } else {
        touched = false;                                                           // `continue`, spcpl.py:697
    }
    if (touched) {
        if (beta >= beta_max) {                                                    // spcpl.py:703-722
            if (ql_ref > ql_av) {
                const double g0 = vn_ql_diff<true>(pl, 0.0, ql_ref), g5 = vn_ql_diff<true>(pl, 5.0, ql_ref);
                int e2 = 0;
                a = vn_brentq<true>(pl, ql_ref, 0.0, 5.0, g0, g5, &e2);
                err |= e2;
                st |= VN_ADD;
                if (!e2) {
#pragma unroll 16
                    for (int ij = 0; ij < p.nij; ++ij) qt[(int64_t)ij * p.ktot] += a * pl.R[ij];       // qt[:,:,k] += a*R
                }
            } else {
                st |= VN_ADD_SKIPPED;
            }
            beta = 1.0;
        } else {
            const double bm1 = beta - 1;                                           // spcpl.py:724-725
#pragma unroll 16
            for (int ij = 0; ij < p.nij; ++ij) {
                const double q = qt[(int64_t)ij * p.ktot];
                qt[(int64_t)ij * p.ktot] = q + bm1 * (q - pl.qt_av);
            }
        }
        if (p.constantT) {                                                         // spcpl.py:726-734
            const double c = (-K<double>::rlv) / (K<double>::cp * spc_pow(div_pref0(p.presf[lev]), K<double>::rd / K<double>::cp));
            double *const thl = p.thl + base;
            const double *const ql = p.ql + base;
#pragma unroll 8
            for (int ij = 0; ij < p.nij; ++ij) {
                const double t = qt[(int64_t)ij * p.ktot] - pl.qsat[(int64_t)ij * pl.stride];
                const double ql_target = (t >= 0.0 || t != t) ? t : 0.0;
                thl[(int64_t)ij * p.ktot] += c * (ql_target - ql[(int64_t)ij * p.ktot]);
            }
        }
    }
    // qt.std(axis=(0, 1)) (spcpl.py:741): numpy reduces over (i, j) with k as the inner loop, i.e. plain sequential
    // sums in C order: mean = sum/N, then sum((x - mean)^2)/N, sqrt
    double s = 0.0;
#pragma unroll 32
    for (int ij = 0; ij < p.nij; ++ij) s += qt[(int64_t)ij * p.ktot];
    const double mean = s / (double)p.nij;
    double v = 0.0;
#pragma unroll 32
    for (int ij = 0; ij < p.nij; ++ij) {
        const double dlt = qt[(int64_t)ij * p.ktot] - mean;
        v += dlt * dlt;
    }
    p.qt_std[lev] = sqrt(v / (double)p.nij);
    p.beta[lev] = beta;
    p.a_add[lev] = a;
    p.status[lev] = st | (err == 1 ? VN_ERR_SIGN : 0) | (err == 2 ? VN_ERR_CONV : 0);
}
