// spc_k4.hpp -- K4, second form: conservative coarsening (splib/spcpl.py:479-489 -> sputils.interp_c / integral,
// splib/sputils.py:94-189) with ONE THREAD PER (GCM level, field).  The first form gave every GCM level to one
// thread, which then ran eight numpy-ordered sums (the weight sum and seven fields) one after the other: 233
// VGPRs, two waves per SIMD, long serial chains.  Here eight consecutive lanes share a level -- lanes 0..6 sum one
// field each, lane 7 the weights -- so the eight sums of a level run side by side, the layer means go through
// LDS, and a second, flat pass forms the seven tendencies with coalesced loads and stores exactly like K3.
// Every sum is numpy's ndarray.sum() of the reference's temporary (vn_npsum: pairwise blocks of 128, 8
// accumulators, 8192-element chunks), so any nL is supported and results stay bit-identical to the NumPy
// evaluation.  LDS per column: q[7][nL+1] = t | qt | ql | ql_ice | u | v | rho (one element of padding per
// field: a field stride of nL x 8 B = 0 mod 64 banks would put the eight lanes of a level on one bank), then
// Zh[nG+1] | X[7][nG] (layer means; its first nG elements hold Zf until start_index has been taken from them) | the
// cell range (ia, ib) of every level as two int16 in one word | start_index; then zh ([nL] when shared, else [CB x nL])
// and the LES top zf[nL-1] per column.
// K4's workgroups live as long as K3's (their loads queue in a saturated memory system), so its rate follows the
// number of RESIDENT COLUMNS per CU, which LDS caps: round 3 first went from one to two columns per workgroup (5 -> 8
// columns per CU: 250 -> 223 us at config 3), then cut the footprint from 19 to 15.2 KB per column -- the edge pieces
// da, db are recomputed by the lanes AFTER their sums (before them they cost the 6 VGPRs that separate five waves per
// SIMD from four), Zf shares the layer means' space, zf is not staged, (ia, ib) are packed -- so that a two-column
// workgroup needs 31 744 B = 25 of gfx950's 1 280-byte LDS allocation granules and FIVE of them fit a CU (32 464 B,
// one granule more, still gave four): 223 -> 194-196 us, 1.48-1.51 x K3.
// Other forms built and measured slower in round 3 (profiles/r03_k4_forms.log): a compacted list of the (column, level)
// pairs that have cells (255 us); one thread per (tendency, level) with the level running fastest, every thread
// repeating the searches and the weight sum, no layer means in LDS (366 us); the eight lanes of a level storing their
// own tendency -- 64-byte segments per array (259-282 us); the fifth wave bought with spilled registers (276 us).
#pragma once

// numpy's pairwise recursion with its depth fixed at compile time (no stack, no scratch memory): for the compile-time
// level geometries a layer covers at most NL cells, and vn_pw_depth(NL) levels of splitting reach leaves of <= 128
// elements for every n <= NL.  The run-time-geometry kernel keeps vn_npsum (explicit stack: 143 VGPRs, 3 waves per SIMD);
// with this form K4 needs 76 VGPRs (6 waves per SIMD).
constexpr int vn_pw_depth_of(int n) { return n <= 128 ? 0 : 1 + (vn_pw_depth_of(n / 2 - (n / 2) % 8) > vn_pw_depth_of(n - (n / 2 - (n / 2) % 8)) ? vn_pw_depth_of(n / 2 - (n / 2) % 8) : vn_pw_depth_of(n - (n / 2 - (n / 2) % 8))); }
constexpr int vn_pw_depth(int nmax) { int d = 0; for (int n = 129; n <= nmax; ++n) { const int dn = vn_pw_depth_of(n); if (dn > d) d = dn; } return d; }
template <int D, typename F> __device__ __forceinline__ auto vn_pw(const F &term, int lo, int n) -> decltype(term(0))
{
    if constexpr (D == 0) {
        return vn_leaf(term, lo, n);
    } else {
        if (n <= 128) return vn_leaf(term, lo, n);
        int n2 = n / 2;
        n2 -= n2 % 8;
        return vn_pw<D - 1>(term, lo, n2) + vn_pw<D - 1>(term, lo + n2, n - n2);
    }
}

// numpy's plain loop for 1 <= n < 8 terms (pairwise_sum's first branch: res = 0; res += a[i]) with ALL the terms fetched before
// the first addition: the loop as written waits for an LDS round trip per term (a dependent chain of n reads), this one for one.
// Terms beyond n re-read term 0 (readable: n >= 1) and are not added.
template <typename F> __device__ __forceinline__ auto vn_short(const F &term, int n) -> decltype(term(0))
{
    using T = decltype(term(0));
    T t[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) t[i] = term(i < n ? i : 0);
    T res = T(0);
#pragma unroll
    for (int i = 0; i < 7; ++i) res = i < n ? res + t[i] : res;
    return res;
}

// NG / NL != 0: level counts fixed at compile time and contiguous columns (as k_forward / k_backward): the flat-index
// divisions become multiply-shifts.
// PD (run-time geometry only): numpy's recursion unrolled to PD levels, enough for every layer of <= nL cells when
// PD >= vn_pw_depth(nL) -- the host picks the instantiation (1, 2, 3: nL up to 248 / 488 / 968 cells); PD = -1 keeps
// the explicit stack of vn_npsum for taller LES grids (152 VGPRs, scratch memory, 3 waves per SIMD).
template <typename T, int NG = 0, int NL = 0, int PD = -1> __global__ __launch_bounds__(BLOCK) void k_backward_cons2(const BwdP<T> p)
{
    const DimsP &d = p.d;
    const int nG = NG ? NG : d.nG, nL = NL ? NL : d.nL, cb = d.cb, tid = threadIdx.x, nLp = nL + 1;
    const int64_t pitchG = NG ? NG : d.pitchG, pitchGh = NG ? NG + 1 : d.pitchGh, pitchL = NL ? NL : d.pitchL;
    const int64_t col0 = (int64_t)slab_index(d.xcd_remap) * cb;
    const int ncol = (int)((d.n_cols - col0) < cb ? (d.n_cols - col0) : cb);
    constexpr int IPE = sizeof(T) / sizeof(int);                 // ints per element
    const size_t o_Zh = (size_t)7 * nLp, o_X = o_Zh + nG + 1, o_cell = o_X + (size_t)7 * nG;
    const size_t o_sidx = o_cell + (size_t)(nG + IPE - 1) / IPE, per_col = o_sidx + 1;
    T *const lds = reinterpret_cast<T *>(spc_smem);
    T *const lzh = lds + (size_t)cb * per_col;                       // zh
    T *const ltop = lzh + (d.shared_grid ? nL : (size_t)cb * nL);    // zf[nL-1] per column
    const int n1 = ncol * nG;
    STAMP(0);

    // ---- stage the LES slab, the LES half levels and the GCM heights --------------------------------------------
    for (int e = tid; e < ncol * nL; e += BLOCK) {
        const int c = e / nL, l = e - c * nL;
        const int64_t o = (col0 + c) * pitchL + l;
        T *const s = lds + (size_t)c * per_col + l;
        s[0] = p.t_d[o];
        s[nLp] = p.qt_d[o];
        s[2 * nLp] = p.ql_d[o];
        s[3 * nLp] = p.ql_ice_d[o];
        s[4 * nLp] = p.u_d[o];
        s[5 * nLp] = p.v_d[o];
        s[6 * nLp] = p.rhobf_d[o];
        if (!d.shared_grid) lzh[e] = p.zh[o];
    }
    if (d.shared_grid)
        for (int e = tid; e < nL; e += BLOCK) lzh[e] = p.zh[e];
    for (int e = tid; e < ncol * (nG + 1); e += BLOCK) {
        const int c = e / (nG + 1), k = e - c * (nG + 1);
        const int64_t col = col0 + c, gh = col * pitchGh;
        T *const s = lds + (size_t)c * per_col;
        s[o_Zh + k] = p.Zh ? p.Zh[gh + k] : div_grav(p.Zghalf[gh + k] - p.Zghalf[gh + nG]);    // spcpl.py:197
        if (k < nG) {
            const int64_t g = col * pitchG + k;
            s[o_X + k] = p.Zf ? p.Zf[g] : div_grav(p.Zgfull[g] - p.Zghalf[gh + nG]);           // Zf, spcpl.py:198 (until phase 3)
        } else {
            ltop[c] = p.zf[d.shared_grid ? (int64_t)(nL - 1) : col * pitchL + (nL - 1)];        // h[-1] of spcpl.py:498
        }
    }
    __syncthreads();
    STAMP(1);

    // ---- per GCM level, once: which LES cells the layer [Zh[k+1], Zh[k]] covers (the scans of integral(), sputils.py:
    //      113-127).  ia < 0 encodes the two special outcomes: -1 layer above the LES top (Q stays 0, sputils.py:187),
    //      -2 an end point outside zh (integral returns None -> NaN); ib < 0 encodes sign = -1 (sputils.py:117-120).
    //      One more thread per column: start_index, from Zf where the layer means will be.
    for (int e = tid; e < n1 + ncol; e += BLOCK) {
        if (e >= n1) {
            T *const s = lds + (size_t)(e - n1) * per_col;
            reinterpret_cast<int *>(s + o_sidx)[0] = ss_left_neg(s + o_X, nG, ltop[e - n1]);     // spcpl.py:498
            continue;
        }
        const int c = e / nG, k = e - c * nG;
        T *const s = lds + (size_t)c * per_col;
        const T *const z = d.shared_grid ? lzh : lzh + (size_t)c * nL;
        const T *const Zh = s + o_Zh;
        int *const cell = reinterpret_cast<int *>(s + o_cell);                         // (ia, ib) as two int16 in one word per level
        int ia = -1, ib = -1;
        if (Zh[k] < z[nL - 1]) {                                                       // sputils.py:187
            T a = Zh[k + 1], b = Zh[k];                                                // integral(ZZ[i+1], ZZ[i], ...)
            if (a < z[0] || a > z[nL - 1] || b < z[0] || b > z[nL - 1]) {
                ia = -2;                                                               // sputils.py:113-115
            } else {
                const bool swap = a > b;                                               // sputils.py:117-120
                if (swap) { const T t = a; a = b; b = t; }
                ia = scan_cell(z, nL, a);                                              // sputils.py:122-124
                ib = scan_cell(z, nL, b);                                              // sputils.py:125-127
                if (ib < ia) ib = ia;
                if (swap) ib = -ib - 2;
            }
        }
        cell[k] = (ia & 0xffff) | (ib * 65536);       // |ia|, |ib| < 2^15: an LES column of that height would not fit the LDS
    }
    __syncthreads();
    STAMP(2);

    // ---- layer means: thread = (column, GCM level, field); lane&7 = field, 7 = the weight sum -------------------
    for (int e = tid; e < n1 * 8; e += BLOCK) {
        const int f = e & 7, ck = e >> 3, c = ck / nG, k = ck - c * nG;
        T *const s = lds + (size_t)c * per_col;
        const T *const z = d.shared_grid ? lzh : lzh + (size_t)c * nL;
        const T *const w = s + 6 * nLp;
        const int *const cell = reinterpret_cast<const int *>(s + o_cell);
        const int pk = cell[k], ia = (int)(short)(pk & 0xffff);
        T X = T(0);                                                                    // Q = zeros (sputils.py:185)
        if (ia == -2) {
            X = T(0) / T(0);           // Q[i] = None stores NaN (numpy 2.x)
        } else if (ia >= 0) {
            int ib = pk >> 16;
            const bool swap = ib < 0;
            if (swap) ib = -ib - 2;
            const T sign = swap ? T(-1) : T(1);
            const int cnt = ib - ia + 1;
            // fields in the order of spcpl.py:482-488: t, qt, ql, ql_water (= ql - ql_ice, :402), ql_ice, u, v
            const T *const qa = s + (size_t)(f < 3 ? f : (f < 7 ? f - 1 : 0)) * nLp;
            const T *const qb = s + (size_t)3 * nLp;
            const bool sub = (f == 3), wsum = (f == 7);
            auto q = [&](int i) { return sub ? qa[i] - qb[i] : qa[i]; };
            auto term = [&](int i) {
                const T dz = z[ia + i + 1] - z[ia + i];
                return wsum ? w[ia + i] * dz : (w[ia + i] * q(ia + i)) * dz;           // sputils.py:152 / 157
            };
            T S;
            if constexpr (NL != 0) S = T(0) + vn_pw<vn_pw_depth(NL)>(term, 0, cnt);         // cnt <= NL <= 8192: one chunk
            else if constexpr (PD >= 0) S = T(0) + vn_pw<PD>(term, 0, cnt);                 // cnt <= nL, depth checked by the host
            else S = cnt <= 128 ? T(0) + vn_leaf(term, 0, cnt) : vn_npsum(term, cnt);
            // the edge pieces, only now: they are not live during the sums (4 VGPRs: the fifth wave per SIMD)
            const T za = s[o_Zh + k + 1], zb = s[o_Zh + k];                            // a, b of integral() before the swap
            const T da = (swap ? zb : za) - z[ia], db = z[ib + 1] - (swap ? za : zb);  // sputils.py:154,159
            const T ea = wsum ? w[ia] * da : (w[ia] * q(ia)) * da;                     // Sa / Swa, sputils.py:154,159
            const T eb = wsum ? w[ib] * db : (w[ib] * q(ib)) * db;                     // Sb / Swb
            const T num = (S - ea) - eb;
            const T den = __shfl(num, (threadIdx.x & 63) | 7);                         // the level's weight lane
            X = num / den * sign;                                                      // sputils.py:161
        }
        if (f < 7) s[o_X + (size_t)f * nG + k] = X;
    }
    __syncthreads();
    STAMP(3);

    // ---- tendencies: flat over the [ncol x nG] slab, as K3 (spcpl.py:498, 518-533) -----------------------------
    for (int e = tid; e < n1; e += BLOCK) {
        const int c = e / nG, k = e - c * nG;
        const int64_t col = col0 + c, cg = col * pitchG, g = cg + k;
        const T *const s = lds + (size_t)c * per_col;
        const T *const X = s + o_X;
        const GcmIn<T> in = load_gcm(p, g, cg + (nG - 1 - k));
        const int start_index = reinterpret_cast<const int *>(s + o_sidx)[0];
        const T X0 = X[k], X1 = X[nG + k], X2 = X[2 * nG + k], X3 = X[3 * nG + k], X4 = X[4 * nG + k], X5 = X[5 * nG + k],
                X6 = X[6 * nG + k];
        T f_T = p.factor * (X0 - in.tt) / p.dt;                                        // spcpl.py:518
        T f_SH = p.factor * ((X1 - X2) - in.sh) / p.dt;                                // spcpl.py:519
        T f_QL = p.factor * (X3 - in.ql) / p.dt;                                       // spcpl.py:520
        T f_QI = p.factor * (X4 - in.qi) / p.dt;                                       // spcpl.py:521
        T f_U = p.factor * (X5 - in.u) / p.dt;                                         // spcpl.py:524
        T f_V = p.factor * (X6 - in.v) / p.dt;                                         // spcpl.py:525
        T f_A = p.factor * (in.a_d - in.a) / p.dt;                                     // spcpl.py:526
        if (k < start_index) {                                                         // spcpl.py:527-533
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
    STAMP(4);
    STAMP(5);
}

// =================================================================================================================
// K4, third form (round 4) for the compile-time level geometries: k_backward_cons3<T, NG, NL, CB>.
// The stamps of the second form at config 3 (profiles/r04_k4_stamps.log) put 7.5 of a workgroup's 13.2 us into its two
// compute phases -- 2.6 us for the cell scans, 4.8 us for the layer sums (K3's whole compute phase: 3.8 us) -- and the SQ
// counters of round 3 6 x K3's LDS instructions.  What the sums did per term: z[i+1], z[i], w[i], q[i] (+ ql_ice[i]) from
// LDS, a subtraction and two multiplications.  Here
//   * the products the reference's temporaries hold are formed ONCE per cell while staging -- wq_f[i] = w[i] q_f[i] for the
//     seven fields (ql_water = ql - ql_ice included, spcpl.py:402) next to w[i], and dz[i] = zh[i+1] - zh[i] once per
//     workgroup for a shared grid -- so a term is wq_f[i] dz[i]: two LDS reads and one multiplication, same bits
//     ((w q) dz, sputils.py:154: the same two roundings);
//   * the cell scans run on a NaN-padded copy of zh with a compile-time trip count (su_count: 3 VALU instructions per step);
//   * start_index is found while staging, by the thread that holds Zf[k] (left neighbour's value by a shuffle), so Zf
//     is never staged;
//   * the layer means do not get a region of their own: every lane keeps the means of its <= MAXIT (level, field) items in
//     registers across a barrier and then writes them over the product arrays, which nobody reads any more.
// LDS per column: A[8][nL+1] (wq_t | wq_qt | wq_ql | wq_ql_water | wq_ql_ice | wq_u | wq_v | w) | Zh[nG+1] | cell[nG] | start
// index = 11.4 KB at 91 <-> 160 (second form: 15.2): a two-column workgroup takes 26.2 KB, six of them fit a CU.
// CB: columns per workgroup (1 or 2), a template argument because the register stash needs its trip count.
// =================================================================================================================
constexpr int k4_sl(int nL) { return cfloor_pow2(nL - 1) == 64 ? 7 : cfloor_pow2(nL - 1) == 128 ? 8 : cfloor_pow2(nL - 1) == 256 ? 9
                                   : cfloor_pow2(nL - 1) == 512 ? 10 : 0; }

// (the unroll request of the register-stash loop is not honoured for the 137 <-> 512 instantiations -- their loop body, two
//  levels of numpy's recursion, is too large -- where the stash is then addressed through the loop counter: no scratch either)
// (column, entry) of item e of a [CB x n] slab with CB <= 2: a compare instead of the magic-number division (v_mul_hi_u32: quarter rate)
template <int CB> __device__ __forceinline__ void k4_split(int e, int n, int &c, int &j)
{
    if constexpr (CB == 1) { c = 0; j = e; }
    else { c = e >= n ? 1 : 0; j = e - (c ? n : 0); }
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
#ifndef K4_SKIP          // diagnostic builds only (-DK4_SKIP=1): no cell scans, no layer sums -- what the kernel's memory side alone costs
#define K4_SKIP 0
#endif
#ifndef SPC_K4_WAVES     // waves per SIMD the register allocator is asked to fit (6 = 80 VGPRs: six workgroups per CU)
#define SPC_K4_WAVES 6
#endif
template <typename T, int NG, int NL, int CB> __global__ __launch_bounds__(BLOCK, (NL <= 160 ? SPC_K4_WAVES : 1)) void k_backward_cons3(const BwdP<T> p)
{
    static_assert(NG > 0 && NL > 1 && (CB == 1 || CB == 2), "compile-time geometry, one or two columns per workgroup");
    const DimsP &d = p.d;
    constexpr int nG = NG, nL = NL, nLp = NL + 1, tid_n = BLOCK;
    constexpr int64_t pitchG = NG, pitchGh = NG + 1, pitchL = NL;
    constexpr int P2 = cfloor_pow2(NL - 1), SL = k4_sl(NL), ZROW = 2 * P2 + 2;          // su_pad(P2): the padded zh row
    constexpr int IPE = sizeof(T) / sizeof(int);
    constexpr size_t o_Zh = (size_t)8 * nLp, o_cell = o_Zh + nG + 1, o_sidx = o_cell + (size_t)(nG + IPE - 1) / IPE, per_col = o_sidx + 1;
    constexpr int MAXIT = (CB * NG * 8 + BLOCK - 1) / BLOCK;                              // (level, field) items per thread
    static_assert(7 * NG <= 8 * (NL + 1), "the layer means must fit over the product arrays");
    const int tid = threadIdx.x;
    const int64_t col0 = (int64_t)slab_index(d.xcd_remap) * CB;
    const int ncol = (int)((d.n_cols - col0) < CB ? (d.n_cols - col0) : CB);
    T *const lds = reinterpret_cast<T *>(spc_smem);
    T *const lzh = lds + (size_t)CB * per_col;                                           // zh rows, NaN-padded: [1 or CB][ZROW]
    T *const ldz = lzh + (d.shared_grid ? ZROW : (size_t)CB * ZROW);                     // dz [nL - 1] (shared grid only)
    const int n1 = ncol * nG;
    STAMP(0);

    // ---- stage: products per cell, the grid, the GCM half levels; start_index from Zf on the way -------------------------
    for (int e = tid; e < ncol * nL; e += tid_n) {
        int c, l;
        k4_split<CB>(e, nL, c, l);
        const int64_t o = (col0 + c) * pitchL + l;
        const T t = p.t_d[o], qt = p.qt_d[o], ql = p.ql_d[o], qi = p.ql_ice_d[o], u = p.u_d[o], v = p.v_d[o], w = p.rhobf_d[o];
        T *const s = lds + (size_t)c * per_col + l;
        s[0] = SPC_MUT(11, t, w * t);                                                    // sputils.py:152: w * q, per field
        s[nLp] = w * qt;                                                                 // (order of spcpl.py:482-488)
        s[2 * nLp] = w * ql;
        s[3 * nLp] = w * (ql - qi);                                                      // ql_water, spcpl.py:402
        s[4 * nLp] = w * qi;
        s[5 * nLp] = w * u;
        s[6 * nLp] = w * v;
        s[7 * nLp] = w;
    }
    {
        const T nan = T(0) / T(0);
        const int nz = d.shared_grid ? ZROW : ncol * ZROW;
        for (int e = tid; e < nz; e += tid_n) {
            int c, j;
            k4_split<CB>(e, ZROW, c, j);
            lzh[e] = j < nL ? (d.shared_grid ? p.zh[j] : p.zh[(col0 + c) * pitchL + j]) : nan;
        }
        if (d.shared_grid)
            for (int e = tid; e < nL - 1; e += tid_n) ldz[e] = p.zh[e + 1] - p.zh[e];    // sputils.py:146 / 154 / 159: z[i+1] - z[i]
    }
    for (int e = tid; e < ncol * (nG + 1); e += tid_n) {
        int c, k;
        k4_split<CB>(e, nG + 1, c, k);
        const int64_t col = col0 + c, gh = col * pitchGh;
        T *const s = lds + (size_t)c * per_col;
        const T zs = p.Zghalf ? p.Zghalf[gh + nG] : T(0);
        s[o_Zh + k] = p.Zh ? p.Zh[gh + k] : div_grav(p.Zghalf[gh + k] - zs);             // spcpl.py:197
        // start_index = numpy.searchsorted(-Zf, -h[-1]) (spcpl.py:498): the first k with !(-Zf[k] < -h_top).  The thread that
        // holds Zf[k] decides with its left neighbour's value (shuffle; a wave's first lane loads it): one writer per
        // column for heights the predicate partitions (the reference's: monotone); k == nG closes the all-above case.
        const int kk = k < nG ? k : nG - 1;
        const int64_t g = col * pitchG + kk;
        const T zfk = p.Zf ? p.Zf[g] : div_grav(p.Zgfull[g] - zs);                       // spcpl.py:198
        T prev = __shfl_up(zfk, 1);
        if ((tid & 63) == 0 && k > 0 && k < nG) prev = p.Zf ? p.Zf[g - 1] : div_grav(p.Zgfull[g - 1] - zs);
        const T key = -p.zf[d.shared_grid ? (int64_t)(nL - 1) : col * pitchL + (nL - 1)];        // -h[-1]
        const bool less_k = np_lt(-zfk, key), less_prev = k > 0 && np_lt(-prev, key);
        int *const sidx = reinterpret_cast<int *>(s + o_sidx);
        if (k < nG && !less_k && (k == 0 || less_prev)) sidx[0] = k;
        if (k == nG && less_k) sidx[0] = nG;                                             // (zfk is Zf[nG-1] here)
    }
    __syncthreads();
    STAMP(1);

    // ---- per GCM level, once: the LES cells of the layer [Zh[k+1], Zh[k]] (integral()'s scans, sputils.py:113-127).
    //      ia = -1: layer above the LES top (Q stays 0, sputils.py:187); -2: an end point outside zh (None -> NaN);
    //      ib < 0 encodes sign = -1 (sputils.py:117-120).
    for (int e = tid; e < n1; e += tid_n) {
        int c, k;
        k4_split<CB>(e, nG, c, k);
        T *const s = lds + (size_t)c * per_col;
        const T *const z = d.shared_grid ? lzh : lzh + (size_t)c * ZROW;
        const T *const Zh = s + o_Zh;
        int ia = -1, ib = -1;
        if (K4_SKIP == 0 && Zh[k] < z[nL - 1]) {                                                         // sputils.py:187
            T a = Zh[k + 1], b = Zh[k];
            if (a < z[0] || a > z[nL - 1] || b < z[0] || b > z[nL - 1]) {
                ia = -2;                                                                 // sputils.py:113-115
            } else {
                const bool swap = a > b;                                                 // sputils.py:117-120
                if (swap) { const T t = a; a = b; b = t; }
                // `while z[i+1] < a: i += 1` = the number of k' >= 1 with z[k'] < a (z[nL-1] < a is false: a <= z[nL-1])
                ia = su_count<SL>(z + 1, P2, [&](T zk) { return zk < a; });              // sputils.py:122-124
                ib = su_count<SL>(z + 1, P2, [&](T zk) { return zk < b; });              // sputils.py:125-127
                if (ib < ia) ib = ia;
                if (swap) ib = -ib - 2;
            }
        }
        reinterpret_cast<int *>(s + o_cell)[k] = (ia & 0xffff) | (ib * 65536);
    }
    __syncthreads();
    STAMP(2);

    // ---- layer means: thread = (column, GCM level, field); lane & 7 = field, 7 = the weight sum; kept in registers ------
    T Xr[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int e = tid + it * BLOCK;
        T X = T(0);                                                                      // Q = zeros (sputils.py:185)
        if (e < n1 * 8) {
            const int f = e & 7, ck = e >> 3;
            int c, k;
            k4_split<CB>(ck, nG, c, k);
            const T *const s = lds + (size_t)c * per_col;
            const T *const z = d.shared_grid ? lzh : lzh + (size_t)c * ZROW;
            const int pk = reinterpret_cast<const int *>(s + o_cell)[k], ia = (int)(short)(pk & 0xffff);
            if (ia == -2) {
                X = T(0) / T(0);                                                         // Q[i] = None stores NaN (numpy 2.x)
            } else if (ia >= 0) {
                int ib = pk >> 16;
                const bool swap = ib < 0;
                if (swap) ib = -ib - 2;
                const T sign = swap ? T(-1) : T(1);
                const int cnt = ib - ia + 1;
                const T *const a = s + (size_t)f * nLp + ia;                             // this lane's products (lane 7: the weights)
                T S;
                if (d.shared_grid) {
                    const T *const dz = ldz + ia;
                    const auto term = [&](int i) { return a[i] * dz[i]; };                   // sputils.py:152 / 157
                    S = cnt < 8 ? T(0) + vn_short(term, cnt) : T(0) + vn_pw<vn_pw_depth(NL)>(term, 0, cnt);
                } else {
                    const T *const zz = z + ia;
                    const auto term = [&](int i) { return a[i] * (zz[i + 1] - zz[i]); };
                    S = cnt < 8 ? T(0) + vn_short(term, cnt) : T(0) + vn_pw<vn_pw_depth(NL)>(term, 0, cnt);
                }
                const T za = s[o_Zh + k + 1], zb = s[o_Zh + k];                          // a, b of integral() before the swap
                const T da = (swap ? zb : za) - z[ia], db = z[ib + 1] - (swap ? za : zb);   // sputils.py:154,159
                const T num = (S - a[0] * da) - a[cnt - 1] * db;                         // S - Sa - Sb / Sw - Swa - Swb
                const T den = __shfl(num, (threadIdx.x & 63) | 7);                       // the level's weight lane
                X = num / den * sign;                                                    // sputils.py:161
            }
        }
        Xr[it] = X;
    }
    __syncthreads();                                                                     // nobody reads the products any more
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int e = tid + it * BLOCK;
        if (e < n1 * 8) {
            const int f = e & 7, ck = e >> 3;
            int c, k;
            k4_split<CB>(ck, nG, c, k);
            if (f < 7) lds[(size_t)c * per_col + (size_t)f * nG + k] = Xr[it];           // X[7][nG] over the product arrays
        }
    }
    __syncthreads();
    STAMP(3);

    // ---- tendencies: flat over the [ncol x nG] slab, as K3 (spcpl.py:498, 518-533) --------------------------------------
    for (int e = tid; e < n1; e += tid_n) {
        int c, k;
        k4_split<CB>(e, nG, c, k);
        const int64_t col = col0 + c, cg = col * pitchG, g = cg + k;
        const T *const s = lds + (size_t)c * per_col;
        const GcmIn<T> in = load_gcm(p, g, cg + (nG - 1 - k));
        const int start_index = reinterpret_cast<const int *>(s + o_sidx)[0];
        const T X0 = s[k], X1 = s[nG + k], X2 = s[2 * nG + k], X3 = s[3 * nG + k], X4 = s[4 * nG + k], X5 = s[5 * nG + k], X6 = s[6 * nG + k];
        T f_T = p.factor * (X0 - in.tt) / p.dt;                                          // spcpl.py:518
        T f_SH = p.factor * ((X1 - X2) - in.sh) / p.dt;                                  // spcpl.py:519
        T f_QL = p.factor * (X3 - in.ql) / p.dt;                                         // spcpl.py:520
        T f_QI = p.factor * (X4 - in.qi) / p.dt;                                         // spcpl.py:521
        T f_U = p.factor * (X5 - in.u) / p.dt;                                           // spcpl.py:524
        T f_V = p.factor * (X6 - in.v) / p.dt;                                           // spcpl.py:525
        T f_A = p.factor * (in.a_d - in.a) / p.dt;                                       // spcpl.py:526
        if (k < start_index) {                                                           // spcpl.py:527-533
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
    STAMP(4);
    STAMP(5);
}
#pragma clang diagnostic pop
