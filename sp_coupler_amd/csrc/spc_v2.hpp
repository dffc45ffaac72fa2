// spc_v2.hpp -- second-generation K1 / K3 for the hot geometries (fp64, contiguous columns, level counts and slab
// size fixed at compile time).  Included by spc_hip.hip after the shared device helpers; same arithmetic,
// operation for operation, as k_forward / k_backward (interp_fields / lerp_np / ss_* are reused), different
// data movement:
//   * every HBM access of the big arrays is 16 B per lane (double2): a slab of CB columns is contiguous in
//     every [n_cols x n_lev] array and starts 16-B aligned because CB is even, so the flat slab is read and
//     written as double2 regardless of the odd row length (91 / 137 doubles);
//   * ALL loads of a workgroup -- the GCM slab AND the LES-side inputs of its output levels -- are issued in the
//     first instructions of the kernel and stay in flight together (~13 KB per column): the kernel is a
//     load burst -> convert -> barrier -> interpolate -> store burst, and the memory system sees deep queues from
//     every resident workgroup instead of one dependent load per loop iteration;
//   * the staged profiles are kept in LDS as ONE RECORD PER LEVEL (array of structures): the two bracketing
//     samples of all fields and their abscissae are 96 contiguous bytes -> 6 ds_read_b128 instead of 12-14
//     ds_read_b64 plus two abscissa reads, and neighbouring lanes (different levels) hit different banks
//     (record stride 48 B; the round-1 SoA field stride of 160 x 8 B / 512 x 8 B was 0 mod 64 banks);
//   * each thread produces TWO consecutive output levels (two independent search + division chains).
// Reference lines: splib/spcpl.py:171-246, 299-385 (K1), 388-555 (K3); splib/sputils.py:28-34, 82-91.
#pragma once

typedef double d2 __attribute__((ext_vector_type(2)));

// 16-B load of flat elements (e, e+1) of a slab holding `lim` valid elements; the tail element of an odd-sized
// last slab is fetched alone, nothing is read beyond `lim`.
// MODE (the kernels' WT parameter): 0 plain, 1 write-through stores (small launches), 2 non-temporal loads AND
// stores (streamed once: measured -2...-6.5 % on K1 between ~4k and ~100k columns, profiles/r02_nt_ab.log).
template <int MODE = 0> __device__ __forceinline__ d2 ld2(const double *q, int e, int lim)
{
    d2 r = {0.0, 0.0};
    if (e + 1 < lim) {
        if constexpr (MODE == 2 || SPC_NT == 1) r = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(q + e));
        else r = *reinterpret_cast<const d2 *>(q + e);
    } else if (e < lim) {
        r.x = ldg(q + e);
    }
    return r;
}

template <int WT> __device__ __forceinline__ void st2(double *q, int e, int lim, d2 v)
{
    if (e + 1 < lim) {
        if constexpr (WT == 2 || SPC_NT == 1) {
            __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(q + e));
        } else if constexpr (WT == 1) {
            // write-through (sc1) 16-B store: nothing stays dirty in L2 for the end-of-kernel release
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(q + e), "v"(v) : "memory");
        } else {
            *reinterpret_cast<d2 *>(q + e) = v;
        }
    } else if (e < lim) {
        stg<(WT == 1 ? 1 : 0)>(q + e, v.x);
    }
}

// numpy.interp bracket of x in the ascending abscissae xs[0..N) (N >= 2, compile time), then all NF fields from the
// two records rec[jc], rec[jc+1] (RS doubles each; fields first, abscissa at index XI): exactly bracket2 +
// interp_fields, with the samples coming from two record reads.
template <int N, int NF, int RS, int XI>
__device__ __forceinline__ void interp_records(const double *xs, const double *rec, double x, double (&r)[NF])
{
    const int j = upper_count(xs, N, cfloor_pow2(N), x) - 1;          // NaN x: every comparison false -> j = -1
    const bool below = j < 0, above = j >= N - 1;
    const int jc = j < 0 ? 0 : (j > N - 2 ? N - 2 : j);
    const d2 *const q = reinterpret_cast<const d2 *>(rec + (size_t)jc * RS);
    double lo[RS], hi[RS];
#pragma unroll
    for (int k = 0; k < RS / 2; ++k) {
        const d2 a = q[k], b = q[RS / 2 + k];
        lo[2 * k] = a.x; lo[2 * k + 1] = a.y;
        hi[2 * k] = b.x; hi[2 * k + 1] = b.y;
    }
    Br<double> b;
    const double x0 = lo[XI], x1 = hi[XI];
    b.take = below | above | (x0 == x);
    b.nanx = (x != x);
    b.x = x;
    b.x0 = b.take ? 0.0 : x0;
    b.x1 = b.take ? 1.0 : x1;
    double f0[NF], f1[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        f0[k] = above ? hi[k] : lo[k];          // take: fp[j0], j0 = above ? N-1 : jc
        f1[k] = b.take ? f0[k] : hi[k];
    }
    interp_fields<NF>(b, f0, f1, r);
}

// =================================================================================================
// K1 v2.  LDS: rec[CB][NG][6] = {thl_, qt_, QL, U, V, Zf} in ascending-height order | zfs[CB][NG] (search copy
// of Zf) | zh[NL] (shared LES half levels, only with the fused index map).
// =================================================================================================
template <int NG, int NL, int CB, int BLOCK, int WT>
__global__ __launch_bounds__(BLOCK) void k_forward_v2(const FwdP<double, false> p)
{
    static_assert(CB % 2 == 0 && NL % 2 == 0 && NG >= 2, "slab must be 16-B aligned in every array");
    constexpr int NI1 = CB * NG / 2, NI2 = CB * NL / 2, NIX = CB * NG;
    constexpr int IT1 = (NI1 + BLOCK - 1) / BLOCK, IT2 = (NI2 + BLOCK - 1) / BLOCK, ITX = (NIX + BLOCK - 1) / BLOCK;
    constexpr int RS = 6;
    const DimsP &d = p.d;
    const int tid = threadIdx.x;
    double *const lrec = reinterpret_cast<double *>(spc_smem);
    double *const lzf = lrec + (size_t)CB * NG * RS;
    double *const lzh = lzf + (size_t)CB * NG;
    const bool want_idx = p.idx != nullptr;
    const int sc = BLOCK - 1 - tid;                    // the LAST threads own the per-column scalars

    struct Slab {
        int64_t col0, g0, h0, o0;
        int ncol, lim1, lim2;
    };
    auto make_slab = [&](int64_t s) {
        Slab sb;
        sb.col0 = s * CB;
        sb.ncol = (int)((d.n_cols - sb.col0) < CB ? (d.n_cols - sb.col0) : CB);
        sb.lim1 = sb.ncol * NG; sb.lim2 = sb.ncol * NL;
        sb.g0 = sb.col0 * NG; sb.h0 = sb.col0 * (NG + 1); sb.o0 = sb.col0 * NL;
        return sb;
    };

    // register sets: G = what the staging pass consumes, L = what the interpolation pass consumes
    d2 gT[IT1], gSH[IT1], gQL[IT1], gQI[IT1], gPf[IT1], gZg[IT1], gU[IT1], gV[IT1];
    double zs0[IT1], zs1[IT1];
    d2 lh[IT2], lu[IT2], lv[IT2], lthl[IT2], lqt[IT2], lql[IT2];
    double xzgh[ITX], xzs[ITX], sc_ps = 0.0, sc_psd = 0.0;

    auto issue_gcm = [&](const Slab &sb) {
#pragma unroll
        for (int it = 0; it < IT1; ++it) {
            const int e = 2 * (tid + it * BLOCK);
            gZg[it] = ld2<WT>(p.Zgfull + sb.g0, e, sb.lim1);
            gPf[it] = ld2<WT>(p.Pf + sb.g0, e, sb.lim1);
            gT[it] = ld2<WT>(p.Tm + sb.g0, e, sb.lim1);
            gQL[it] = ld2<WT>(p.QL + sb.g0, e, sb.lim1);
            gQI[it] = ld2<WT>(p.QI + sb.g0, e, sb.lim1);
            gSH[it] = ld2<WT>(p.SH + sb.g0, e, sb.lim1);
            gU[it] = ld2<WT>(p.U + sb.g0, e, sb.lim1);
            gV[it] = ld2<WT>(p.V + sb.g0, e, sb.lim1);
            const int c0 = e / NG, c1 = (e + 1) / NG;
            zs0[it] = e < sb.lim1 ? p.Zghalf[sb.h0 + (int64_t)c0 * (NG + 1) + NG] : 0.0;     // spcpl.py:197-198
            zs1[it] = e + 1 < sb.lim1 ? p.Zghalf[sb.h0 + (int64_t)c1 * (NG + 1) + NG] : 0.0;
        }
    };
    auto issue_les = [&](const Slab &sb) {
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
            // odd iterations run over the threads in reverse, so a short last iteration lands on the waves that got
            // no GCM items
            const int e = 2 * (it * BLOCK + ((it & 1) ? BLOCK - 1 - tid : tid));
            const int l = e % NL;
            lh[it] = d.shared_grid ? ld2<0>(p.zf, l, NL) : ld2<WT>(p.zf + sb.o0, e, sb.lim2);  // spcpl.py:222
            lu[it] = ld2<WT>(p.u_d + sb.o0, e, sb.lim2);
            lv[it] = ld2<WT>(p.v_d + sb.o0, e, sb.lim2);
            lthl[it] = ld2<WT>(p.thl_d + sb.o0, e, sb.lim2);
            lqt[it] = ld2<WT>(p.qt_d + sb.o0, e, sb.lim2);
            lql[it] = ld2<WT>(p.ql_d + sb.o0, e, sb.lim2);
        }
        if (want_idx) {
#pragma unroll
            for (int it = 0; it < ITX; ++it) {
                const int e = tid + it * BLOCK, c = e / NG, m = e - c * NG;
                const int64_t gh = sb.h0 + (int64_t)c * (NG + 1);
                xzgh[it] = e < sb.lim1 ? p.Zghalf[gh + (NG - 1 - m)] : 0.0;
                xzs[it] = e < sb.lim1 ? p.Zghalf[gh + NG] : 0.0;
            }
        }
        if (sc < sb.ncol) {
            sc_ps = p.Ph[sb.h0 + (int64_t)sc * (NG + 1) + NG];                                 // spcpl.py:246
            sc_psd = p.ps_d[sb.col0 + sc];
        }
    };
    // convert the GCM levels and stage one record per level, reversed to ascending height
    auto stage = [&](const Slab &sb) {
        if (want_idx && !d.shared_grid)                                                        // per-column LES half levels
            for (int e = 2 * tid; e < sb.lim2; e += 2 * BLOCK) *reinterpret_cast<d2 *>(lzh + e) = ld2<WT>(p.zh + sb.o0, e, sb.lim2);
#pragma unroll
        for (int it = 0; it < IT1; ++it) {
            const int e0 = 2 * (tid + it * BLOCK);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int e = e0 + s;
                if (e < sb.lim1) {
                    const int c = e / NG, k = e - c * NG;
                    const double tt = s ? gT[it].y : gT[it].x, sh = s ? gSH[it].y : gSH[it].x, ql = s ? gQL[it].y : gQL[it].x;
                    const double qi = s ? gQI[it].y : gQI[it].x, pf = s ? gPf[it].y : gPf[it].x, zg = s ? gZg[it].y : gZg[it].x;
                    const double uu = s ? gU[it].y : gU[it].x, vv = s ? gV[it].y : gV[it].x;
                    const double zf_k = div_grav(zg - (s ? zs1[it] : zs0[it]));                // spcpl.py:198
                    const double qt_ = sh + ql + qi;                                           // spcpl.py:215
                    const double iex = spc_pow(div_pref0(pf), (-K<double>::rd) / K<double>::cp);   // sputils.py:34
                    const double thl_ = (tt - div_cp(K<double>::rlv * (ql + qi))) * iex;       // spcpl.py:214
                    const int kr = c * NG + (NG - 1 - k);                                      // [::-1], spcpl.py:224
                    d2 *const q = reinterpret_cast<d2 *>(lrec + (size_t)kr * RS);
                    q[0] = d2{thl_, qt_};
                    q[1] = d2{ql, uu};
                    q[2] = d2{vv, zf_k};
                    lzf[kr] = zf_k;
                }
            }
        }
    };
    // two LES levels per thread: 5 fields each, forcings, 16-B stores; per-column scalars; fused K2
    auto compute = [&](const Slab &sb) {
        if (sc < sb.ncol) stg<(WT == 1 ? 1 : 0)>(&p.f_ps[sb.col0 + sc], SPC_DIV(p.factor * (sc_ps - sc_psd), p.dt));   // spcpl.py:332
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
            const int e = 2 * (it * BLOCK + ((it & 1) ? BLOCK - 1 - tid : tid));
            if (e < sb.lim2) {
                const int c = e / NL;
                const double *const xs = lzf + (size_t)c * NG;
                const double *const rc = lrec + (size_t)c * NG * RS;
                double ra[5], rb[5];
                interp_records<NG, 5, RS, 5>(xs, rc, lh[it].x, ra);                            // spcpl.py:224-228
                interp_records<NG, 5, RS, 5>(xs, rc, lh[it].y, rb);
                // record order: thl, qt, ql, u, v
                const int64_t o0 = sb.o0;
                const int lim2 = sb.lim2;
                st2<WT>(p.f_u + o0, e, lim2, d2{SPC_DIV(p.factor * (ra[3] - lu[it].x), p.dt), SPC_DIV(p.factor * (rb[3] - lu[it].y), p.dt)});       // :328
                st2<WT>(p.f_v + o0, e, lim2, d2{SPC_DIV(p.factor * (ra[4] - lv[it].x), p.dt), SPC_DIV(p.factor * (rb[4] - lv[it].y), p.dt)});       // :329
                st2<WT>(p.f_thl + o0, e, lim2, d2{SPC_DIV(p.factor * (ra[0] - lthl[it].x), p.dt), SPC_DIV(p.factor * (rb[0] - lthl[it].y), p.dt)}); // :330
                st2<WT>(p.f_qt + o0, e, lim2, d2{SPC_DIV(p.factor * (ra[1] - lqt[it].x), p.dt), SPC_DIV(p.factor * (rb[1] - lqt[it].y), p.dt)});    // :331
                st2<WT>(p.f_ql + o0, e, lim2, d2{SPC_DIV(p.factor * (ra[2] - lql[it].x), p.dt), SPC_DIV(p.factor * (rb[2] - lql[it].y), p.dt)});    // :333
                st2<WT>(p.ql_ref + o0, e, lim2, d2{ra[2], rb[2]});                                                                   // :347-348
            }
        }
        if (want_idx) {                                                                        // spcpl.py:764
#pragma unroll
            for (int it = 0; it < ITX; ++it) {
                const int e = tid + it * BLOCK;
                if (e < sb.lim1) {
                    const int c = e / NG;
                    const double Zh_k = div_grav(xzgh[it] - xzs[it]);                          // spcpl.py:197
                    const double *const zh = d.shared_grid ? lzh : lzh + (size_t)c * NL;
                    p.idx[sb.g0 + e] = ss_right(zh, NL, Zh_k);
                }
            }
        }
    };

    // One slab per workgroup.  (A persistent form -- each workgroup walking a contiguous run of slabs with the loads of
    // slab s+1 re-issued into the register set the pass before had just consumed -- was built and measured in round 2:
    // bit-correct, but 227 VGPRs -> two waves per SIMD, and 2.3x SLOWER at every size, profiles/r02_persistent_k1_ab.log;
    // the interleaving of 4-5 resident workgroups per CU already provides that overlap.)
    const Slab cur = make_slab((int64_t)slab_index(d.xcd_remap));
    if (want_idx && d.shared_grid)                                                              // LES half levels -> LDS
        for (int e = 2 * tid; e < NL; e += 2 * BLOCK) *reinterpret_cast<d2 *>(lzh + e) = ld2<0>(p.zh, e, NL);
    issue_gcm(cur);
    issue_les(cur);
    stage(cur);
    __syncthreads();
    compute(cur);
}

// =================================================================================================
// K3 v2.  LDS: rec[CB][NL][6] = {t, qt, ql, ql_ice, u, v} per LES level | Zf[CB][NG] | h[NL] (shared LES grid)
// or h[CB][NL].
// =================================================================================================
template <int NG, int NL, int CB, int BLOCK, int WT>
__global__ __launch_bounds__(BLOCK) void k_backward_v2(const BwdP<double> p)
{
    static_assert(CB % 2 == 0 && NL % 2 == 0 && NL >= 2, "slab must be 16-B aligned in every array");
    constexpr int NI1 = CB * NG / 2, NI2 = CB * NL / 2;
    constexpr int IT1 = (NI1 + BLOCK - 1) / BLOCK, IT2 = (NI2 + BLOCK - 1) / BLOCK;
    constexpr int RS = 6;
    const DimsP &d = p.d;
    const int tid = threadIdx.x;
    const int64_t col0 = (int64_t)slab_index(d.xcd_remap) * CB;
    const int ncol = (int)((d.n_cols - col0) < CB ? (d.n_cols - col0) : CB);
    const int lim1 = ncol * NG, lim2 = ncol * NL;
    double *const lrec = reinterpret_cast<double *>(spc_smem);
    double *const lZf = lrec + (size_t)CB * NL * RS;
    double *const lh = lZf + (size_t)CB * NG;
    const int64_t g0 = col0 * NG, h0 = col0 * (NG + 1), o0 = col0 * NL;

    // ---- load burst: the LES slab (staged through registers into records), then the GCM side -------------------
    d2 st[IT2], sqt[IT2], sql[IT2], sqi[IT2], su[IT2], sv[IT2];
#pragma unroll
    for (int it = 0; it < IT2; ++it) {
        const int e = 2 * (it * BLOCK + ((it & 1) ? BLOCK - 1 - tid : tid));
        st[it] = ld2<WT>(p.t_d + o0, e, lim2);
        sqt[it] = ld2<WT>(p.qt_d + o0, e, lim2);
        sql[it] = ld2<WT>(p.ql_d + o0, e, lim2);
        sqi[it] = ld2<WT>(p.ql_ice_d + o0, e, lim2);
        su[it] = ld2<WT>(p.u_d + o0, e, lim2);
        sv[it] = ld2<WT>(p.v_d + o0, e, lim2);
    }
    {
        const int nz = d.shared_grid ? NL : lim2;
        for (int e = 2 * tid; e < nz; e += 2 * BLOCK)
            *reinterpret_cast<d2 *>(lh + e) = d.shared_grid ? ld2<0>(p.zf, e, NL) : ld2<WT>(p.zf + o0, e, lim2);
    }
    d2 gZ[IT1], gT[IT1], gSH[IT1], gQL[IT1], gQI[IT1], gU[IT1], gV[IT1], gA[IT1];
    double ad0[IT1], ad1[IT1];
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int e = 2 * (tid + it * BLOCK);
        const int c0 = e / NG, c1 = (e + 1) / NG;
        if (p.Zf) {
            gZ[it] = ld2<WT>(p.Zf + g0, e, lim1);
        } else {                                                                           // spcpl.py:198
            const d2 zg = ld2<WT>(p.Zgfull + g0, e, lim1);
            const double z0 = e < lim1 ? p.Zghalf[h0 + (int64_t)c0 * (NG + 1) + NG] : 0.0;
            const double z1 = e + 1 < lim1 ? p.Zghalf[h0 + (int64_t)c1 * (NG + 1) + NG] : 0.0;
            gZ[it] = d2{div_grav(zg.x - z0), div_grav(zg.y - z1)};
        }
        gT[it] = ld2<WT>(p.Tm + g0, e, lim1);
        gSH[it] = ld2<WT>(p.SH + g0, e, lim1);
        gQL[it] = ld2<WT>(p.QL + g0, e, lim1);
        gQI[it] = ld2<WT>(p.QI + g0, e, lim1);
        gU[it] = ld2<WT>(p.U + g0, e, lim1);
        gV[it] = ld2<WT>(p.V + g0, e, lim1);
        gA[it] = ld2<WT>(p.A + g0, e, lim1);
        // profile["A"][::-1] (spcpl.py:404): element (c, k) pairs with A_prof[c][NG-1-k]
        ad0[it] = e < lim1 ? p.A_prof[g0 + (int64_t)c0 * NG + (NG - 1 - (e - c0 * NG))] : 0.0;
        ad1[it] = e + 1 < lim1 ? p.A_prof[g0 + (int64_t)c1 * NG + (NG - 1 - (e + 1 - c1 * NG))] : 0.0;
    }

    // ---- stage: one record per LES level; Zf per column -------------------------------------------------------
#pragma unroll
    for (int it = 0; it < IT2; ++it) {
        const int e = 2 * (it * BLOCK + ((it & 1) ? BLOCK - 1 - tid : tid));
        if (e < lim2) {                                       // NL even: both levels belong to the same column
            d2 *const q = reinterpret_cast<d2 *>(lrec + (size_t)e * RS);
            q[0] = d2{st[it].x, sqt[it].x};
            q[1] = d2{sql[it].x, sqi[it].x};
            q[2] = d2{su[it].x, sv[it].x};
            q[3] = d2{st[it].y, sqt[it].y};
            q[4] = d2{sql[it].y, sqi[it].y};
            q[5] = d2{su[it].y, sv[it].y};
        }
    }
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int e = 2 * (tid + it * BLOCK);
        if (e < lim1) lZf[e] = gZ[it].x;
        if (e + 1 < lim1) lZf[e + 1] = gZ[it].y;
    }
    __syncthreads();

    // ---- two GCM levels per thread ----------------------------------------------------------------------------
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int e0 = 2 * (tid + it * BLOCK);
        double o[7][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int e = e0 + s;
            const int c = e < lim1 ? e / NG : 0, k = e - c * NG;
            const double *const h = d.shared_grid ? lh : lh + (size_t)c * NL;
            const double *const Zf = lZf + (size_t)c * NG;
            const double x = s ? gZ[it].y : gZ[it].x;
            const int start_index = ss_left_neg(Zf, NG, h[NL - 1]);                        // spcpl.py:498
            // record: t, qt, ql, ql_ice, u, v -> interpolate the six stored fields plus ql_water = ql - ql_ice,
            // which the reference forms on LES levels BEFORE interpolating (spcpl.py:402, 474)
            const int j = upper_count(h, NL, cfloor_pow2(NL), x) - 1;
            const bool below = j < 0, above = j >= NL - 1;
            const int jc = j < 0 ? 0 : (j > NL - 2 ? NL - 2 : j);
            const d2 *const q = reinterpret_cast<const d2 *>(lrec + ((size_t)c * NL + jc) * RS);
            const d2 a0 = q[0], a1 = q[1], a2 = q[2], b0 = q[3], b1 = q[4], b2 = q[5];
            const double x0 = h[jc], x1 = h[jc + 1];
            Br<double> b;
            b.take = below | above | (x0 == x);
            b.nanx = (x != x);
            b.x = x;
            b.x0 = b.take ? 0.0 : x0;
            b.x1 = b.take ? 1.0 : x1;
            const double lo[7] = {a0.x, a0.y, a1.x, a1.x - a1.y, a1.y, a2.x, a2.y};       // t, qt, ql, ql_water, ql_ice, u, v
            const double hi[7] = {b0.x, b0.y, b1.x, b1.x - b1.y, b1.y, b2.x, b2.y};
            double f0[7], f1[7], r[7];
#pragma unroll
            for (int f = 0; f < 7; ++f) {
                f0[f] = above ? hi[f] : lo[f];
                f1[f] = b.take ? f0[f] : hi[f];
            }
            interp_fields<7>(b, f0, f1, r);                                                // spcpl.py:471-477
            const double tt = s ? gT[it].y : gT[it].x, sh = s ? gSH[it].y : gSH[it].x, ql = s ? gQL[it].y : gQL[it].x;
            const double qi = s ? gQI[it].y : gQI[it].x, uu = s ? gU[it].y : gU[it].x, vv = s ? gV[it].y : gV[it].x;
            const double aa = s ? gA[it].y : gA[it].x, a_d = s ? ad1[it] : ad0[it];
            double f_T = SPC_DIV(p.factor * (r[0] - tt), p.dt);                                    // spcpl.py:518
            double f_SH = SPC_DIV(p.factor * ((r[1] - r[2]) - sh), p.dt);                          // spcpl.py:519
            double f_QL = SPC_DIV(p.factor * (r[3] - ql), p.dt);                                   // spcpl.py:520
            double f_QI = SPC_DIV(p.factor * (r[4] - qi), p.dt);                                   // spcpl.py:521
            double f_U = SPC_DIV(p.factor * (r[5] - uu), p.dt);                                    // spcpl.py:524
            double f_V = SPC_DIV(p.factor * (r[6] - vv), p.dt);                                    // spcpl.py:525
            double f_A = SPC_DIV(p.factor * (a_d - aa), p.dt);                                     // spcpl.py:526
            if (k < start_index) {  // `f[0:start_index] *= 0` (spcpl.py:527-533): -x -> -0, NaN stays NaN
                f_T *= 0.0; f_SH *= 0.0; f_QL *= 0.0; f_QI *= 0.0; f_U *= 0.0; f_V *= 0.0; f_A *= 0.0;
            }
            o[0][s] = f_T; o[1][s] = f_SH; o[2][s] = f_QL; o[3][s] = f_QI; o[4][s] = f_U; o[5][s] = f_V; o[6][s] = f_A;
            if (p.start_index && k == 0 && e < lim1) p.start_index[col0 + c] = start_index;
        }
        st2<WT>(p.f_T + g0, e0, lim1, d2{o[0][0], o[0][1]});
        st2<WT>(p.f_SH + g0, e0, lim1, d2{o[1][0], o[1][1]});
        st2<WT>(p.f_QL + g0, e0, lim1, d2{o[2][0], o[2][1]});
        st2<WT>(p.f_QI + g0, e0, lim1, d2{o[3][0], o[3][1]});
        st2<WT>(p.f_U + g0, e0, lim1, d2{o[4][0], o[4][1]});
        st2<WT>(p.f_V + g0, e0, lim1, d2{o[5][0], o[5][1]});
        st2<WT>(p.f_A + g0, e0, lim1, d2{o[6][0], o[6][1]});
    }
}
