// spc_f32v.hpp -- K1 of the fp32 arithmetic variant with 8-BYTE accesses (two adjacent elements per lane); included by
// spc_hip.hip after k_forward, whose device functions (bracket2, interp_fields, Divisor, ss_right, spc_pow, stg) it uses unchanged.
//
// Why: a 256-MiB device copy runs at 4.3-4.7 TB/s with 4 bytes per lane and at 5.7-5.8 TB/s with 8 (tools/copy_width.py,
// profiles/r05_copy_width.log), and the scalar float kernels sat AT the 4-byte rate (K1 4.85, K3 5.2 TB/s at config 3), which
// made the access width the suspect.  Here a lane moves float2: two adjacent levels of ONE column on the LES
// side (nL is even in every compile-time geometry), two adjacent elements of the flat [ncol x nG] slab on the GCM side (nG is
// odd -- 91, 137, 19 --, so a pair may straddle two columns: each half carries its own (column, level)).  Same arithmetic,
// same order, same bits as k_forward<float, ...> (tests/test_parity_gpu.py compares them).
// Measured (profiles/r05_f32_vec_ab.log): K1 77.0 against 78.7 us at config 3, 331 against 341 us at 174 264 columns, 441-445
// against 448 at config 5 -- 2-3 %, far less than the copy rates promise: at these sizes the kernel is bound as much by its
// items (LDS searches, 12 work items per thread) as by its bytes.  The same form of K3 was built, bit-checked and measured
// 13 % SLOWER (74 against 66 us: K3 keeps two columns per workgroup, i.e. 91 pairs for 256 threads) and was removed again.
//
// Conditions the launcher checks (else the scalar kernels run): compile-time geometry (contiguous columns), an EVEN number
// of columns per workgroup (every slab then starts on an even element of the odd-pitched GCM arrays), every array base
// 8-byte aligned, lean outputs, multi-round launches (no 512 / 1024-thread form, no prologue prefetch).
#pragma once

struct F2 { float x, y; };

__device__ __forceinline__ F2 ld2(const float *q)
{
    const float2 v = *reinterpret_cast<const float2 *>(q);
    return {v.x, v.y};
}
template <int WT> __device__ __forceinline__ void st2(float *q, float a, float b)
{
    if constexpr (WT == 1) {
        unsigned long long bits = ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(q), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *reinterpret_cast<float2 *>(q) = make_float2(a, b);
    }
}

// ---- K1 -----------------------------------------------------------------------------------------------------------------
// Waves per SIMD the register allocator is asked to fit: eight (<= 64 VGPRs, <= 96 SGPRs: 45 scalar spills into VGPR lanes) for
// the large 160-level launches -- 74.6-75.0 against 77.6 us at config 3 -- and no request elsewhere (137 <-> 512: 446 against
// 443 us; write-through launches of a few thousand columns: 13.6 against 13.0 us) (profiles/r05_f32_vec_ab.log, second block).
#ifndef SPC_F32V_WAVES
#define SPC_F32V_WAVES(NL_, WT_) (((NL_) <= 160 && (WT_) == 0) ? 8 : 1)
#endif
template <int NG, int NL, int WT> __global__ __launch_bounds__(BLOCK, SPC_F32V_WAVES(NL, WT)) void k_forward_f32v(const FwdP<float, false> p)
{
    using T = float;
    static_assert(NG > 0 && NL > 0 && NL % 2 == 0, "compile-time geometry with an even LES level count");
    const DimsP &d = p.d;
    constexpr int nG = NG, nL = NL, nLh = NL / 2, p2G = cfloor_pow2(NG);
    constexpr int64_t pitchG = NG, pitchGh = NG + 1, pitchL = NL;
    const int cb = d.cb, tid = threadIdx.x;
    const int64_t col0 = (int64_t)slab_index(d.xcd_remap) * cb;
    const int ncol = (int)((d.n_cols - col0) < cb ? (d.n_cols - col0) : cb);
    T *const lds = reinterpret_cast<T *>(spc_smem);
    T *const lzh = lds + (size_t)cb * 6 * nG;
    const int n1 = ncol * nG, n2h = ncol * nLh, nI = p.idx ? n1 : 0, nitems = n2h + nI;

    if (p.idx) {  // stage the LES half levels for the fused index map
        const int nz = d.shared_grid ? nL : ncol * nL;
        for (int e = tid; e < nz; e += BLOCK) {
            const int c = e / nL, l = e - c * nL;
            lzh[e] = d.shared_grid ? p.zh[e] : p.zh[(col0 + c) * pitchL + l];
        }
    }

    // ---- phase 1: the flat [ncol x nG] slab two elements at a time (spcpl.py:198, 214-215, 224-228) ----------------------
    auto stage = [&](int f, T tt, T sh, T ql, T qi, T pf, T zg, T uu, T vv, T zsurf) {
        const int c = f / nG, k = f - c * nG;
        const T zf_k = div_grav(zg - zsurf);                                          // spcpl.py:198
        T *const s = lds + (size_t)c * 6 * nG + (nG - 1 - k);                         // [::-1], spcpl.py:224
        s[0] = zf_k;
        s[2 * nG] = sh + ql + qi;                                                     // spcpl.py:215
        s[3 * nG] = ql;
        s[4 * nG] = uu;
        s[5 * nG] = vv;
        const T iex = spc_pow(div_pref0(pf), (-K<T>::rd) / K<T>::cp);                 // sputils.py:34
        s[nG] = (tt - div_cp(K<T>::rlv * (ql + qi))) * iex;                           // spcpl.py:214
    };
    const int64_t gbase = col0 * pitchG;                                              // even: cb is even
    for (int e = tid; 2 * e < n1; e += BLOCK) {
        const int f = 2 * e;
        const int64_t g = gbase + f;
        const int c0 = f / nG, c1 = (f + 1) / nG;
        const T zs0 = ldg(&p.Zghalf[(col0 + c0) * pitchGh + nG]);
        if (f + 1 < n1) {
            const T zs1 = ldg(&p.Zghalf[(col0 + c1) * pitchGh + nG]);
            const F2 tt = ld2(&p.Tm[g]), sh = ld2(&p.SH[g]), ql = ld2(&p.QL[g]), qi = ld2(&p.QI[g]), pf = ld2(&p.Pf[g]), zg = ld2(&p.Zgfull[g]);
            const F2 uu = ld2(&p.U[g]), vv = ld2(&p.V[g]);
            stage(f, tt.x, sh.x, ql.x, qi.x, pf.x, zg.x, uu.x, vv.x, zs0);
            stage(f + 1, tt.y, sh.y, ql.y, qi.y, pf.y, zg.y, uu.y, vv.y, zs1);
        } else {                                                                      // the odd tail of the batch's last slab
            stage(f, ldg(&p.Tm[g]), ldg(&p.SH[g]), ldg(&p.QL[g]), ldg(&p.QI[g]), ldg(&p.Pf[g]), ldg(&p.Zgfull[g]), ldg(&p.U[g]), ldg(&p.V[g]), zs0);
        }
    }
    __syncthreads();

    // ---- per-column scalars (spcpl.py:246, 332) ---------------------------------------------------------------------------
    const SPC_DIVISOR(T) ddt(p.dt);
    {
        const int sc = BLOCK - 1 - tid;
        if (sc < ncol) {
            const int64_t col = col0 + sc;
            const T ps = ldg(&p.Ph[col * pitchGh + nG]), psd = ldg(&p.ps_d[col]);
            stg<WT>(&p.f_ps[col], ddt.div(p.factor * (ps - psd)));
        }
    }

    // ---- phase 2: two adjacent LES levels of one column per item; then the index-map entries (scalar) ---------------------
    auto level = [&](const T *s, T h, T ud, T vd, T thld, T qtd, T qld, T (&o)[6]) {
        const Br<T> b = bracket2(s, nG, p2G, h);
        T f0[5], f1[5], r[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            f0[k] = s[(k + 1) * nG + b.j0];
            f1[k] = s[(k + 1) * nG + b.j1];
        }
        interp_fields<5>(b, f0, f1, r);
        const T thl = r[0], qt = r[1], ql = r[2], u = r[3], v = r[4];                  // spcpl.py:224-228
        o[0] = ddt.div(p.factor * (u - ud));                                          // spcpl.py:328
        o[1] = ddt.div(p.factor * (v - vd));                                          // spcpl.py:329
        o[2] = ddt.div(p.factor * (thl - thld));                                      // spcpl.py:330
        o[3] = ddt.div(p.factor * (qt - qtd));                                        // spcpl.py:331
        o[4] = ddt.div(p.factor * (ql - qld));                                        // spcpl.py:333
        o[5] = ql;                                                                    // spcpl.py:347-348
    };
    for (int e = tid; e < nitems; e += BLOCK) {
        if (e < n2h) {
            const int c = e / nLh, l = 2 * (e - c * nLh);
            const int64_t o = (col0 + c) * pitchL + l;
            const T *const s = lds + (size_t)c * 6 * nG;
            const F2 h = d.shared_grid ? ld2(&p.zf[l]) : ld2(&p.zf[o]);                // spcpl.py:222
            const F2 ud = ld2(&p.u_d[o]), vd = ld2(&p.v_d[o]), thld = ld2(&p.thl_d[o]), qtd = ld2(&p.qt_d[o]), qld = ld2(&p.ql_d[o]);
            T a[6], b[6];
            level(s, h.x, ud.x, vd.x, thld.x, qtd.x, qld.x, a);
            level(s, h.y, ud.y, vd.y, thld.y, qtd.y, qld.y, b);
            st2<WT>(&p.f_u[o], a[0], b[0]);
            st2<WT>(&p.f_v[o], a[1], b[1]);
            st2<WT>(&p.f_thl[o], a[2], b[2]);
            st2<WT>(&p.f_qt[o], a[3], b[3]);
            st2<WT>(&p.f_ql[o], a[4], b[4]);
            st2<WT>(&p.ql_ref[o], a[5], b[5]);
        } else {                                                                      // fused K2, spcpl.py:764
            const int ei = e - n2h, c = ei / nG, m = ei - c * nG;
            const int64_t col = col0 + c, gh = col * pitchGh;
            const T Zh_k = div_grav(ldg(&p.Zghalf[gh + (nG - 1 - m)]) - ldg(&p.Zghalf[gh + nG]));      // spcpl.py:197
            const T *const zh = d.shared_grid ? lzh : lzh + (size_t)c * nL;
            p.idx[col * pitchG + m] = ss_right(zh, nL, Zh_k);
        }
    }
}
