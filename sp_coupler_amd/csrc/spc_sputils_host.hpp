// spc_sputils_host.hpp -- argument checks and launches of the K7 operators (kernels: spc_sputils.hpp); included by
// spc_hip.hip after its host helpers (fail, REQUIRE, launch_status, floor_pow2).
#pragma once

// ---- host side ----------------------------------------------------------------------------------
// Write-through stores for launches that leave <= 64 MiB behind.  (K1 / K3's small_batch() draws that line at the 32 MiB of
// the aggregate L2; measured on these operators -- profiles/r04_write_through_sweep.log -- write-through still wins at the
// 45.7 MB interp GCM->LES writes at 35 718 rows: 20.8 against 23.7 us, the dirty lines otherwise wait for the end-of-kernel
// release.)  SPC_FORCE_WT=0/1 overrides as for K1 / K3.
inline int su_write_through(int64_t bytes_written)
{
    if (wt_forced() >= 0) return wt_forced();
    return bytes_written <= (int64_t)64 * 1024 * 1024 ? 1 : 0;
}
#define SU_PICK_WT(KERN_WT1, KERN_WT0, bytes) (su_write_through((int64_t)(bytes)) ? (KERN_WT1) : (KERN_WT0))

template <typename T> int exner_impl(int64_t n, const void *p, void *out, int inverse, void *stream)
{
    if (n < 0) return fail(SPC_ERR_INVALID_ARGUMENT, "%sexner: n < 0");
    if (n == 0) return SPC_OK;
    REQUIRE(p, "p"); REQUIRE(out, "out");
    const int64_t per = (int64_t)SU_THREADS * SU_EX_PER;
    if ((n + per - 1) / per > 0x7fffffff) return fail(SPC_ERR_UNSUPPORTED, "%sexner: more than 2^41 elements");
    auto kern = SU_PICK_WT((k_exner<T, 1>), (k_exner<T, 0>), n * (int64_t)sizeof(T));
    hipLaunchKernelGGL(kern, dim3((unsigned)((n + per - 1) / per)), dim3(SU_THREADS), 0, (hipStream_t)stream, n, (const T *)p, (T *)out, inverse);
    return launch_status("k_exner");
}

// Rows per workgroup (the slab): enough rows for ~`target` outputs per workgroup -- 4-5 per thread, the shape of K1's
// 8-column slabs -- within the LDS budget, but never so many that the grid drops under four workgroups per CU (small
// batches are latency-bound: more, smaller workgroups).  SPC_SU_TARGET overrides for A/B runs.
// `max_pitch`: the largest row pitch (elements) of the launch -- the kernels address a slab with 32-bit BYTE offsets r * pitch + i
// off uniform bases, so rb * max_pitch * 8 must stay below 2^32 whatever chose rb (the SPC_SU_RB override included): rows
// padded to millions of elements get fewer rows per workgroup (round-4 advisor: the bound was a comment, not a check).
inline int su_rows(int64_t n_rows, int n_out, size_t lds_per_row, size_t lds_fixed, size_t esize, int *stage, int64_t max_pitch, int lds_kib = 16,
                   bool fit_rounds = true)
{
    static const int target = [] { const char *e = getenv("SPC_SU_TARGET"); const int v = e ? atoi(e) : 1100; return v < 1 ? 1 : v; }();
    static const int cap_env = [] { const char *e = getenv("SPC_SU_LDS_KIB"); return e ? atoi(e) : 0; }();
    const size_t cap = (size_t)(cap_env > 0 ? cap_env : lds_kib) * 1024;
    int rb = n_out > 0 ? (target + n_out - 1) / n_out : 1;
    if (rb < 1) rb = 1;
    if (rb > 64) rb = 64;
    const int64_t most = n_rows / (4 * (int64_t)device_cus());   // >= 4 workgroups per CU (MI355X: 1024)
    if (rb > most) rb = most < 1 ? 1 : (int)most;
    while (rb > 1 && (lds_per_row * rb + lds_fixed) * esize > cap) --rb;      // 16 KiB: >= 8 workgroups per CU (SPC_SU_LDS_KIB: A/B runs)
    // every wave of a workgroup runs ceil(rb n_out / 256) rounds of the output loop, the last one partly idle: among the
    // slab heights within two rows of that choice take the one that wastes the fewest lane-rounds (91 -> 160 levels: 8 rows =
    // exactly 5 rounds instead of 7 rows = 4.4 rounds paid as 5; the operators are bound by VALU issue)
    static const bool fit = [] { const char *e = getenv("SPC_SU_FIT"); return !e || atoi(e) != 0; }();
    if (fit && fit_rounds && n_out > 0) {
        auto waste = [&](int c) { const int64_t o = (int64_t)c * n_out, rounds = (o + SU_THREADS - 1) / SU_THREADS; return (double)(rounds * SU_THREADS - o) / (double)(rounds * SU_THREADS); };
        int best = rb;
        for (int c = rb > 2 ? rb - 2 : 1; c <= rb + 2 && c <= 64; ++c) {
            if (c > most && c > 1) break;
            if ((lds_per_row * c + lds_fixed) * esize > cap && c > rb) break;
            if (waste(c) < waste(best) - 0.02) best = c;
        }
        rb = best;
    }
    static const int forced = [] { const char *e = getenv("SPC_SU_RB"); return e ? atoi(e) : 0; }();      // A/B runs
    if (forced > 0 && forced <= 64) rb = forced;
    while (rb > 1 && (int64_t)rb * max_pitch * 8 >= ((int64_t)1 << 32)) --rb;      // (8: the widest element; output rows of searchsorted are int64)
    *stage = (lds_per_row * rb + lds_fixed) * esize <= SU_MAX_LDS;
    return rb;
}

// the kernels address a slab with 32-bit element offsets r * pitch + i (r < 64 rows, 24-bit multiply): pitches stay below 2^24
// elements -- except for a ONE-row launch, whose row index is always 0: the pitch is never multiplied (sputils.interp /
// searchsorted on one long vector of 2^24 or more points; round-4 advisor)
inline bool su_pitch_ok(int64_t n_rows, std::initializer_list<int64_t> pitches)
{
    if (n_rows <= 1) return true;
    for (int64_t p : pitches) if (p >= ((int64_t)1 << 24)) return false;
    return true;
}
inline int64_t su_max(std::initializer_list<int64_t> v) { int64_t m = 0; for (int64_t x : v) if (x > m) m = x; return m; }

// template argument SL of the staged kernels for a row whose power-of-two floor is p2: log2 p2 + 1 where that depth is
// instantiated (rows of 64-127, 128-255, 256-511, 512-1023 entries: every level count of BASELINE.json's configs), else 0
// (the same descent with the trip count at run time)
inline int su_sl(int p2) { return p2 == 64 ? 7 : p2 == 128 ? 8 : p2 == 256 ? 9 : p2 == 512 ? 10 : 0; }

// (the unrolled depths exist for double; the float twin runs the run-time descent)
template <typename T, int WT> auto interp_kernel(int sl) -> void (*)(const SuInterpP)
{
    if constexpr (sizeof(T) == 8) {
        if (sl == 7) return k_interp<T, 7, WT>;
        if (sl == 8) return k_interp<T, 8, WT>;
        if (sl == 9) return k_interp<T, 9, WT>;
        if (sl == 10) return k_interp<T, 10, WT>;
    }
    return sl >= 0 ? k_interp<T, 0, WT> : k_interp<T, -1, WT>;
}

template <typename T, int WT> auto searchsorted_kernel(int sl) -> void (*)(const SuSearchP)
{
    if constexpr (sizeof(T) == 8) {
        if (sl == 7) return k_searchsorted<T, 7, WT>;
        if (sl == 8) return k_searchsorted<T, 8, WT>;
        if (sl == 9) return k_searchsorted<T, 9, WT>;
        if (sl == 10) return k_searchsorted<T, 10, WT>;
    }
    return sl >= 0 ? k_searchsorted<T, 0, WT> : k_searchsorted<T, -1, WT>;
}

template <typename T> int interp_impl(const spc_interp_args *a, void *stream)
{
    if (!a) return fail(SPC_ERR_INVALID_ARGUMENT, "%sargs is NULL");
    if (a->n_rows < 0 || a->n_x < 0 || a->n_xp < 0) return fail(SPC_ERR_INVALID_ARGUMENT, "%sinterp: negative extent");
    if (a->n_rows == 0 || a->n_x == 0) return SPC_OK;
    if (a->n_xp == 0) return fail(SPC_ERR_INVALID_ARGUMENT, "%sinterp: array of sample points is empty");   // numpy: ValueError
    REQUIRE(a->x, "x"); REQUIRE(a->xp, "xp"); REQUIRE(a->fp, "fp"); REQUIRE(a->out, "out");
    if ((a->pitch_x && a->pitch_x < a->n_x) || (a->pitch_xp && a->pitch_xp < a->n_xp) || a->pitch_fp < a->n_xp || a->pitch_out < a->n_x)
        return fail(SPC_ERR_INVALID_ARGUMENT, "%sinterp: a pitch is smaller than its row (only x and xp may be shared, pitch 0)");
    if (!su_pitch_ok(a->n_rows, {a->pitch_x, a->pitch_xp, a->pitch_fp, a->pitch_out})) return fail(SPC_ERR_UNSUPPORTED, "%sinterp: a row pitch of 2^24 elements or more");
    SuInterpP q;
    q.n_rows = a->n_rows; q.pitch_x = a->pitch_x; q.pitch_xp = a->pitch_xp; q.pitch_fp = a->pitch_fp; q.pitch_out = a->pitch_out;
    q.n_x = a->n_x; q.n_xp = a->n_xp; q.p2 = floor_pow2(a->n_xp);
    int stage;
    const size_t xrow = (size_t)su_pad(q.p2);            // a padded xp row in LDS
    const size_t per_row = (size_t)a->n_xp + (a->pitch_xp ? xrow : 0), fixed = a->pitch_xp ? 0 : xrow;
    q.rb = su_rows(a->n_rows, a->n_x, per_row, fixed, sizeof(T), &stage, a->n_rows > 1 ? su_max({a->pitch_x, a->pitch_xp, a->pitch_fp, a->pitch_out}) : 0);
    if ((a->n_rows + q.rb - 1) / q.rb > 0x7fffffff) return fail(SPC_ERR_UNSUPPORTED, "%sinterp: too many rows for one launch");
    q.x = a->x; q.xp = a->xp; q.fp = a->fp; q.out = a->out;
    const size_t smem = stage ? (per_row * q.rb + fixed) * sizeof(T) : 0;
    const int64_t wr = a->n_rows * (int64_t)a->n_x * (int64_t)sizeof(T);
    const int sl = stage ? su_sl(q.p2) : -1;
    void (*kern)(const SuInterpP) = su_write_through(wr) ? interp_kernel<T, 1>(sl) : interp_kernel<T, 0>(sl);
    hipLaunchKernelGGL(kern, dim3((unsigned)((a->n_rows + q.rb - 1) / q.rb)), dim3(SU_THREADS), smem, (hipStream_t)stream, q);
    return launch_status("k_interp");
}

template <typename T> int searchsorted_impl(const spc_searchsorted_args *a, void *stream)
{
    if (!a) return fail(SPC_ERR_INVALID_ARGUMENT, "%sargs is NULL");
    if (a->n_rows < 0 || a->n_a < 0 || a->n_v < 0) return fail(SPC_ERR_INVALID_ARGUMENT, "%ssearchsorted: negative extent");
    if (a->n_rows == 0 || a->n_v == 0) return SPC_OK;
    REQUIRE(a->v, "v"); REQUIRE(a->out, "out");
    if (a->n_a) REQUIRE(a->a, "a");
    if ((a->pitch_a && a->pitch_a < a->n_a) || (a->pitch_v && a->pitch_v < a->n_v) || a->pitch_out < a->n_v)
        return fail(SPC_ERR_INVALID_ARGUMENT, "%ssearchsorted: a pitch is smaller than its row (only a and v may be shared, pitch 0)");
    if (!su_pitch_ok(a->n_rows, {a->pitch_a, a->pitch_v, a->pitch_out})) return fail(SPC_ERR_UNSUPPORTED, "%ssearchsorted: a row pitch of 2^24 elements or more");
    SuSearchP q;
    q.n_rows = a->n_rows; q.pitch_a = a->pitch_a; q.pitch_v = a->pitch_v; q.pitch_out = a->pitch_out;
    q.n_a = a->n_a; q.n_v = a->n_v; q.right = a->side_right != 0; q.p2 = floor_pow2(a->n_a);
    int stage;
    const size_t arow = (size_t)su_pad(q.p2);
    q.rb = su_rows(a->n_rows, a->n_v, a->pitch_a ? arow : 0, a->pitch_a ? 0 : arow, sizeof(T), &stage,
                   a->n_rows > 1 ? su_max({a->pitch_a, a->pitch_v, a->pitch_out}) : 0);
    if ((a->n_rows + q.rb - 1) / q.rb > 0x7fffffff) return fail(SPC_ERR_UNSUPPORTED, "%ssearchsorted: too many rows for one launch");
    q.a = a->a; q.v = a->v; q.out = a->out;
    const size_t smem = stage ? ((a->pitch_a ? arow : 0) * q.rb + (a->pitch_a ? 0 : arow)) * sizeof(T) : 0;
    const int64_t wr = a->n_rows * (int64_t)a->n_v * 8;
    const int sl = stage ? su_sl(q.p2) : -1;
    void (*kern)(const SuSearchP) = su_write_through(wr) ? searchsorted_kernel<T, 1>(sl) : searchsorted_kernel<T, 0>(sl);
    hipLaunchKernelGGL(kern, dim3((unsigned)((a->n_rows + q.rb - 1) / q.rb)), dim3(SU_THREADS), smem, (hipStream_t)stream, q);
    return launch_status("k_searchsorted");
}

// the instantiation of k_interp_c for (PD, SL, WEIGHTED, WT); the unrolled sum depths exist for the staged double kernel
template <typename T, int SL, bool W, int WT> auto interp_c_kernel(int pd) -> void (*)(const SuCoarseP)
{
    if constexpr (sizeof(T) == 8 && SL >= 0) {
        if (pd == 1) return k_interp_c<T, 1, SL, W, WT>;
        if (pd == 2) return k_interp_c<T, 2, SL, W, WT>;
        if (pd == 3) return k_interp_c<T, 3, SL, W, WT>;
    }
    (void)pd;
    return k_interp_c<T, -1, SL, W, WT>;
}

template <typename T, bool W, int WT> auto interp_c_kernel_sl(int sl, int pd) -> void (*)(const SuCoarseP)
{
    if constexpr (sizeof(T) == 8) {              // nL = 160 -> rows z[1:] of 159 entries: SL 8; nL = 512: SL 9
        if (sl == 8) return interp_c_kernel<T, 8, W, WT>(pd);
        if (sl == 9) return interp_c_kernel<T, 9, W, WT>(pd);
    }
    return sl >= 0 ? interp_c_kernel<T, 0, W, WT>(pd) : interp_c_kernel<T, -1, W, WT>(pd);
}

template <typename T> int interp_c_impl(const spc_interp_c_args *a, void *stream)
{
    if (!a) return fail(SPC_ERR_INVALID_ARGUMENT, "%sargs is NULL");
    if (a->n_rows < 0 || a->nG < 0) return fail(SPC_ERR_INVALID_ARGUMENT, "%sinterp_c: negative extent");
    if (a->mode < SU_INTERP_C || a->mode > SU_INTEGRAL) return fail(SPC_ERR_INVALID_ARGUMENT, "%sinterp_c: mode must be 0, 1 or 2");
    if (a->n_rows == 0 || a->nG == 0) return SPC_OK;
    if (a->nL < 2) return fail(SPC_ERR_INVALID_ARGUMENT, "%sinterp_c: the fine grid needs at least 2 points");
    REQUIRE(a->Zh, "Zh"); REQUIRE(a->zh, "zh"); REQUIRE(a->q, "q"); REQUIRE(a->out, "out");
    if (a->mode == SU_INTERP_C && !a->rho) return fail(SPC_ERR_INVALID_ARGUMENT, "%sinterp_c: rho is NULL");
    if (a->n_rows > 0x7fffffff) return fail(SPC_ERR_UNSUPPORTED, "%sinterp_c: more than 2^31-1 rows");
    if (a->pitch_Zh < a->nG + 1 || (a->pitch_zh && a->pitch_zh < a->nL) || a->pitch_q < a->nL - 1 || a->pitch_out < a->nG)
        return fail(SPC_ERR_INVALID_ARGUMENT, "%sinterp_c: a pitch is smaller than its row (only zh may be shared, pitch 0)");
    if (!su_pitch_ok(a->n_rows, {a->pitch_Zh, a->pitch_zh, a->pitch_q, a->pitch_out})) return fail(SPC_ERR_UNSUPPORTED, "%sinterp_c: a row pitch of 2^24 elements or more");
    SuCoarseP q;
    q.n_rows = a->n_rows; q.pitch_Zh = a->pitch_Zh; q.pitch_zh = a->pitch_zh; q.pitch_q = a->pitch_q; q.pitch_out = a->pitch_out;
    q.nG = a->nG; q.nL = a->nL; q.mode = a->mode;
    q.Zh = a->Zh; q.zh = a->zh; q.q = a->q; q.rho = a->mode == SU_INTERP_RHO ? nullptr : a->rho; q.out = a->out;
    const bool weighted = q.rho != nullptr;
    q.p2 = floor_pow2(a->nL - 1);
    // LDS per row: the cell terms tn (and td with weights) [nL - 1], the coarse levels [nG + 1], the results [nG] + the row's
    // padded grid unless the grid is shared
    const size_t zrow = (size_t)su_pad(q.p2);
    const size_t per_row = (size_t)(a->nL - 1) * (weighted ? 2 : 1) + (size_t)(2 * a->nG + 1) + (a->pitch_zh ? zrow : 0), fixed = a->pitch_zh ? 0 : zrow;
    int stage;
    // 32 KiB (five workgroups per CU, what the registers allow): the layer-major walk skips whole waves above the fine grid's top, the better the more rows a wave spans (measured
    // at 35 718 rows: 5 rows 40.1 us, 7-9 rows 35.4-36.3, 12 rows 37.6: profiles/r04_k7_slab_sweep.log)
    q.rb = su_rows(a->n_rows, a->nG, per_row, fixed, sizeof(T), &stage, a->n_rows > 1 ? su_max({a->pitch_Zh, a->pitch_zh, a->pitch_q, a->pitch_out}) : 0, 32, false);
    const size_t smem = stage ? (per_row * q.rb + fixed) * sizeof(T) : 0;
    // numpy's pairwise recursion unrolled to the depth a layer of <= nL - 1 cells needs (cons_depth, as K4); the float twin
    // and grids of more than 1024 points keep the explicit stack
    const int pd = sizeof(T) == 8 ? cons_depth(a->nL) : -1;
    const bool wt = su_write_through(a->n_rows * (int64_t)a->nG * (int64_t)sizeof(T)) != 0;
    const int sl = stage ? su_sl(q.p2) : -1;
    void (*kern)(const SuCoarseP) = weighted ? (wt ? interp_c_kernel_sl<T, true, 1>(sl, pd) : interp_c_kernel_sl<T, true, 0>(sl, pd))
                                             : (wt ? interp_c_kernel_sl<T, false, 1>(sl, pd) : interp_c_kernel_sl<T, false, 0>(sl, pd));
    hipLaunchKernelGGL(kern, dim3((unsigned)((a->n_rows + q.rb - 1) / q.rb)), dim3(SU_THREADS), smem, (hipStream_t)stream, q);
    return launch_status("k_interp_c");
}

template <typename T> int rms_impl(int64_t n_rows, int64_t n, int64_t pitch, const void *a, void *out, void *stream)
{
    if (n_rows < 0 || n < 0) return fail(SPC_ERR_INVALID_ARGUMENT, "%srms: negative extent");
    if (n_rows == 0) return SPC_OK;
    if (n > 0x7fffffff) return fail(SPC_ERR_UNSUPPORTED, "%srms: more than 2^31-1 elements per row");
    REQUIRE(out, "out");
    if (n) REQUIRE(a, "a");
    if (pitch < n) return fail(SPC_ERR_INVALID_ARGUMENT, "%srms: pitch smaller than the row");
    const int64_t grid = (n_rows + SU_THREADS / 8 - 1) / (SU_THREADS / 8);
    if (grid > 0x7fffffff) return fail(SPC_ERR_UNSUPPORTED, "%srms: too many rows for one launch");
    // depth of numpy's recursion over one row (a single chunk up to 8192 elements); -1: explicit stack, any length
    const int pd = n <= 128 ? 0 : (n <= 1024 ? cons_depth((int)n) : -1);
    void (*kern)(int64_t, int, int64_t, const T *, T *) =
        pd == 0 ? k_rms<T, 0, 1> : pd == 1 ? k_rms<T, 1, 1> : pd == 2 ? k_rms<T, 2, 1> : pd == 3 ? k_rms<T, 3, 1> : k_rms<T, -1, 1>;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(SU_THREADS), 0, (hipStream_t)stream, n_rows, (int)n, pitch, (const T *)a, (T *)out);
    return launch_status("k_rms");
}
