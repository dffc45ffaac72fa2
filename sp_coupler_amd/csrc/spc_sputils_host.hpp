// spc_sputils_host.hpp -- argument checks and launches of the K7 operators (kernels: spc_sputils.hpp); included by
// spc_hip.hip after its host helpers (fail, REQUIRE, launch_status, floor_pow2).
#pragma once

// ---- host side ----------------------------------------------------------------------------------
inline unsigned su_grid(int64_t items, int per_block, unsigned cap)
{
    const int64_t g = (items + per_block - 1) / per_block;
    return (unsigned)(g < 1 ? 1 : (g > (int64_t)cap ? cap : g));
}

template <typename T> int exner_impl(int64_t n, const void *p, void *out, int inverse, void *stream)
{
    if (n < 0) return fail(SPC_ERR_INVALID_ARGUMENT, "%sexner: n < 0");
    if (n == 0) return SPC_OK;
    REQUIRE(p, "p"); REQUIRE(out, "out");
    hipLaunchKernelGGL(k_exner<T>, dim3(su_grid(n, SU_THREADS, 256 * 16)), dim3(SU_THREADS), 0, (hipStream_t)stream, n,
                       (const T *)p, (T *)out, inverse);
    return launch_status("k_exner");
}

// rows per workgroup: enough to give the workgroup's threads two outputs each (measured over 1, 2, 4, 8 at 35 718 rows: 2 is the fastest or within 3 %; fewer
// workgroups; SPC_SU_ITEMS overrides for A/B runs), within the LDS budget
inline int su_rows_per_block(int n_out, size_t lds_per_row, size_t lds_fixed, size_t esize, int *stage)
{
    static const int items = [] { const char *e = getenv("SPC_SU_ITEMS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v; }();
    int rb = n_out > 0 ? (SU_THREADS * items + n_out - 1) / n_out : 1;
    if (rb < 1) rb = 1;
    if (rb > 64) rb = 64;
    while (rb > 1 && (lds_per_row * rb + lds_fixed) * esize > SU_MAX_LDS) --rb;
    *stage = (lds_per_row * rb + lds_fixed) * esize <= SU_MAX_LDS;
    return rb;
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
    SuInterpP q;
    q.n_rows = a->n_rows; q.pitch_x = a->pitch_x; q.pitch_xp = a->pitch_xp; q.pitch_fp = a->pitch_fp; q.pitch_out = a->pitch_out;
    q.n_x = a->n_x; q.n_xp = a->n_xp; q.p2 = floor_pow2(a->n_xp);
    q.rb = su_rows_per_block(a->n_x, (size_t)a->n_xp * (a->pitch_xp ? 2 : 1), a->pitch_xp ? 0 : a->n_xp, sizeof(T), &q.stage);
    q.x = a->x; q.xp = a->xp; q.fp = a->fp; q.out = a->out;
    const size_t smem = q.stage ? ((size_t)a->n_xp * (a->pitch_xp ? 2 : 1) * q.rb + (a->pitch_xp ? 0 : a->n_xp)) * sizeof(T) : 0;
    hipLaunchKernelGGL(k_interp<T>, dim3((unsigned)((a->n_rows + q.rb - 1) / q.rb)), dim3(SU_THREADS), smem, (hipStream_t)stream, q);
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
    SuSearchP q;
    q.n_rows = a->n_rows; q.pitch_a = a->pitch_a; q.pitch_v = a->pitch_v; q.pitch_out = a->pitch_out;
    q.n_a = a->n_a; q.n_v = a->n_v; q.right = a->side_right != 0;
    q.rb = su_rows_per_block(a->n_v, a->pitch_a ? a->n_a : 0, a->pitch_a ? 0 : a->n_a, sizeof(T), &q.stage);
    q.a = a->a; q.v = a->v; q.out = a->out;
    const size_t smem = q.stage ? ((size_t)(a->pitch_a ? a->n_a : 0) * q.rb + (a->pitch_a ? 0 : a->n_a)) * sizeof(T) : 0;
    hipLaunchKernelGGL(k_searchsorted<T>, dim3((unsigned)((a->n_rows + q.rb - 1) / q.rb)), dim3(SU_THREADS), smem, (hipStream_t)stream, q);
    return launch_status("k_searchsorted");
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
    SuCoarseP q;
    q.n_rows = a->n_rows; q.pitch_Zh = a->pitch_Zh; q.pitch_zh = a->pitch_zh; q.pitch_q = a->pitch_q; q.pitch_out = a->pitch_out;
    q.nG = a->nG; q.nL = a->nL; q.mode = a->mode;
    q.Zh = a->Zh; q.zh = a->zh; q.q = a->q; q.rho = a->mode == SU_INTERP_RHO ? nullptr : a->rho; q.out = a->out;
    q.rb = su_rows_per_block(a->nG, (size_t)a->nL * (a->pitch_zh ? 3 : 2), a->pitch_zh ? 0 : a->nL, sizeof(T), &q.stage);
    const size_t smem = q.stage ? ((size_t)a->nL * (a->pitch_zh ? 3 : 2) * q.rb + (a->pitch_zh ? 0 : a->nL)) * sizeof(T) : 0;
    // numpy's pairwise recursion unrolled to the depth a layer of <= nL - 1 cells needs (cons_depth, as K4); the float twin
    // and grids of more than 1024 points keep the explicit stack
    const int pd = sizeof(T) == 8 ? cons_depth(a->nL) : -1;
    void (*kern)(const SuCoarseP) = q.stage ? k_interp_c<T, -1, true> : k_interp_c<T, -1, false>;
    if constexpr (sizeof(T) == 8) {
        if (q.stage) kern = pd == 1 ? k_interp_c<T, 1, true> : pd == 2 ? k_interp_c<T, 2, true> : pd == 3 ? k_interp_c<T, 3, true> : kern;
    }
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
    hipLaunchKernelGGL(k_rms<T>, dim3((unsigned)((n_rows + 63) / 64)), dim3(64), 0, (hipStream_t)stream, n_rows, (int)n, pitch,
                       (const T *)a, (T *)out);
    return launch_status("k_rms");
}
