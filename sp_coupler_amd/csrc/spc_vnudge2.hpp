// spc_vnudge2.hpp -- the kernels of the variability nudge (K6): transpose, solve, update, std.
//
// A sweep of a level's qt / qsat plane per evaluation of the root finder (~30 per level) is what the reference does and
// what round 1's kernel did from memory.  Here a workgroup owns KT levels of one column, KT chosen so that their qt and
// qsat planes (KT x nij x 16 B) FIT THE CU's LDS (64 x 64 planes: KT = 2, 128 KiB); it loads them once -- coalesced, from
// contiguous planes k_vnudge_transpose wrote into the caller's workspace -- and runs every evaluation from LDS.  The 512
// threads split into KT groups, one per level; inside a group 8 consecutive lanes carry the 8 accumulators of ONE numpy
// pairwise leaf (<= 128 elements; lane j sums elements j, j+8, ...) and combine them ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
// by shuffles -- the order numpy uses -- so a 4096-point plane is summed by 256 threads in 16 steps; the leaf sums are
// then combined in numpy's tree order by dependency rounds (vn_build_rounds).  scipy's brentq runs as a resumable step
// function, one evaluation per level and round.  Planes too large for the LDS (128 x 128 and up) are streamed from the
// workspace instead, one workgroup per level (k_vnudge_solve<true>).  The solve leaves beta / a and the apply code in
// `status`; k_vnudge_update rewrites qt (thl with constantT) elementwise and k_vnudge_std takes qt.std(axis=(0,1)) in
// numpy's sequential order.  Bit-identical to the NumPy / SciPy oracle on every path (tests/test_vnudge.py).
#pragma once

constexpr int VN2_THREADS = 512;
constexpr int VN2_MAX_LDS = 150 * 1024;           // planes + scratch per workgroup (of the CU's 160 KiB)
enum { VN2_APPLY_MULT = 1 << 20, VN2_APPLY_ADD = 1 << 21, VN2_TOUCHED = 1 << 22, VN2_INTERNAL = 7 << 20 };

// LDS position of plane element e: 8 doubles of skew per 128 elements, so that the 8 leaf groups of a wave (leaf starts
// 128 elements = 1 KiB apart, i.e. on the SAME banks) read 8 different 64-B bank ranges (measured without it: 8 us per
// evaluation round, LDS-conflict bound)
__device__ __forceinline__ int vn2_pos(int e) { return e + ((e >> 7) << 3); }
__host__ __device__ inline int vn2_plane(int nij) { return nij + ((nij >> 7) << 3) + 8; }

// numpy's pairwise tree of the two chunk shapes (0 = a full 8192-element chunk, 1 = the last or only chunk), flattened
// ON THE HOST (vn_build_tree walks a stack: in a kernel that stack is scratch memory, 90 us per workgroup) and handed to
// the kernel in its argument block: leaves (lo, n), the post-order combine program slot[pl] += slot[pr], and the
// dependency round of every step
struct Vn2Tables {
    unsigned short lo[2][VN_MAXLEAF], n[2][VN_MAXLEAF], pl[2][VN_MAXLEAF], pr[2][VN_MAXLEAF];
    unsigned char rnd[2][VN_MAXLEAF];
    int nleaf[2], nround[2];
    int balanced[2];      // the tree is the perfectly balanced one over a power of two of <= 64 leaves (vn_tree_balanced)
};

// Is the combine program the balanced binary tree over nleaf = 2^m <= 64 leaves -- every step slot[l] += slot[l + d] with d a
// power of two, l a multiple of 2 d, in dependency round log2 d + 1?  (numpy's pairwise split halves exactly whenever the
// length is 128 x 2^m: 64 x 64 planes, the 8192-element chunks of larger ones.)  Then ONE wave combines the leaves in
// registers: lane i holds leaf i, v = v + shfl_down(v, d) for d = 1, 2, 4, ...: the same additions in the same order
// (left + right), without the LDS read-add-write round trips of the general program (1.0 us of a 2.2-us evaluation round:
// tools/stamps_k6.py, profiles/r05_k6_stamps.log).
__host__ inline int vn_tree_balanced(int nleaf, const unsigned short *pl, const unsigned short *pr, const unsigned char *rnd)
{
    if (nleaf < 2 || nleaf > 64 || (nleaf & (nleaf - 1))) return 0;
    for (int t = 0; t + 1 < nleaf; ++t) {
        const int d = (int)pr[t] - (int)pl[t];
        if (d <= 0 || (d & (d - 1)) || (pl[t] % (2 * d)) != 0 || (1 << (rnd[t] - 1)) != d) return 0;
    }
    return 1;
}

struct Vn2P {
    VnP p;
    int kt, log2_kt, tiles, gpc, tg;    // levels per workgroup, tiles per column, line groups per column, tiles per group
    int64_t groups;
    int nleaf_max;
    const double *work;                  // planes transposed to [col][field][k][ij] by k_vnudge_transpose, or NULL
    Vn2Tables tab;
};

// K6t: qt and qsat of every column from the reference's [ij][k] order into contiguous planes [col][field][k][ij] of the
// caller's workspace: the solve then fills its LDS with coalesced loads (128 KiB = 1 024 line requests) instead of one
// 128-B line request per (ij, field) for 16 useful bytes (8 192 requests, 90 us per workgroup at the per-CU limit of
// outstanding requests).  Tile = 64 rows x 16 levels through LDS; reads and writes are whole 128-B segments.
__global__ __launch_bounds__(256) void k_vnudge_transpose(const VnP p, double *work)
{
    __shared__ double s_t[16][65];
    const int lane = threadIdx.x & 15, row = threadIdx.x >> 4;
    const int ij0 = blockIdx.x * 64, k0 = blockIdx.y * 16, nij = p.nij, ktot = p.ktot;
    const int64_t col = blockIdx.z >> 1;
    const int field = blockIdx.z & 1;
    const double *const src = (field ? p.qsat : (const double *)p.qt) + col * (int64_t)nij * ktot;
    double *const dst = work + (col * 2 + field) * (int64_t)ktot * nij;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int ij = ij0 + row + 16 * u, k = k0 + lane;
        s_t[lane][row + 16 * u] = (ij < nij && k < ktot) ? src[(int64_t)ij * ktot + k] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k0 + row, ij = ij0 + lane + 16 * u;
        if (ij < nij && k < ktot) dst[(int64_t)k * nij + ij] = s_t[row][lane + 16 * u];
    }
}

// dependency rounds of the combine program of vn_build_tree: step t may run in round rnd[t] once both of its
// operands are final (leaves are ready in round 0)
__host__ __device__ inline int vn_build_rounds(int nleaf, const unsigned short *pl, const unsigned short *pr, unsigned char *rnd, unsigned char *ready)
{
    for (int i = 0; i < nleaf; ++i) ready[i] = 0;
    int nr = 0;
    for (int t = 0; t + 1 < nleaf; ++t) {
        const int a = ready[pl[t]], b = ready[pr[t]], r = (a > b ? a : b) + 1;
        rnd[t] = (unsigned char)r; ready[pl[t]] = (unsigned char)r;
        if (r > nr) nr = r;
    }
    return nr;
}

// lane i <- lane i + N of the same row of 16 lanes (DPP row_shl: data moves towards lower lanes); lanes whose source lies outside
// the row keep their own value -- the balanced tree never uses those
template <int N> __device__ __forceinline__ double vn_row_shl(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x100 | N, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x100 | N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// GLOBAL = false: the planes of the workgroup's KT levels are copied into LDS once and every evaluation runs from there
// (planes of up to ~9 000 points).  GLOBAL = true: planes too large for the LDS (128 x 128 and up -- ordinary DALES sizes)
// stay in the caller's transposed workspace, contiguous per level, and every evaluation streams them from L2 / the
// Infinity Cache with coalesced 64-B-per-leaf-group reads: one workgroup per level (KT = 1), i.e. n_cols x ktot workgroups
// and no strided [ij][k] access.  Same sums, same order, same bits as the LDS form.
// RCACHE (LDS form only, planes of <= 8192 points with one leaf per 8-lane group -- 64 x 64 and smaller): every lane keeps its
// share of the noise plane R in registers (32 more VGPRs; an instantiation of its own, so that the general form keeps its
// registers and schedule: with the cache compiled into it the 92 x 92 planes ran 12-18 % slower, profiles/r05_k6_stamps.log).
template <bool GLOBAL, bool RCACHE = false> __global__ __launch_bounds__(VN2_THREADS, RCACHE ? 2 : 4) void k_vnudge_solve(const Vn2P q)
{
    const VnP &p = q.p;
    extern __shared__ __align__(16) unsigned char vn2_smem[];
    __shared__ Vn2Tables s_tab;
    __shared__ int s_flag;
    unsigned short (&s_lo)[2][VN_MAXLEAF] = s_tab.lo, (&s_n)[2][VN_MAXLEAF] = s_tab.n, (&s_pl)[2][VN_MAXLEAF] = s_tab.pl, (&s_pr)[2][VN_MAXLEAF] = s_tab.pr;
    unsigned char (&s_rnd)[2][VN_MAXLEAF] = s_tab.rnd;
    int (&s_nleaf)[2] = s_tab.nleaf, (&s_nround)[2] = s_tab.nround;
    __shared__ double s_x[16];
    __shared__ int s_mode[16];

    // ---- which (column, tile): the tiles that share 128-B lines of the [ij][k] rows run on ONE XCD (blockIdx % 8) ----
    const unsigned b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
    const int tig = (int)(slot % (unsigned)q.tg);
    const int64_t group = (int64_t)(slot / (unsigned)q.tg) * 8 + xcd;
    if (group >= q.groups) return;
    const int64_t col = group / q.gpc;
    const int tile = (int)(group % q.gpc) * q.tg + tig;
    if (tile >= q.tiles) return;

    const int KT = q.kt, nij = p.nij, ktot = p.ktot, tid = threadIdx.x;
    const int NT = blockDim.x;                               // 512, or 256 where two workgroups share a CU
    const int TL = NT >> q.log2_kt;                          // threads per level
    const int kl = tid / TL, tl = tid - kl * TL;              // my level of the tile, my index inside its group
    const int grp = tl >> 3, acc = tl & 7, ngrp = TL >> 3;
    const int k0 = tile * KT, k = k0 + kl;
    const bool valid = k < ktot;
    const bool own = tl == 0;
    const int npl = vn2_plane(nij);                                            // skewed plane length
    double *const s_qt = reinterpret_cast<double *>(vn2_smem);               // [KT][npl]   (GLOBAL: no planes in LDS)
    double *const s_qs = s_qt + (GLOBAL ? 0 : (size_t)KT * npl);                // [KT][npl]
    double *const s_leaf = s_qs + (GLOBAL ? 0 : (size_t)KT * npl);              // [KT][nleaf_max]
    double *const s_part = s_leaf + (size_t)KT * q.nleaf_max;                  // [VN2_THREADS] argmax values
    int *const s_parti = reinterpret_cast<int *>(s_part + VN2_THREADS);        // [VN2_THREADS] argmax indices
    double *const my_leaf = s_leaf + (size_t)kl * q.nleaf_max;
    const int kq_ = (k < ktot) ? k : ktot - 1;
    // this level's planes: LDS copies (skewed, vn2_pos) or the contiguous planes of the transposed workspace
    const double *const my_qt = GLOBAL ? q.work + ((col * 2 + 0) * (int64_t)ktot + kq_) * nij : s_qt + (size_t)kl * npl;
    const double *const my_qs = GLOBAL ? q.work + ((col * 2 + 1) * (int64_t)ktot + kq_) * nij : s_qs + (size_t)kl * npl;
    auto POS = [](int e) { return GLOBAL ? e : vn2_pos(e); };

    {   // the host-built tree tables: argument block -> LDS, one 4-byte word per lane
        static_assert(sizeof(Vn2Tables) % 4 == 0, "Vn2Tables is copied by words");
        const unsigned __attribute__((address_space(4))) *const src = (const unsigned __attribute__((address_space(4))) *)(
            (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Vn2P, tab));
        unsigned *const dstw = reinterpret_cast<unsigned *>(&s_tab);
        for (int i = threadIdx.x; i < (int)(sizeof(Vn2Tables) / 4); i += blockDim.x) dstw[i] = src[i];
    }

    // ---- planes -> LDS ---------------------------------------------------------------------------------------------------
    if constexpr (GLOBAL) {
    } else if (q.work) {                            // transposed workspace: every level's group streams its own planes
        const int kq = valid ? k : ktot - 1;
        const double *const wq = q.work + ((col * 2 + 0) * (int64_t)ktot + kq) * nij;
        const double *const ws = q.work + ((col * 2 + 1) * (int64_t)ktot + kq) * nij;
        double *const dq = s_qt + (size_t)kl * npl, *const ds = s_qs + (size_t)kl * npl;
#pragma unroll 4
        for (int ij = tl; ij < nij; ij += TL) {
            dq[vn2_pos(ij)] = wq[ij];
            ds[vn2_pos(ij)] = ws[ij];
        }
    } else {                                        // [ij][k] order: element e = (ij, level of the tile), levels fastest
        const int64_t base = col * (int64_t)nij * ktot;
        const int total = nij << q.log2_kt;
#pragma unroll 8
        for (int e = tid; e < total; e += NT) {
            const int ij = e >> q.log2_kt, l = e & (KT - 1);
            const int kq = (k0 + l) < ktot ? (k0 + l) : ktot - 1;
            const int64_t g = base + (int64_t)ij * ktot + kq;
            s_qt[(size_t)l * npl + vn2_pos(ij)] = p.qt[g];
            s_qs[(size_t)l * npl + vn2_pos(ij)] = p.qsat[g];
        }
    }
    const int kk = valid ? k : ktot - 1;
    const int64_t lev = col * ktot + kk;
    const double *const R = p.R + col * (int64_t)nij;
    const double qt_av = p.qt_av[lev], ql_ref = p.ql_ref[lev], ql_av = p.ql_av[lev];

    // ---- per-level state (lane 0 of each level's group) ----------------------------------------------------------------
    int stage = VS_DONE, st = VN_NONE, err = 0, apply = 0;
    bool touched = false, want_argmax = false;
    double beta = 1.0, a = 0.0, f_lo = 0.0;
    VnBrent br = {};
    if (own) {
        if (valid) {
            if (ql_ref > SPC_MUT(20, 1e-6, 1e-9)) { stage = VS_M0; touched = true; }  // spcpl.py:665
            else if (ql_av > ql_ref) { want_argmax = true; touched = true; }         // spcpl.py:679
        }
        s_mode[kl] = want_argmax ? 3 : 0;
    }
    if (tid == 0) s_flag = 0;
    __syncthreads();
    if (own && want_argmax) atomicOr(&s_flag, 1);
    __syncthreads();

    // ---- "barely unsaturated" branch (spcpl.py:679-695): numpy.argmax(qt - qsat), first maximum, a NaN wins -------------
    if (s_flag) {
        const int seg = (nij + TL - 1) / TL, lo = tl * seg, hi = (lo + seg) < nij ? (lo + seg) : nij;
        double bv = 0.0;
        int bi = -1;
        if (s_mode[kl] == 3 && lo < hi) {
            bi = lo; bv = my_qt[POS(lo)] - my_qs[POS(lo)];
            for (int ij = lo + 1; ij < hi && !(bv != bv); ++ij) {
                const double v = my_qt[POS(ij)] - my_qs[POS(ij)];
                if (v > bv || v != v) { bv = v; bi = ij; }
            }
        }
        s_part[tid] = bv; s_parti[tid] = bi;
        __syncthreads();
        if (own && want_argmax) {
            double best = s_part[tid];
            int idx = s_parti[tid];
            for (int w = 1; w < TL && !(best != best); ++w) {
                const int qi = s_parti[tid + w];
                const double v = s_part[tid + w];
                if (qi >= 0 && (v > best || v != v)) { best = v; idx = qi; }
            }
            beta = (my_qs[POS(idx)] - qt_av) / (my_qt[POS(idx)] - qt_av);          // spcpl.py:683
            if (beta < 0) beta = 1.0;                                                // spcpl.py:692-695
            st = VN_UNSAT;
        }
        __syncthreads();
    }

    auto after_beta = [&]() {                                                        // spcpl.py:703-725
        if (beta >= 5.0) {
            if (ql_ref > ql_av) { stage = VS_A0; }
            else { st |= VN_ADD_SKIPPED; beta = 1.0; stage = VS_DONE; }
        } else {
            apply = 1; stage = VS_DONE;
        }
    };
    if (own && want_argmax) after_beta();

    // ---- root-finding rounds: one evaluation per level and round, every evaluation from LDS ------------------------------
    // Two workgroup barriers per round: the level's first wave (CW lanes) combines the leaf sums in numpy's tree order by
    // dependency rounds -- LDS operations of one wave execute in order, so no barrier between them -- and its lane 0 runs
    // the level's state machine on the result.
    const int CW = TL < 64 ? TL : 64;
    volatile double *const vleaf = my_leaf;
    // this lane's steps of the combine program are the same in every evaluation round: read them from the LDS tables ONCE
    // (up to two steps per lane and chunk shape: trees of <= 2 CW + 1 leaves, i.e. every plane the LDS path takes and the
    // 128-leaf chunks of the streamed path); taller trees walk the tables
    int tq_rnd[2][2] = {{0, 0}, {0, 0}}, tq_l[2][2] = {{0, 0}, {0, 0}}, tq_r[2][2] = {{0, 0}, {0, 0}};
    bool tq_fast[2];
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
        const int nl = s_nleaf[sh];
        tq_fast[sh] = nl - 1 <= 2 * CW;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int t = tl + qq * CW;
            if (tl < CW && t + 1 < nl) { tq_rnd[sh][qq] = s_rnd[sh][t]; tq_l[sh][qq] = s_pl[sh][t]; tq_r[sh][qq] = s_pr[sh][t]; }
        }
    }
    // Diagnostic build only (-DSPC_STAMPS, tools/stamps_k6.py): thread 0 sums, over all evaluation rounds, the shader-clock
    // time it spends in each part of a round (LDS / memory counters drained at every stamp) and leaves the sums in g_stamps.
#ifdef SPC_STAMPS
    unsigned long long k6_acc[6] = {0, 0, 0, 0, 0, 0}, k6_rounds = 0, k6_t = 0, k6_m2 = 0;
    const unsigned long long k6_w0 = wall_clock64();
#define K6_STAMP(i)                                                              \
    do {                                                                         \
        if (tid == 0 && g_stamps) {                                              \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          \
            const unsigned long long now_ = __builtin_readcyclecounter();        \
            if ((i) >= 0) k6_acc[(i) < 0 ? 0 : (i)] += now_ - k6_t;              \
            k6_t = now_;                                                         \
        }                                                                        \
    } while (0)
#else
#define K6_STAMP(i) do { } while (0)
#endif
    // The additive search (mode 2, spcpl.py:653-656) evaluates sum(max(qt + a R - qsat, 0)): R is the same in every round, and
    // with one leaf per 8-lane group a lane reads the SAME <= 16 elements of it each time.  The round stamps (tools/stamps_k6.py,
    // profiles/r05_k6_stamps.log) put 3.0 us of a 5.1-us mode-2 round into these reads -- 16 dependent-latency trips to L2 per
    // lane and round, where a mode-1 round (planes in LDS only) sums its leaves in 0.43 us -- so the lane's share of R is
    // loaded ONCE into registers.  Same terms, same order, same bits.
    constexpr int RC = 16;                                          // a leaf has <= 128 elements: <= 16 per lane
    double Rreg[RC];
    bool r_cached = false;
    if constexpr (RCACHE) {
        r_cached = nij <= 8192 && s_nleaf[1] <= ngrp;              // (the host launches this instantiation only then)
        if (r_cached) {
            const bool have = grp < s_nleaf[1];
            const int lo = have ? (int)s_lo[1][grp] : 0, n = have ? (int)s_n[1][grp] : 0;
#pragma unroll
            for (int i = 0; i < RC; ++i) Rreg[i] = (n >= 8 && i < (n >> 3)) ? R[lo + 8 * i + acc] : 0.0;
        }
    }
    for (;;) {
        K6_STAMP(-1);
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
            s_x[kl] = x; s_mode[kl] = mode;
        }
        __syncthreads();
        K6_STAMP(0);                                                 // hand-over of x / mode + barrier
        int any = 0;
        for (int l = 0; l < KT; ++l) any |= s_mode[l];
        if (!any) break;
        const double x = s_x[kl];
        const int mode = s_mode[kl];
        double total = 0.0;
        for (int c0 = 0; c0 < nij; c0 += 8192) {                    // ndarray.sum(): 0.0 + chunk sums
            const int cn = (nij - c0) < 8192 ? (nij - c0) : 8192;
            const int shape = (c0 + cn < nij) ? 0 : 1;
            const int nleaf = s_nleaf[shape];
            // one leaf per 8-lane group: lane j carries numpy's accumulator r[j]; `term` reads the planes from LDS
            auto leaves = [&](auto term) {
                for (int li = grp; li < nleaf; li += ngrp) {        // uniform inside every 8-lane group
                    const int lo = (int)s_lo[shape][li], n = (int)s_n[shape][li];
                    double res;
                    if (n < 8) {
                        res = 0.0;
                        if (acc == 0)
                            for (int i = 0; i < n; ++i) res += term(lo + i);
                    } else {
                        const int cnt = n >> 3, n8 = cnt << 3;
                        double r = term(lo + acc);
                        if constexpr (GLOBAL) {
#pragma unroll 15
                            for (int i = 1; i < cnt; ++i) r += term(lo + 8 * i + acc);
                        } else {
#pragma unroll 4
                            for (int i = 1; i < cnt; ++i) r += term(lo + 8 * i + acc);
                        }
                        r = r + __shfl_down(r, 1, 8);               // lanes 0,2,4,6: r0+r1, r2+r3, r4+r5, r6+r7
                        r = r + __shfl_down(r, 2, 8);               // lanes 0,4: (r0+r1)+(r2+r3), (r4+r5)+(r6+r7)
                        r = r + __shfl_down(r, 4, 8);               // lane 0: the leaf's 8-accumulator sum
                        res = r;
                        if (acc == 0)
                            for (int i = n8; i < n; ++i) res += term(lo + i);
                    }
                    if (acc == 0) my_leaf[li] = res;
                }
            };
            // RCACHE, a FULL leaf (128 elements, 16 per lane -- every leaf of a 64 x 64 plane): all 32 LDS reads of the lane are
            // issued before the first term is formed (the scheduler otherwise keeps 4 in flight and waits 8 times: 1.37 us per
            // mode-2 round against 0.5 us, tools/stamps_k6.py); the terms and their order are those of `leaves`
            // (a leaf that starts on a multiple of 128 lies in ONE skew block of the plane: element lo + 8 i + acc sits 8 i doubles
            //  behind the lane's first one, an immediate offset of the read)
            const bool fast_leaf = RCACHE && r_cached && grp < nleaf && (int)s_n[shape][grp] == 8 * RC && ((int)s_lo[shape][grp] & 127) == 0;
            auto full_leaf = [&](auto term) {
                const int e0 = POS((int)s_lo[shape][grp] + acc);
                const double *const pa = my_qt + e0, *const pb = my_qs + e0;
                double qa[RC], qb[RC];
#pragma unroll
                for (int i = 0; i < RC; ++i) {
                    qa[i] = pa[8 * i];
                    qb[i] = pb[8 * i];
                }
                __builtin_amdgcn_sched_barrier(0);
                double r = term(0, qa[0], qb[0]);
#pragma unroll
                for (int i = 1; i < RC; ++i) r += term(i, qa[i], qb[i]);
                r = r + __shfl_down(r, 1, 8);
                r = r + __shfl_down(r, 2, 8);
                r = r + __shfl_down(r, 4, 8);
                if (acc == 0) my_leaf[grp] = r;
            };
            if (fast_leaf && mode == 1)
                full_leaf([&](int, double a_, double b_) {
                    const double t = ((x * (a_ - qt_av)) + qt_av) - b_;
                    return (t >= 0.0 || t != t) ? t : 0.0;
                });
            else if (fast_leaf && mode == 2)
                full_leaf([&](int i, double a_, double b_) {
                    const double t = (a_ + (x * Rreg[i])) - b_;
                    return (t >= 0.0 || t != t) ? t : 0.0;
                });
            else if (mode == 1)
                leaves([&](int ij) {
                    const int e = POS(c0 + ij);
                    const double t = ((x * (my_qt[e] - qt_av)) + qt_av) - my_qs[e];
                    return (t >= 0.0 || t != t) ? t : 0.0;                           // numpy.maximum(t, 0)
                });
            else if (RCACHE && mode == 2 && r_cached) {                // (one chunk, one leaf per group: li = grp)
                if (grp < nleaf) {
                    const int lo = (int)s_lo[shape][grp], n = (int)s_n[shape][grp];
                    const auto term_g = [&](int ij) {
                        const int e = POS(ij);
                        const double t = (my_qt[e] + (x * R[ij])) - my_qs[e];
                        return (t >= 0.0 || t != t) ? t : 0.0;
                    };
                    double res;
                    if (n < 8) {
                        res = 0.0;
                        if (acc == 0)
                            for (int i = 0; i < n; ++i) res += term_g(lo + i);
                    } else {
                        const int cnt = n >> 3, n8 = cnt << 3;
                        const auto term_r = [&](int i) {
                            const int e = POS(lo + 8 * i + acc);
                            const double t = (my_qt[e] + (x * Rreg[i])) - my_qs[e];
                            return (t >= 0.0 || t != t) ? t : 0.0;
                        };
                        double r = term_r(0);                       // (a short leaf: the last one of a plane that is no multiple of 128)
#pragma unroll
                        for (int i = 1; i < RC; ++i)
                            if (i < cnt) r += term_r(i);
                        r = r + __shfl_down(r, 1, 8);
                        r = r + __shfl_down(r, 2, 8);
                        r = r + __shfl_down(r, 4, 8);
                        res = r;
                        if (acc == 0)
                            for (int i = n8; i < n; ++i) res += term_g(lo + i);
                    }
                    if (acc == 0) my_leaf[grp] = res;
                }
            } else if (mode == 2)
                leaves([&](int ij) {
                    const int e = POS(c0 + ij);
                    const double t = (my_qt[e] + (x * R[c0 + ij])) - my_qs[e];
                    return (t >= 0.0 || t != t) ? t : 0.0;
                });
#ifdef SPC_STAMPS
            if (mode == 2) { K6_STAMP(5); if (tid == 0) ++k6_m2; } else { K6_STAMP(1); }   // this wave's leaf sums, by mode
#endif
            __syncthreads();
            K6_STAMP(2);                                            // barrier: every wave's leaves are in LDS
            if (tl < CW && mode != 0 && s_tab.balanced[shape] && nleaf <= CW) {      // the balanced tree: in registers (vn_tree_balanced);
                double v = tl < nleaf ? my_leaf[tl] : 0.0;                            // lanes tl .. tl + nleaf - 1 are this level's
                // lane i + lane i + d: inside a row of 16 lanes by DPP row shifts (a register move), beyond by ds_bpermute
                if (nleaf > 1) v = v + vn_row_shl<1>(v);
                if (nleaf > 2) v = v + vn_row_shl<2>(v);
                if (nleaf > 4) v = v + vn_row_shl<4>(v);
                if (nleaf > 8) v = v + vn_row_shl<8>(v);
                for (int sft = 16; sft < nleaf; sft <<= 1) v = v + __shfl_down(v, sft);
                if (own) total += v;
            } else if (tl < CW && mode != 0) {                      // numpy's tree, one dependency round at a time
                const int nround = s_nround[shape];
                if (shape ? tq_fast[1] : tq_fast[0]) {
                    const int r0 = shape ? tq_rnd[1][0] : tq_rnd[0][0], r1 = shape ? tq_rnd[1][1] : tq_rnd[0][1];
                    const int l0 = shape ? tq_l[1][0] : tq_l[0][0], l1 = shape ? tq_l[1][1] : tq_l[0][1];
                    const int p0 = shape ? tq_r[1][0] : tq_r[0][0], p1 = shape ? tq_r[1][1] : tq_r[0][1];
                    for (int rd = 1; rd <= nround; ++rd) {
                        if (r0 == rd) vleaf[l0] = vleaf[l0] + vleaf[p0];
                        if (r1 == rd) vleaf[l1] = vleaf[l1] + vleaf[p1];
                        __builtin_amdgcn_wave_barrier();
                    }
                } else {
                    for (int rd = 1; rd <= nround; ++rd) {
                        for (int t = tl; t + 1 < nleaf; t += CW)
                            if (s_rnd[shape][t] == rd) {
                                const int l = s_pl[shape][t], r = s_pr[shape][t];
                                vleaf[l] = vleaf[l] + vleaf[r];
                            }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                if (own) total += vleaf[0];
            }
            if (c0 + 8192 < nij) __syncthreads();                   // the next chunk's leaves reuse the slots
        }
        K6_STAMP(3);                                                // tree combine (dependency rounds of one wave)
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
        K6_STAMP(4);                                                // f, one step of the level's brentq state machine
#ifdef SPC_STAMPS
        ++k6_rounds;
#endif
    }
#ifdef SPC_STAMPS
    if (tid == 0 && g_stamps) {
        unsigned long long *const o = g_stamps + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 5; ++i) o[i] = k6_acc[i];
        o[5] = k6_rounds; o[6] = wall_clock64() - k6_w0; o[7] = k6_acc[5] | (k6_m2 << 48);      // mode-2 leaf time | mode-2 rounds
    }
#endif
#undef K6_STAMP

    if (own && valid) {
        p.beta[lev] = beta;
        p.a_add[lev] = a;
        p.status[lev] = st | ((err & 1) ? VN_ERR_SIGN : 0) | ((err & 2) ? VN_ERR_CONV : 0) |
                        (apply == 1 ? VN2_APPLY_MULT : 0) | (apply == 2 ? VN2_APPLY_ADD : 0) | (touched ? VN2_TOUCHED : 0);
    }
}

// K6b: qt (and thl with constantT) of every touched level (spcpl.py:716-733).  Elementwise, no order constraint: as many
// workgroups as the data allows (a workgroup takes VU_ROWS (i, j) rows of one column; 32 lanes along k, the fastest
// index of the reference's [ij][k] layout, 8 rows at a time), levels that need nothing are neither read nor written.
constexpr int VU_ROWS = 64;
__global__ __launch_bounds__(256) void k_vnudge_update(const VnP p)
{
    extern __shared__ __align__(16) unsigned char vu_smem[];
    const int ktot = p.ktot, nij = p.nij, tid = threadIdx.x;
    double *const s_coef = reinterpret_cast<double *>(vu_smem);          // [ktot] beta - 1 or a
    double *const s_av = s_coef + ktot;                                    // [ktot] qt_av
    double *const s_tc = s_av + ktot;                                      // [ktot] -rlv / (cp exner(presf)), constantT
    int *const s_ap = reinterpret_cast<int *>(s_tc + ktot);               // [ktot] 0 nothing, 1 multiplicative, 2 additive, +4 thl
    __shared__ int s_any;
    const int64_t col = blockIdx.y;
    if (tid == 0) s_any = 0;
    __syncthreads();
    for (int k = tid; k < ktot; k += 256) {
        const int64_t lev = col * ktot + k;
        const int stv = p.status[lev];
        const int ap = (stv & VN2_APPLY_MULT) ? 1 : ((stv & VN2_APPLY_ADD) ? 2 : 0);
        const bool th = p.constantT && (stv & VN2_TOUCHED);
        s_coef[k] = ap == 1 ? p.beta[lev] - 1 : p.a_add[lev];
        s_av[k] = p.qt_av[lev];
        s_tc[k] = th ? SPC_MUT(19, K<double>::rlv, -K<double>::rlv) / (K<double>::cp * spc_pow(div_pref0(p.presf[lev]), K<double>::rd / K<double>::cp)) : 0.0;   // spcpl.py:731
        s_ap[k] = ap | (th ? 4 : 0);
        if (ap | (th ? 4 : 0)) s_any = 1;
    }
    __syncthreads();
    if (!s_any) return;
    const int kq = tid & 31, rq = tid >> 5;
    const int ij0 = blockIdx.x * VU_ROWS, ij1 = (ij0 + VU_ROWS) < nij ? (ij0 + VU_ROWS) : nij;
    const int64_t base = col * (int64_t)nij * ktot;
    const double *const R = p.R + col * (int64_t)nij;
    for (int k = kq; k < ktot; k += 32) {
        const int ap = s_ap[k];
        if (!ap) continue;
        const double coef = s_coef[k], qt_av = s_av[k], tc = s_tc[k];
#pragma unroll 4
        for (int ij = ij0 + rq; ij < ij1; ij += 8) {
            const int64_t g = base + (int64_t)ij * ktot + k;
            double v = p.qt[g];
            if ((ap & 3) == 1) { v = v + coef * SPC_MUT(18, v, (v - qt_av)); p.qt[g] = v; } // spcpl.py:724-725
            else if ((ap & 3) == 2) { v = SPC_MUT(21, v - coef * R[ij], v + coef * R[ij]); p.qt[g] = v; }   // spcpl.py:716-719
            if (ap & 4) {                                                                   // spcpl.py:726-733
                const double tt = v - p.qsat[g];
                const double ql_target = (tt >= 0.0 || tt != tt) ? tt : 0.0;
                p.thl[g] += tc * (ql_target - p.ql[g]);
            }
        }
    }
}

// K6c: qt.std(axis=(0, 1)) (spcpl.py:741): numpy reduces over (i, j) with k as the inner loop, i.e. plain SEQUENTIAL sums
// in C order (mean = sum/N, then sum((x - mean)^2)/N, sqrt).  The order binds the ADDS, not the loads: a workgroup owns
// 16 consecutive levels of one column (one 128-B line per (i, j) row).  Waves 1-4 (256 threads) only LOAD: tiles of ROWS
// rows (ROWS / 16 rows per thread in flight, unconditional loads at clamped indices) into an LDS ping-pong; wave 0 only
// ADDS: its first 16 lanes take tile t's rows IN ORDER from LDS -- reads software-pipelined 16 rows ahead of the dependent
// add chain -- while tile t + 1 sits in the other buffer and the loads of tile t + 2 are in flight.  (Round 2's kernel
// kept ONE 128-row tile in flight per workgroup, its summing wave also loaded, and it was bound by the load latency of
// each tile: 143 us for a 64 x 64 x 160 LES even when no level needed a nudge.)  ROWS = 512 (128 KiB of LDS, one
// workgroup per CU) for few workgroups, 256 for many.
// rows [0, nr) of one LDS tile added IN ORDER to `acc` (SQ: their squared deviations from `mean`): the reads of the next G
// rows are in flight while the current G are added.  Separate instantiations for the two passes keep the loop body free of
// selects and register copies.
template <bool SQ, int ROWS> __device__ __forceinline__ double vs_sum_tile(const double (*tl)[16], int lane, int nr, double mean, double acc)
{
    constexpr int G = 16;
    if (nr == ROWS) {
        // a full tile: the whole tile unrolled, group g + 1 read from LDS while group g is added (no loop-carried registers,
        // hence no copies between them); sched_barrier keeps the reads IN FRONT of the chain they overlap
        double v[2][G];
#pragma unroll
        for (int i = 0; i < G; ++i) v[0][i] = tl[i][lane];
#pragma unroll
        for (int g = 0; g < ROWS / G; ++g) {
            if (g + 1 < ROWS / G) {
#pragma unroll
                for (int i = 0; i < G; ++i) v[(g + 1) & 1][i] = tl[(g + 1) * G + i][lane];
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SQ) {
#pragma unroll
                for (int i = 0; i < G; ++i) { const double dlt = v[g & 1][i] - mean; v[g & 1][i] = dlt * dlt; }
            }
#pragma unroll
            for (int i = 0; i < G; ++i) acc += v[g & 1][i];
            __builtin_amdgcn_sched_barrier(0);
        }
        return acc;
    }
    for (int r0 = 0; r0 < nr; ++r0) {                         // the ragged last tile
        const double x = tl[r0][lane];
        if constexpr (SQ) { const double dlt = x - mean; acc += dlt * dlt; }
        else acc += x;
    }
    return acc;
}

constexpr int VS_THREADS = 320;
template <int ROWS> __global__ __launch_bounds__(VS_THREADS) void k_vnudge_std(const VnP p)
{
    extern __shared__ __align__(16) unsigned char vs_smem[];
    double (*const s_tile)[ROWS][16] = reinterpret_cast<double (*)[ROWS][16]>(vs_smem);      // [2][ROWS][16]
    __shared__ double s_mean[16];
    constexpr int U = ROWS / 16;                         // rows per loader thread and tile, all in flight together
    const int tid = threadIdx.x;
    const bool adder = tid < 64;                         // wave 0; its lanes 0..15 carry one level each
    const int lt = adder ? tid : tid - 64;
    const int lane = lt & 15, row = lt >> 4;
    const int64_t col = blockIdx.y;
    const int k = blockIdx.x * 16 + lane, nij = p.nij;
    const bool valid = k < p.ktot;
    const int kk = valid ? k : p.ktot - 1;               // in-bounds addresses for the lanes past the last level
    const int64_t lev = col * p.ktot + kk, ks = p.ktot, base = col * (int64_t)nij * ks + kk;
    const double *const qt = p.qt + base;
    const int ntile = (nij + ROWS - 1) / ROWS;
    double w[U];
    const int dbg = p.pad;                               // experiments only (SPC_VN_STD_DEBUG): 1 no adds, 2 no loads
    auto load_tile = [&](int t) {                        // rows past the plane's end re-read its last row: never added
        if (dbg & 2) return;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ij = t * ROWS + row + 16 * u;
            w[u] = qt[(int64_t)(ij < nij ? ij : nij - 1) * ks];
        }
    };
    double mean = 0.0, acc = 0.0;
    int pass = 0;
    auto store_tile = [&](int t) {                       // pass 1: the LOADER waves square the deviations, the adder only adds
        if (pass == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) s_tile[t & 1][row + 16 * u][lane] = w[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) { const double dlt = w[u] - mean; s_tile[t & 1][row + 16 * u][lane] = dlt * dlt; }
        }
    };
    for (pass = 0; pass < 2; ++pass) {                   // pass 0: sum -> mean; pass 1: sum of squared deviations
        acc = 0.0;
        if (!adder) {
            load_tile(0);
            store_tile(0);
            if (ntile > 1) load_tile(1);
        }
        __syncthreads();
        for (int t = 0; t < ntile; ++t) {
            if (adder) {
                if (row == 0 && !(dbg & 1)) {
                    const int nr = (nij - t * ROWS) < ROWS ? (nij - t * ROWS) : ROWS;
                    acc = vs_sum_tile<false, ROWS>(s_tile[t & 1], lane, nr, mean, acc);
                }
            } else if (t + 1 < ntile) {
                store_tile(t + 1);                        // the other buffer: its last reader (tile t - 1) finished before the last barrier
                if (t + 2 < ntile) load_tile(t + 2);
            }
            __syncthreads();
        }
        if (pass == 0) {
            if (adder && row == 0) s_mean[lane] = acc / (double)nij;
            __syncthreads();
            mean = s_mean[lane];
        }
    }
    if (adder && row == 0 && valid) {
        p.qt_std[lev] = sqrt(acc / (double)nij);
        p.status[lev] = p.status[lev] & ~VN2_INTERNAL;
    }
}
