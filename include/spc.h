/*
 * spc.h -- C ABI of the MI355X-native batched superparameterization coupling step.
 *
 * The reference (CloudResolvingClimateModeling/sp-coupler) is pure Python and defines no FFI; its
 * boundary for this path is the Python call contract of splib/spcpl.py.  Each entry point below
 * replaces the *arithmetic* of one group of reference functions for ALL SP columns at once; the
 * Python host mirror (sp_coupler_amd/spcpl.py) keeps the reference's function names on top of it.
 *
 *   spc_forward_*        <- spcpl.convert_profiles        splib/spcpl.py:171-246
 *                           spcpl.set_les_forcings        splib/spcpl.py:299-385 (arithmetic 316-333)
 *                           spcpl.convert_surface_fluxes  splib/spcpl.py:136-167 (optional)
 *                           index map of get_les_profiles splib/spcpl.py:761-764 (optional, fused)
 *                           sputils.interp / iexner       splib/sputils.py:82-86, 33-34
 *   spc_cloud_indices_*  <- spcpl.get_cloud_fraction      splib/spcpl.py:22-29 (line 26)
 *                           spcpl.get_les_profiles        splib/spcpl.py:761-764
 *                           sputils.searchsorted          splib/sputils.py:88-91
 *   spc_backward_*       <- spcpl.set_gcm_tendencies      splib/spcpl.py:388-555 (arithmetic 402-533)
 *                           sputils.interp_c / integral   splib/sputils.py:94-189 (conservative=1)
 *   spc_surface_fluxes_* <- spcpl.convert_surface_fluxes  splib/spcpl.py:136-167 (columns without LES)
 *   spc_variability_nudge_f64 <- spcpl.variability_nudge  splib/spcpl.py:613-744 (qt_forcing == 'variance')
 *   spc_diagnostics_*    <- spifs.nc diagnostics          splib/spcpl.py:176,214-215,408-409;
 *                           spcpl.output_column_conversion splib/spcpl.py:251-267
 *   spc_exner_* / spc_interp_* / spc_searchsorted_* / spc_interp_c_* / spc_rms_*
 *                        <- the helpers of splib/sputils.py on their own (exner, iexner :28-34; interp :82-86;
 *                           searchsorted :88-91; integral, interp_c, interp_rho :94-197; rms :23-24), batched over rows
 *
 * Conventions
 *   - All data pointers are DEVICE pointers (HBM) owned by the caller; the library never allocates,
 *     frees or copies them and keeps no global state except the last-error string (thread local).
 *   - Element type is double for *_f64 and float for *_f32; index outputs are int32_t.
 *   - Arrays are row-major [n_cols x n_lev] with an explicit element pitch between columns.
 *       GCM full-level arrays   [n_cols x nG]     pitchG  (index 0 = model top, nG-1 = lowest level)
 *       GCM half-level arrays   [n_cols x (nG+1)] pitchGh (index nG = surface)
 *       LES arrays              [n_cols x nL]     pitchL  (index 0 = lowest level, ascending)
 *       per-column scalars      [n_cols]
 *     The LES grids zf / zh are [nL] shared by all columns when les_grid_shared != 0, else
 *     [n_cols x nL] with pitchL.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls only enqueue work.
 *   - Optional pointers may be NULL; the corresponding work/outputs are skipped.
 *   - Every function returns 0 on success or a negative spc_status; spc_last_error() gives the text.
 *     No C++ exception crosses the ABI.
 */
#ifndef SPC_H
#define SPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPC_ABI_VERSION 4

typedef enum spc_status {
    SPC_OK = 0,
    SPC_ERR_INVALID_ARGUMENT = -1, /* NULL required pointer, bad dims or pitches              */
    SPC_ERR_UNSUPPORTED = -2,      /* level counts exceed what the kernels' LDS staging holds */
    SPC_ERR_LAUNCH = -3,           /* hipLaunchKernel / runtime error (see spc_last_error)    */
    SPC_ERR_NO_DEVICE = -4         /* no HIP device available                                 */
} spc_status;

typedef struct spc_dims {
    int64_t n_cols;          /* number of SP columns in the batch (0 is allowed: no-op)          */
    int32_t nG;              /* GCM full levels (19 / 91 / 137 ...)                              */
    int32_t nL;              /* LES levels (160 / 512 ...)                                       */
    int64_t pitchG;          /* elements between columns of [n_cols x nG] arrays,   >= nG        */
    int64_t pitchGh;         /* elements between columns of [n_cols x nG+1] arrays, >= nG+1      */
    int64_t pitchL;          /* elements between columns of [n_cols x nL] arrays,   >= nL        */
    int32_t les_grid_shared; /* 1: zf/zh are [nL]; 0: zf/zh are [n_cols x nL]                    */
    int32_t cols_per_block;  /* tuning: columns per 256-thread workgroup (0 = library heuristic) */
} spc_dims;

/* ---- forward: GCM state -> LES profiles + nudging forcings (kernel K1, K2 fused) ------------- */
typedef struct spc_forward_args {
    /* GCM inputs, spcpl.gather_gcm_data var list (spcpl.py:32). `A` is not used by this pass. */
    const void *U, *V, *T, *SH, *QL, *QI, *Pf; /* [n_cols x nG]                                  */
    const void *Ph;                            /* [n_cols x nG+1]; only Ph[nG] (surface) is read */
    const void *Zgfull;                        /* [n_cols x nG]   geopotential, full levels      */
    const void *Zghalf;                        /* [n_cols x nG+1] geopotential, half levels      */
    /* LES grid and slab means (profile[...] of spcpl.py:310-315)                               */
    const void *zf;                            /* LES full-level heights (les.zf_cache)          */
    const void *zh;                            /* LES half-level heights; only needed for `idx`  */
    const void *u_d, *v_d, *thl_d, *qt_d, *ql_d; /* [n_cols x nL]                                */
    const void *ps_d;                          /* [n_cols]                                       */
    const void *rain, *rain_last;              /* [n_cols], optional (both or neither)           */
    double factor;                             /* les_forcing_factor                             */
    double dt;                                 /* dt_gcm in seconds                              */
    /* required outputs (what the 7 setters of spcpl.py:341-347 receive)                        */
    void *f_u, *f_v, *f_thl, *f_qt, *f_ql, *ql_ref; /* [n_cols x nL]                            */
    void *f_ps;                                /* [n_cols]                                       */
    /* optional outputs                                                                         */
    void *u, *v, *thl, *qt;                    /* [n_cols x nL] interpolated profiles (return value
                                                  of convert_profiles; ql == ql_ref)            */
    void *ps;                                  /* [n_cols] Ph[nG]                                */
    void *Zf;                                  /* [n_cols x nG]   les.gcm_Zf (spcpl.py:200)      */
    void *Zh;                                  /* [n_cols x nG+1] les.gcm_Zh (spcpl.py:201)      */
    void *rainrate;                            /* [n_cols] (rain-rain_last)/dt (spcpl.py:325)    */
    int32_t *idx;                              /* [n_cols x nG] pitchG: cloud-fraction level map,
                                                  searchsorted(zh,Zh,'right')[:-1][::-1]         */
    /* optional surface coupling (cplsurf): inputs [n_cols], outputs [n_cols]                   */
    const void *Z0M, *Z0H, *QLflux, *QIflux, *SHflux, *TSflux;
    void *z0m, *z0h, *wthl, *wqt;
} spc_forward_args;

int spc_forward_f64(const spc_dims *dims, const spc_forward_args *args, void *stream);
int spc_forward_f32(const spc_dims *dims, const spc_forward_args *args, void *stream);

/* ---- index map only (kernel K2 standalone) --------------------------------------------------- */
/* Zh: [n_cols x nG+1] heights of GCM half levels (descending, Zh[nG] = 0), as cached by forward.
 * idx[c][m] = searchsorted(zh, Zh[c], side='right')[nG-1-m], m = 0..nG-1; values in 0..nL.       */
int spc_cloud_indices_f64(const spc_dims *dims, const void *zh, const void *Zh, int32_t *idx, void *stream);
int spc_cloud_indices_f32(const spc_dims *dims, const void *zh, const void *Zh, int32_t *idx, void *stream);

/* ---- backward: LES slab means -> GCM tendencies (kernel K3; K4 when conservative) ------------ */
typedef struct spc_backward_args {
    const void *T, *SH, *QL, *QI, *U, *V, *A; /* [n_cols x nG] GCM state                        */
    const void *Zf;                           /* [n_cols x nG] les.gcm_Zf; may be NULL when
                                                 Zgfull and Zghalf are given (recomputed)       */
    const void *Zgfull, *Zghalf;              /* optional, see Zf                               */
    const void *zf;                           /* LES full-level heights                         */
    const void *t_d, *qt_d, *ql_d, *ql_ice_d, *u_d, *v_d; /* [n_cols x nL] profile[T,QT,QL,QL_ice,U,V] */
    const void *A_prof;                       /* [n_cols x nG] pitchG: profile["A"] in the order
                                                 get_cloudfraction(indices) returns it (ascending
                                                 height); reversed in-kernel (spcpl.py:404)     */
    /* conservative coarsening (sputils.interp_c); required only when conservative != 0        */
    const void *zh;                           /* LES half-level heights                         */
    const void *Zh;                           /* [n_cols x nG+1] or NULL with Zghalf given      */
    const void *rhobf_d;                      /* [n_cols x nL] profile["Rhobf"]                 */
    int32_t conservative;
    int32_t reserved;
    double factor;                            /* gcm_forcing_factor                             */
    double dt;                                /* dt_gcm in seconds                              */
    void *f_T, *f_SH, *f_QL, *f_QI, *f_U, *f_V, *f_A; /* [n_cols x nG] outputs                  */
    int32_t *start_index;                     /* [n_cols] optional                              */
} spc_backward_args;

int spc_backward_f64(const spc_dims *dims, const spc_backward_args *args, void *stream);
int spc_backward_f32(const spc_dims *dims, const spc_backward_args *args, void *stream);

/* ---- diagnostics for spifs.nc (kernel K5) ----------------------------------------------------- */
typedef struct spc_diagnostics_args {
    const void *T, *SH, *QL, *QI, *Pf;        /* [n_cols x nG]                                  */
    const void *Zgfull, *Zghalf;              /* geopotential                                   */
    const void *zf;                           /* LES heights; with thl_d, ql_d for `t`          */
    const void *thl_d, *ql_d, *ql_ice_d;      /* [n_cols x nL] optional                         */
    void *Tv, *THL, *QT;                      /* [n_cols x nG] optional outputs (spcpl.py:176,214,215) */
    void *Zf;                                 /* [n_cols x nG]   optional                       */
    void *Zh;                                 /* [n_cols x nG+1] optional                       */
    void *pf, *t, *ql_water;                  /* [n_cols x nL] optional (spcpl.py:408,409,402)  */
} spc_diagnostics_args;

int spc_diagnostics_f64(const spc_dims *dims, const spc_diagnostics_args *args, void *stream);
int spc_diagnostics_f32(const spc_dims *dims, const spc_diagnostics_args *args, void *stream);

/* ---- surface fluxes for columns without an LES (extra output columns) --------------------------- */
/* spcpl.convert_surface_fluxes (splib/spcpl.py:136-167) on per-column scalars [n]:
 *   rho = Ph_s/(rd*T_s); wqt = -(QLflux+QIflux+SHflux)/rho; wthl = -TSflux*iexner(Ph_s)/(cp*rho)
 * Ph_s = Phalf[:, nG], T_s = T[:, nG-1].  (z0m, z0h are passed through by the caller.)  For SP columns
 * the same arithmetic is fused into spc_forward_* (wthl / wqt outputs).                               */
int spc_surface_fluxes_f64(int64_t n, const void *Ph_s, const void *T_s, const void *QLflux, const void *QIflux,
                           const void *SHflux, const void *TSflux, void *wthl, void *wqt, void *stream);
int spc_surface_fluxes_f32(int64_t n, const void *Ph_s, const void *T_s, const void *QLflux, const void *QIflux,
                           const void *SHflux, const void *TSflux, void *wthl, void *wqt, void *stream);

/* ---- variability nudge (qt_forcing == 'variance'): spcpl.variability_nudge, splib/spcpl.py:613-744 ------------ */
/* 3-D LES fields in the reference's layout [n_cols][itot][jtot][ktot] (k fastest), float64.  For every level the
 * kernel finds beta (multiplicative, brentq on [0,5]) or a (additive noise a*R, brentq on [0,5]) such that the plane
 * mean of max(qt' - qsat, 0) equals ql_ref[k], updates qt in place (and thl with constantT), and returns beta, a,
 * qt.std(axis=(0,1)) and a status word per (column, level):
 *   bit 0 multiplicative root found, 1 "barely unsaturated" branch (spcpl.py:679-695), 2 additive root found,
 *   3 additive branch skipped (ql_ref <= ql_av), 4 no bracket -> beta_max; bit 8: brentq sign error (the reference
 *   raises ValueError there), bit 9: no convergence in 100 iterations (RuntimeError).
 * R: [n_cols][itot*jtot] the zero-mean Gaussian field of spcpl.py:620-621, drawn by the caller.  At most 32 767
 * columns per call.  With the workspace (`work`) planes of up to ~9 000 points (KT levels x itot*jtot x 16 B <= 150 KiB of
 * LDS, e.g. 64 x 64, 90 x 90) are solved from the CU's LDS and larger ones (96 x 96, 128 x 128, 256 x 256 ...) by one
 * workgroup per level streaming its contiguous transposed planes; the update and qt.std are two further launches.
 * Without the workspace planes that fit the LDS are loaded strided (slower) and larger ones are refused
 * (SPC_ERR_INVALID_ARGUMENT); results are bit-identical on every path.                                            */
typedef struct spc_vnudge_args {
    int64_t n_cols;
    int32_t itot, jtot, ktot;
    int32_t constantT;                 /* variability_nudge_constant_T */
    void *qt;                          /* [n][itot][jtot][ktot] in/out: les.get_field("QT") -> les.fields.QT  */
    const void *qsat;                  /* [n][itot][jtot][ktot] les.get_field("Qsat")                         */
    void *thl;                         /* in/out, constantT only: les.get_field("THL")                        */
    const void *ql;                    /* constantT only: les.get_field("QL")                                 */
    const void *R;                     /* [n][itot*jtot]                                                      */
    const void *ql_av, *qt_av, *presf; /* [n][ktot] les.get_profile("QL"/"QT"), les.get_presf()               */
    const void *ql_ref;                /* [n][ktot] les.ql_ref (K1's ql_ref output)                           */
    void *beta, *a_add, *qt_std;       /* [n][ktot] outputs                                                   */
    int32_t *status;                   /* [n][ktot] output                                                    */
    void *work;                        /* device scratch of work_bytes >= spc_vnudge_workspace_bytes(): qt and qsat as  */
    int64_t work_bytes;                /* contiguous planes; optional (NULL) only for planes that fit the LDS           */
} spc_vnudge_args;

int spc_variability_nudge_f64(const spc_vnudge_args *args, void *stream);

/* ---- the helpers of splib/sputils.py as standalone batched operators (kernel family K7) ---------------------- */
/* The fused kernels above contain this arithmetic already; these entry points serve callers that use a helper on its
 * own (sp_coupler_amd/sputils.py keeps the reference's names on top of them).  A "row" is one independent 1-D problem
 * (one column); arrays are [n_rows x n] with an element pitch between rows; where stated a pitch of 0 means ONE row
 * shared by all rows (the LES grid).  Results are bit-identical to NumPy for interp / searchsorted / integral /
 * interp_c / interp_rho / rms; exner / iexner agree with numpy.power to <= 2 ulp.  Row pitches must stay below 2^24
 * elements (SPC_ERR_UNSUPPORTED otherwise): the kernels address a slab of rows with 24-bit multiplies.                */

/* sputils.exner (inverse == 0) / iexner (inverse != 0), splib/sputils.py:28-34: out[i] = (p[i]/pref0)**(+-rd/cp)   */
int spc_exner_f64(int64_t n, const void *p, void *out, int32_t inverse, void *stream);
int spc_exner_f32(int64_t n, const void *p, void *out, int32_t inverse, void *stream);

/* sputils.interp, splib/sputils.py:82-86 == numpy.interp(x, xp, fp) per row (end-clamped, exact-hit shortcut, NaN
 * fallbacks; xp increasing; no left / right / period).  n_xp == 0 is refused as numpy does (ValueError).            */
typedef struct spc_interp_args {
    int64_t n_rows;
    int32_t n_x, n_xp;
    int64_t pitch_x, pitch_xp;     /* 0 = shared by all rows */
    int64_t pitch_fp, pitch_out;
    const void *x;                 /* [n_rows x n_x]  points to evaluate at   */
    const void *xp, *fp;           /* [n_rows x n_xp] sample points / values  */
    void *out;                     /* [n_rows x n_x]                          */
} spc_interp_args;
int spc_interp_f64(const spc_interp_args *args, void *stream);
int spc_interp_f32(const spc_interp_args *args, void *stream);

/* sputils.searchsorted, splib/sputils.py:88-91 == numpy.searchsorted(a, v, side) per row; NaN sorts to the end.    */
typedef struct spc_searchsorted_args {
    int64_t n_rows;
    int32_t n_a, n_v;
    int64_t pitch_a, pitch_v;      /* 0 = shared by all rows */
    int64_t pitch_out;
    const void *a;                 /* [n_rows x n_a] sorted ascending */
    const void *v;                 /* [n_rows x n_v]                  */
    int64_t *out;                  /* [n_rows x n_v] insertion indices (numpy's intp) */
    int32_t side_right;            /* 0: side='left', 1: side='right' */
    int32_t reserved;
} spc_searchsorted_args;
int spc_searchsorted_f64(const spc_searchsorted_args *args, void *stream);
int spc_searchsorted_f32(const spc_searchsorted_args *args, void *stream);

/* sputils.integral / interp_c / interp_rho, splib/sputils.py:94-197.  Per row: Zh [nG+1] coarse layer bounds
 * (descending in the reference's use), zh [nL] fine grid points (ascending; they bound nL-1 cells), q [>= nL-1] cell
 * values, rho cell weights.
 *   mode 0 interp_c  : out[k] = integral(Zh[k+1], Zh[k], zh, q, rho) where Zh[k] < zh[nL-1], else 0 (sputils.py:185-188)
 *   mode 1 interp_rho: out[k] = integral(Zh[k+1], Zh[k], zh, q) / (Zh[k] - Zh[k+1]) where Zh[k] < zh[nL-1], else 0
 *                      (sputils.py:191-197; q is the density, rho is ignored)
 *   mode 2 integral  : out[k] = integral(Zh[k+1], Zh[k], zh, q, rho or NULL), no test against the top
 * Where integral() returns None (an end point outside zh) the output is NaN (what Q[i] = None stores).  Sums in
 * numpy's ndarray.sum() order.                                                                                      */
typedef struct spc_interp_c_args {
    int64_t n_rows;
    int32_t nG, nL;
    int64_t pitch_Zh;
    int64_t pitch_zh;              /* 0 = shared by all rows */
    int64_t pitch_q;               /* of q and rho           */
    int64_t pitch_out;
    const void *Zh, *zh, *q, *rho;
    void *out;                     /* [n_rows x nG]          */
    int32_t mode;
    int32_t reserved;
} spc_interp_c_args;
int spc_interp_c_f64(const spc_interp_c_args *args, void *stream);
int spc_interp_c_f32(const spc_interp_c_args *args, void *stream);

/* sputils.rms, splib/sputils.py:23-24: out[r] = sqrt(mean(a[r]**2)), the mean in numpy's pairwise order             */
int spc_rms_f64(int64_t n_rows, int64_t n, int64_t pitch, const void *a, void *out, void *stream);
int spc_rms_f32(int64_t n_rows, int64_t n, int64_t pitch, const void *a, void *out, void *stream);

/* ---- misc ----------------------------------------------------------------------------------- */
int spc_abi_version(void);          /* == SPC_ABI_VERSION                                          */
const char *spc_last_error(void);   /* text of the calling thread's last failure ("" if none)     */
int spc_device_count(void);         /* number of visible HIP devices (0 if none / no driver)      */
/* columns per workgroup the library would pick (pass 0 forward [lean, fused index map], 1 backward, 2 index, 3 diag,
 * 4 conservative backward): the `cb` of spc_describe_launch */
int spc_pick_cols_per_block(const spc_dims *dims, int pass);
/* WHICH kernel instantiation, slab size and grid the library launches for this batch, as text, e.g.
 *   "k_forward<f64,lean,91,160,wt=1,blk=1024,pre=1> cb=4 grid=256 block=1024 lds=17472 cus=256"
 * (template arguments: element type, lean / full output set, compile-time level counts [0,0 = run-time geometry],
 *  write-through stores, workgroup size, prologue prefetch; float batches of a compile-time geometry that take the
 *  8-byte-access forward kernel read "k_forward_f32v<91,160,wt=0>" -- pointers that are only 4-byte aligned fall back to
 *  the scalar kernel at launch time; pass 3: "k_diag<f64,91,160,wt=0>"; `cus` = the compute units of the current device
 *  the residency rules counted with, SPC_CUS overrides).  pass as above; flags (pass 0 only): bit 0 = the index
 * map is fused (idx != NULL), bit 1 = FULL variant (any optional output or surface coupling requested); elem_size 8
 * (f64) or 4 (f32).  The text comes from the very function the launchers use to choose, so a test can walk the
 * dispatch table and require that every instantiation it reaches is bit-checked (tests/test_dispatch_gpu.py).
 * Writes at most buflen-1 characters + NUL; returns the length of the full text or a negative spc_status. */
int spc_describe_launch(const spc_dims *dims, int pass, int flags, int elem_size, char *buf, int buflen);
/* Bytes of spc_vnudge_args.work spc_variability_nudge_f64 wants for these extents (n_cols*2*itot*jtot*ktot*8: qt and
 * qsat transposed to contiguous planes; required for planes of more than ~9 000 points); negative spc_status on bad
 * extents. */
int64_t spc_vnudge_workspace_bytes(int64_t n_cols, int32_t itot, int32_t jtot, int32_t ktot);

#ifdef __cplusplus
}
#endif
#endif /* SPC_H */
