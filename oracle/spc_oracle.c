/*
 * TEST INFRASTRUCTURE ONLY -- plain-C CPU restatement of the reference coupling math, batched.
 *
 * Second, independent restatement (the first is oracle/spcpl_oracle.py, which calls numpy.interp /
 * numpy.searchsorted like the reference does).  It takes the SAME argument structs as the product's
 * C ABI (include/spc.h) but with HOST pointers, so parity tests hand identical structs to both.
 * It may be linked / loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Pinning: see the header of oracle/spcpl_oracle.py ("parity unpinned" beyond exner/iexner/rms and
 * the cloud-fraction index map, which the reference's own tests pin).  This file is additionally
 * checked bit-for-bit (interp, indices) / to 1e-13 (pow) against the NumPy oracle in tests/.
 *
 * Citations are file:line relative to the reference root (/root/reference).
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA, like the NumPy build).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/spc.h"

/* splib/sputils.py:14-20 */
static const double pref0 = 1e5, rd = 287.04, rv = 461.5, cp = 1004., rlv = 2.53e6, grav = 9.81;

/* numpy.searchsorted side='right' (npy_binsearch<right>): first i with key < a[i]; NaN sorts last */
static int64_t ss_right(const double *a, int64_t n, double key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        double m = a[mid];
        int key_lt_m = (key < m) || (m != m && key == key);
        if (key_lt_m) hi = mid; else lo = mid + 1;
    }
    return lo;
}

/* numpy.searchsorted side='left' on the NEGATED array -a with key -v (splib/spcpl.py:498):
 * first i with !(-a[i] < -v) */
static int64_t ss_left_neg(const double *a, int64_t n, double v)
{
    double key = -v;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        double m = -a[mid];
        int m_lt_key = (m < key) || (key != key && m == m);
        if (m_lt_key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* numpy.interp for one x (arr_interp in numpy/_core/src/multiarray/compiled_base.c), with
 * left=fp[0], right=fp[n-1]; xp ascending, fp addressed through a stride so that reversed GCM
 * arrays (splib/spcpl.py:224-228, Zf[::-1]) need no copy: element i is p[i*s]. */
static double interp1(double x, const double *xp, int64_t sx, const double *fp, int64_t sf, int64_t n)
{
    if (n == 1) return fp[0]; /* numpy's lenxp == 1 branch: fp[0] for every x, NaN included */
    if (x != x) return x;
    if (x > xp[(n - 1) * sx]) return fp[(n - 1) * sf];
    if (x < xp[0]) return fp[0];
    /* j = upper_bound(xp, x) - 1 */
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (x >= xp[mid * sx]) lo = mid + 1; else hi = mid;
    }
    int64_t j = lo - 1;
    if (j == n - 1) return fp[j * sf];
    if (xp[j * sx] == x) return fp[j * sf];
    {
        double slope = (fp[(j + 1) * sf] - fp[j * sf]) / (xp[(j + 1) * sx] - xp[j * sx]);
        double r = slope * (x - xp[j * sx]) + fp[j * sf];
        if (r != r) {
            r = slope * (x - xp[(j + 1) * sx]) + fp[(j + 1) * sf];
            if (r != r && fp[j * sf] == fp[(j + 1) * sf]) r = fp[j * sf];
        }
        return r;
    }
}

static int check_dims(const spc_dims *d)
{
    if (!d || d->n_cols < 0 || d->nG < 1 || d->nL < 1) return SPC_ERR_INVALID_ARGUMENT;
    if (d->pitchG < d->nG || d->pitchGh < d->nG + 1 || d->pitchL < d->nL) return SPC_ERR_INVALID_ARGUMENT;
    return SPC_OK;
}

#define D(p) ((const double *)(p))
#define W(p) ((double *)(p))

/* splib/spcpl.py:171-246 + 299-385 + 136-167 + 761-764, for every column */
int oracle_forward_f64(const spc_dims *d, const spc_forward_args *a)
{
    int rc = check_dims(d);
    if (rc) return rc;
    const int64_t n = d->n_cols, nG = d->nG, nL = d->nL;
    double *Zf = (double *)malloc(sizeof(double) * (size_t)nG * 3);
    double *thl_ = Zf + nG, *qt_ = thl_ + nG;
    double *Zh = (double *)malloc(sizeof(double) * (size_t)(nG + 1));
    for (int64_t c = 0; c < n; ++c) {
        const int64_t g = c * d->pitchG, gh = c * d->pitchGh, l = c * d->pitchL;
        const double *zf = D(a->zf) + (d->les_grid_shared ? 0 : l);
        const double zsurf = D(a->Zghalf)[gh + nG];
        for (int64_t k = 0; k < nG; ++k) {
            double T = D(a->T)[g + k], SH = D(a->SH)[g + k], QL = D(a->QL)[g + k], QI = D(a->QI)[g + k];
            Zf[k] = (D(a->Zgfull)[g + k] - zsurf) / grav;                       /* spcpl.py:198 */
            thl_[k] = (T - (rlv * (QL + QI)) / cp) * pow(D(a->Pf)[g + k] / pref0, -rd / cp); /* :214 */
            qt_[k] = SH + QL + QI;                                              /* spcpl.py:215 */
        }
        for (int64_t k = 0; k <= nG; ++k) Zh[k] = (D(a->Zghalf)[gh + k] - zsurf) / grav; /* :197 */
        if (a->Zf) memcpy(W(a->Zf) + g, Zf, sizeof(double) * (size_t)nG);
        if (a->Zh) memcpy(W(a->Zh) + gh, Zh, sizeof(double) * (size_t)(nG + 1));
        /* reversed views: element i of X[::-1] is (X + nG-1)[-i] */
        const double *xp = Zf + nG - 1;
        for (int64_t i = 0; i < nL; ++i) {
            double h = zf[i];
            double thl = interp1(h, xp, -1, thl_ + nG - 1, -1, nG);             /* spcpl.py:224 */
            double qt = interp1(h, xp, -1, qt_ + nG - 1, -1, nG);               /* spcpl.py:225 */
            double ql = interp1(h, xp, -1, D(a->QL) + g + nG - 1, -1, nG);      /* spcpl.py:226 */
            double u = interp1(h, xp, -1, D(a->U) + g + nG - 1, -1, nG);        /* spcpl.py:227 */
            double v = interp1(h, xp, -1, D(a->V) + g + nG - 1, -1, nG);        /* spcpl.py:228 */
            W(a->f_u)[l + i] = a->factor * (u - D(a->u_d)[l + i]) / a->dt;      /* spcpl.py:328 */
            W(a->f_v)[l + i] = a->factor * (v - D(a->v_d)[l + i]) / a->dt;      /* spcpl.py:329 */
            W(a->f_thl)[l + i] = a->factor * (thl - D(a->thl_d)[l + i]) / a->dt; /* spcpl.py:330 */
            W(a->f_qt)[l + i] = a->factor * (qt - D(a->qt_d)[l + i]) / a->dt;   /* spcpl.py:331 */
            W(a->f_ql)[l + i] = a->factor * (ql - D(a->ql_d)[l + i]) / a->dt;   /* spcpl.py:333 */
            W(a->ql_ref)[l + i] = ql;                                           /* spcpl.py:347 */
            if (a->u) W(a->u)[l + i] = u;
            if (a->v) W(a->v)[l + i] = v;
            if (a->thl) W(a->thl)[l + i] = thl;
            if (a->qt) W(a->qt)[l + i] = qt;
        }
        {
            double ps = D(a->Ph)[gh + nG];                                      /* spcpl.py:246 */
            W(a->f_ps)[c] = a->factor * (ps - D(a->ps_d)[c]) / a->dt;           /* spcpl.py:332 */
            if (a->ps) W(a->ps)[c] = ps;
            if (a->rainrate && a->rain && a->rain_last)
                W(a->rainrate)[c] = (D(a->rain)[c] - D(a->rain_last)[c]) / a->dt; /* spcpl.py:325 */
            if (a->wthl && a->wqt) {                                            /* spcpl.py:136-167 */
                double rho = ps / (rd * D(a->T)[g + nG - 1]);                   /* spcpl.py:153 */
                W(a->wqt)[c] = -(D(a->QLflux)[c] + D(a->QIflux)[c] + D(a->SHflux)[c]) / rho; /* :159 */
                W(a->wthl)[c] = -D(a->TSflux)[c] * pow(ps / pref0, -rd / cp) / (cp * rho);   /* :161 */
                if (a->z0m) W(a->z0m)[c] = D(a->Z0M)[c];
                if (a->z0h) W(a->z0h)[c] = D(a->Z0H)[c];
            }
        }
        if (a->idx && a->zh) {                                                  /* spcpl.py:764 */
            const double *zh = D(a->zh) + (d->les_grid_shared ? 0 : l);
            for (int64_t m = 0; m < nG; ++m)
                a->idx[g + m] = (int32_t)ss_right(zh, nL, Zh[nG - 1 - m]);
        }
    }
    free(Zf);
    free(Zh);
    return SPC_OK;
}

/* splib/spcpl.py:26 / 764 */
int oracle_cloud_indices_f64(const spc_dims *d, const void *zh_, const void *Zh_, int32_t *idx)
{
    int rc = check_dims(d);
    if (rc) return rc;
    const int64_t nG = d->nG, nL = d->nL;
    for (int64_t c = 0; c < d->n_cols; ++c) {
        const double *zh = D(zh_) + (d->les_grid_shared ? 0 : c * d->pitchL);
        const double *Zh = D(Zh_) + c * d->pitchGh;
        for (int64_t m = 0; m < nG; ++m)
            idx[c * d->pitchG + m] = (int32_t)ss_right(zh, nL, Zh[nG - 1 - m]);
    }
    return SPC_OK;
}

/* numpy's pairwise summation of a contiguous double array (DOUBLE_pairwise_sum, numpy/_core/src/umath/
 * loops_utils.h.src): what `ndarray.sum()` evaluates at splib/sputils.py:144,152,157 */
static double np_pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        int64_t i;
        for (i = 0; i < 8; ++i) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

/* exported for tests/test_properties.py */
double oracle_pairwise_sum(const double *a, int64_t n) { return np_pairwise_sum(a, n); }

/* sputils.integral with weights (splib/sputils.py:94-161); *ok = 0 where the reference returns None */
static double integral_w(double a, double b, const double *z, int64_t nz, const double *q, const double *w,
                         double *tmp, int *ok)
{
    *ok = 1;
    if (a < z[0] || a > z[nz - 1] || b < z[0] || b > z[nz - 1]) { *ok = 0; return 0; }   /* sputils.py:113-115 */
    double sign = 1;
    if (a > b) { sign = -1; double t = a; a = b; b = t; }                                /* sputils.py:117-120 */
    int64_t ia = 0;
    while (z[ia + 1] < a) ia++;                                                          /* sputils.py:122-124 */
    int64_t ib = ia;
    while (z[ib + 1] < b) ib++;                                                          /* sputils.py:125-127 */
    const int64_t n = ib - ia + 1;
    for (int64_t i = 0; i < n; ++i) tmp[i] = w[ia + i] * q[ia + i] * (z[ia + i + 1] - z[ia + i]);
    double S = np_pairwise_sum(tmp, n);                                                  /* sputils.py:152 */
    double Sa = w[ia] * q[ia] * (a - z[ia]);                                             /* sputils.py:154 */
    double Sb = w[ib] * q[ib] * (z[ib + 1] - b);                                         /* sputils.py:155 */
    for (int64_t i = 0; i < n; ++i) tmp[i] = w[ia + i] * (z[ia + i + 1] - z[ia + i]);
    double Sw = np_pairwise_sum(tmp, n);                                                 /* sputils.py:157 */
    double Swa = w[ia] * (a - z[ia]);                                                    /* sputils.py:159 */
    double Swb = w[ib] * (z[ib + 1] - b);                                                /* sputils.py:160 */
    return (S - Sa - Sb) / (Sw - Swa - Swb) * sign;                                      /* sputils.py:161 */
}

/* sputils.interp_c (splib/sputils.py:173-189); a None from integral becomes NaN here (the reference
 * raises on `Q[i] = None`) */
static void interp_c(const double *Zh, int64_t nG, const double *zh, int64_t nL, const double *q, const double *rho,
                     double *tmp, double *Q)
{
    for (int64_t i = 0; i < nG; ++i) {
        Q[i] = 0;
        if (Zh[i] < zh[nL - 1]) {
            int ok;
            double v = integral_w(Zh[i + 1], Zh[i], zh, nL, q, rho, tmp, &ok);
            Q[i] = ok ? v : (0.0 / 0.0);
        }
    }
}

/* splib/spcpl.py:388-555: linear interpolation branch (468-478) or conservative branch (479-489) */
int oracle_backward_f64(const spc_dims *d, const spc_backward_args *a)
{
    int rc = check_dims(d);
    if (rc) return rc;
    const int64_t n = d->n_cols, nG = d->nG, nL = d->nL;
    double *Zf = (double *)malloc(sizeof(double) * (size_t)nG);
    double *qlw = (double *)malloc(sizeof(double) * (size_t)nL);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)nL);
    double *Zh = (double *)malloc(sizeof(double) * (size_t)(nG + 1));
    double *Q = (double *)malloc(sizeof(double) * (size_t)nG * 7);
    for (int64_t c = 0; c < n; ++c) {
        const int64_t g = c * d->pitchG, gh = c * d->pitchGh, l = c * d->pitchL;
        const double *h = D(a->zf) + (d->les_grid_shared ? 0 : l);
        if (a->Zf) {
            memcpy(Zf, D(a->Zf) + g, sizeof(double) * (size_t)nG);
        } else {
            const double zsurf = D(a->Zghalf)[gh + nG];
            for (int64_t k = 0; k < nG; ++k) Zf[k] = (D(a->Zgfull)[g + k] - zsurf) / grav; /* :198 */
        }
        for (int64_t i = 0; i < nL; ++i) qlw[i] = D(a->ql_d)[l + i] - D(a->ql_ice_d)[l + i]; /* :402 */
        const int64_t start_index = ss_left_neg(Zf, nG, h[nL - 1]);             /* spcpl.py:498 */
        if (a->start_index) a->start_index[c] = (int32_t)start_index;
        if (a->conservative) {                                                  /* spcpl.py:482-488 */
            const double *zh = D(a->zh) + (d->les_grid_shared ? 0 : l);
            const double *rho = D(a->rhobf_d) + l;
            if (a->Zh) memcpy(Zh, D(a->Zh) + gh, sizeof(double) * (size_t)(nG + 1));
            else for (int64_t k = 0; k <= nG; ++k) Zh[k] = (D(a->Zghalf)[gh + k] - D(a->Zghalf)[gh + nG]) / grav;
            interp_c(Zh, nG, zh, nL, D(a->t_d) + l, rho, tmp, Q);
            interp_c(Zh, nG, zh, nL, D(a->qt_d) + l, rho, tmp, Q + nG);
            interp_c(Zh, nG, zh, nL, D(a->ql_d) + l, rho, tmp, Q + 2 * nG);
            interp_c(Zh, nG, zh, nL, qlw, rho, tmp, Q + 3 * nG);
            interp_c(Zh, nG, zh, nL, D(a->ql_ice_d) + l, rho, tmp, Q + 4 * nG);
            interp_c(Zh, nG, zh, nL, D(a->u_d) + l, rho, tmp, Q + 5 * nG);
            interp_c(Zh, nG, zh, nL, D(a->v_d) + l, rho, tmp, Q + 6 * nG);
        }
        for (int64_t k = 0; k < nG; ++k) {
            double x = Zf[k];
            double t_i, qt_i, ql_i, qlw_i, qli_i, u_i, v_i;
            if (a->conservative) {
                t_i = Q[k]; qt_i = Q[nG + k]; ql_i = Q[2 * nG + k]; qlw_i = Q[3 * nG + k]; qli_i = Q[4 * nG + k];
                u_i = Q[5 * nG + k]; v_i = Q[6 * nG + k];
            } else {
                t_i = interp1(x, h, 1, D(a->t_d) + l, 1, nL);                   /* spcpl.py:471 */
                qt_i = interp1(x, h, 1, D(a->qt_d) + l, 1, nL);                 /* spcpl.py:472 */
                ql_i = interp1(x, h, 1, D(a->ql_d) + l, 1, nL);                 /* spcpl.py:473 */
                qlw_i = interp1(x, h, 1, qlw, 1, nL);                           /* spcpl.py:474 */
                qli_i = interp1(x, h, 1, D(a->ql_ice_d) + l, 1, nL);            /* spcpl.py:475 */
                u_i = interp1(x, h, 1, D(a->u_d) + l, 1, nL);                   /* spcpl.py:476 */
                v_i = interp1(x, h, 1, D(a->v_d) + l, 1, nL);                   /* spcpl.py:477 */
            }
            double A_d = D(a->A_prof)[g + nG - 1 - k];                          /* spcpl.py:404 */
            double f_T = a->factor * (t_i - D(a->T)[g + k]) / a->dt;            /* spcpl.py:518 */
            double f_SH = a->factor * ((qt_i - ql_i) - D(a->SH)[g + k]) / a->dt; /* spcpl.py:519 */
            double f_QL = a->factor * (qlw_i - D(a->QL)[g + k]) / a->dt;        /* spcpl.py:520 */
            double f_QI = a->factor * (qli_i - D(a->QI)[g + k]) / a->dt;        /* spcpl.py:521 */
            double f_U = a->factor * (u_i - D(a->U)[g + k]) / a->dt;            /* spcpl.py:524 */
            double f_V = a->factor * (v_i - D(a->V)[g + k]) / a->dt;            /* spcpl.py:525 */
            double f_A = a->factor * (A_d - D(a->A)[g + k]) / a->dt;            /* spcpl.py:526 */
            if (k < start_index) {                                              /* spcpl.py:527-533 */
                f_T *= 0; f_SH *= 0; f_QL *= 0; f_QI *= 0; f_U *= 0; f_V *= 0; f_A *= 0;
            }
            W(a->f_T)[g + k] = f_T;
            W(a->f_SH)[g + k] = f_SH;
            W(a->f_QL)[g + k] = f_QL;
            W(a->f_QI)[g + k] = f_QI;
            W(a->f_U)[g + k] = f_U;
            W(a->f_V)[g + k] = f_V;
            W(a->f_A)[g + k] = f_A;
        }
    }
    free(Zf);
    free(qlw);
    free(tmp);
    free(Zh);
    free(Q);
    return SPC_OK;
}

/* spifs diagnostics: splib/spcpl.py:176, 197-198, 214-215, 402, 408-409 */
int oracle_diagnostics_f64(const spc_dims *d, const spc_diagnostics_args *a)
{
    int rc = check_dims(d);
    if (rc) return rc;
    const int64_t n = d->n_cols, nG = d->nG, nL = d->nL;
    const double cc = rv / rd - 1;                                              /* spcpl.py:175 */
    double *Zf = (double *)malloc(sizeof(double) * (size_t)nG);
    for (int64_t c = 0; c < n; ++c) {
        const int64_t g = c * d->pitchG, gh = c * d->pitchGh, l = c * d->pitchL;
        const double zsurf = D(a->Zghalf)[gh + nG];
        for (int64_t k = 0; k < nG; ++k) {
            double T = D(a->T)[g + k], SH = D(a->SH)[g + k], QL = D(a->QL)[g + k], QI = D(a->QI)[g + k];
            Zf[k] = (D(a->Zgfull)[g + k] - zsurf) / grav;
            if (a->Tv) W(a->Tv)[g + k] = T * (1 + cc * SH - (QL + QI));         /* spcpl.py:176 */
            if (a->THL) W(a->THL)[g + k] = (T - (rlv * (QL + QI)) / cp) * pow(D(a->Pf)[g + k] / pref0, -rd / cp);
            if (a->QT) W(a->QT)[g + k] = SH + QL + QI;
            if (a->Zf) W(a->Zf)[g + k] = Zf[k];
        }
        if (a->Zh)
            for (int64_t k = 0; k <= nG; ++k) W(a->Zh)[gh + k] = (D(a->Zghalf)[gh + k] - zsurf) / grav;
        if (a->zf && (a->pf || a->t || a->ql_water)) {
            const double *h = D(a->zf) + (d->les_grid_shared ? 0 : l);
            for (int64_t i = 0; i < nL; ++i) {
                double pf = interp1(h[i], Zf + nG - 1, -1, D(a->Pf) + g + nG - 1, -1, nG); /* :408 */
                if (a->pf) W(a->pf)[l + i] = pf;
                if (a->t)                                                       /* spcpl.py:409 */
                    W(a->t)[l + i] = D(a->thl_d)[l + i] * pow(pf / pref0, rd / cp) + rlv * D(a->ql_d)[l + i] / cp;
                if (a->ql_water) W(a->ql_water)[l + i] = D(a->ql_d)[l + i] - D(a->ql_ice_d)[l + i]; /* :402 */
            }
        }
    }
    free(Zf);
    return SPC_OK;
}
