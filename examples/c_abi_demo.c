/* c_abi_demo.c -- the drop-in boundary used from plain C: no Python, no torch, no HIP headers.
 *
 * Builds a small analytic batch (8 columns, 20 GCM levels <-> 32 LES levels), runs one coupling exchange through
 * include/spc.h -- spc_forward_f64 (K1 + fused index map), spc_backward_f64 (K3) -- and one standalone helper
 * (spc_interp_f64), prints checksums, and, given a file name, dumps inputs and outputs as raw float64 so that
 * tests/test_c_example_gpu.py can push the same inputs through the Python engine and compare bit for bit.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -o build/c_abi_demo -Lsp_coupler_amd -lspc_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/sp_coupler_amd -Wl,-rpath,/opt/rocm/lib
 *   ./build/c_abi_demo [dump.bin]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spc.h"

/* the four HIP runtime calls the caller of a C ABI over device pointers needs (libamdhip64) */
extern int hipMalloc(void **ptr, size_t bytes);
extern int hipFree(void *ptr);
extern int hipMemcpy(void *dst, const void *src, size_t bytes, int kind); /* 1 = host to device, 2 = device to host */
extern int hipDeviceSynchronize(void);

enum { N = 8, NG = 20, NL = 32 };

static void *to_device(const void *host, size_t bytes)
{
    void *d = NULL;
    if (hipMalloc(&d, bytes) != 0 || hipMemcpy(d, host, bytes, 1) != 0) { fprintf(stderr, "hipMalloc / hipMemcpy failed\n"); exit(2); }
    return d;
}

static void *device_zeros(size_t bytes)
{
    void *h = calloc(1, bytes), *d = to_device(h, bytes);
    free(h);
    return d;
}

static void check(int rc, const char *what)
{
    if (rc != SPC_OK) { fprintf(stderr, "%s: spc error %d: %s\n", what, rc, spc_last_error()); exit(1); }
}

static double checksum(const double *a, size_t n)
{
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += a[i] * (double)(1 + i % 7);
    return s;
}

int main(int argc, char **argv)
{
    if (spc_abi_version() != SPC_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    if (spc_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 3; }

    /* ---- an analytic batch: pressure-level GCM columns (index 0 = model top) and LES slab means ---------------- */
    static double U[N][NG], V[N][NG], T[N][NG], SH[N][NG], QL[N][NG], QI[N][NG], A[N][NG], Pf[N][NG], Ph[N][NG + 1],
        Zgf[N][NG], Zgh[N][NG + 1], zf[NL], zh[NL], u_d[N][NL], v_d[N][NL], thl_d[N][NL], qt_d[N][NL], ql_d[N][NL],
        qi_d[N][NL], t_d[N][NL], A_prof[N][NG], ps_d[N];
    for (int l = 0; l < NL; ++l) { zh[l] = 150.0 * l; zf[l] = 150.0 * l + 75.0; }
    for (int c = 0; c < N; ++c) {
        const double ps = 1.0e5 - 700.0 * c, zs = 9.81 * 40.0 * c;
        for (int k = 0; k <= NG; ++k) {
            Ph[c][k] = ps * pow((double)k / NG, 1.5);
            Zgh[c][k] = zs + 9.81 * 7400.0 * log(ps / (Ph[c][k] > 50.0 ? Ph[c][k] : 50.0));
        }
        for (int k = 0; k < NG; ++k) {
            Pf[c][k] = 0.5 * (Ph[c][k] + Ph[c][k + 1]);
            Zgf[c][k] = zs + 9.81 * 7400.0 * log(ps / Pf[c][k]);
            const double z = (Zgf[c][k] - zs) / 9.81;
            T[c][k] = fmax(210.0, 288.0 - 0.0065 * z) + 0.3 * sin(k + c);
            SH[c][k] = 0.012 * exp(-z / 2500.0);
            QL[c][k] = (z > 800.0 && z < 2500.0) ? 2.0e-4 * (1 + c % 3) : 0.0;
            QI[c][k] = (z > 2000.0 && z < 6000.0) ? 5.0e-5 : 0.0;
            A[c][k] = QL[c][k] > 0.0 ? 0.4 : 0.0;
            U[c][k] = 8.0 + 6.0 * cos(0.3 * k + c);
            V[c][k] = -3.0 + 4.0 * sin(0.2 * k - c);
            A_prof[c][k] = 0.05 * ((k + c) % 5);
        }
        ps_d[c] = ps + 12.0;
        for (int l = 0; l < NL; ++l) {
            u_d[c][l] = 7.5 + 0.002 * zf[l] + 0.1 * c;
            v_d[c][l] = -2.0 + 0.001 * zf[l];
            thl_d[c][l] = 289.0 + 0.004 * zf[l];
            qt_d[c][l] = 0.011 * exp(-zf[l] / 2600.0);
            ql_d[c][l] = (zf[l] > 900.0 && zf[l] < 2200.0) ? 1.5e-4 : 0.0;
            qi_d[c][l] = 0.25 * ql_d[c][l];
            t_d[c][l] = 288.5 - 0.006 * zf[l];
        }
    }

    /* ---- device copies (the caller owns every buffer) -------------------------------------------------------- */
#define DEV(x) void *d_##x = to_device(x, sizeof(x))
    DEV(U); DEV(V); DEV(T); DEV(SH); DEV(QL); DEV(QI); DEV(A); DEV(Pf); DEV(Ph); DEV(Zgf); DEV(Zgh); DEV(zf); DEV(zh);
    DEV(u_d); DEV(v_d); DEV(thl_d); DEV(qt_d); DEV(ql_d); DEV(qi_d); DEV(t_d); DEV(A_prof); DEV(ps_d);
    void *o_fu = device_zeros(sizeof(u_d)), *o_fv = device_zeros(sizeof(u_d)), *o_fthl = device_zeros(sizeof(u_d)),
         *o_fqt = device_zeros(sizeof(u_d)), *o_fql = device_zeros(sizeof(u_d)), *o_qlref = device_zeros(sizeof(u_d)),
         *o_fps = device_zeros(sizeof(ps_d)), *o_Zf = device_zeros(sizeof(Zgf)), *o_Zh = device_zeros(sizeof(Zgh)),
         *o_idx = device_zeros(sizeof(int32_t) * N * NG);
    void *o_fT = device_zeros(sizeof(T)), *o_fSH = device_zeros(sizeof(T)), *o_fQL = device_zeros(sizeof(T)), *o_fQI = device_zeros(sizeof(T)),
         *o_fU = device_zeros(sizeof(T)), *o_fV = device_zeros(sizeof(T)), *o_fA = device_zeros(sizeof(T)), *o_si = device_zeros(sizeof(int32_t) * N),
         *o_int = device_zeros(sizeof(T));

    const spc_dims dims = {N, NG, NL, NG, NG + 1, NL, 1, 0};
    const double factor = 1.0, dt = 900.0;

    /* ---- forward: convert_profiles + set_les_forcings (+ the cloud-fraction index map) ------------------------ */
    spc_forward_args f;
    memset(&f, 0, sizeof(f));
    f.U = d_U; f.V = d_V; f.T = d_T; f.SH = d_SH; f.QL = d_QL; f.QI = d_QI; f.Pf = d_Pf; f.Ph = d_Ph; f.Zgfull = d_Zgf; f.Zghalf = d_Zgh;
    f.zf = d_zf; f.zh = d_zh; f.u_d = d_u_d; f.v_d = d_v_d; f.thl_d = d_thl_d; f.qt_d = d_qt_d; f.ql_d = d_ql_d; f.ps_d = d_ps_d;
    f.factor = factor; f.dt = dt;
    f.f_u = o_fu; f.f_v = o_fv; f.f_thl = o_fthl; f.f_qt = o_fqt; f.f_ql = o_fql; f.ql_ref = o_qlref; f.f_ps = o_fps;
    f.Zf = o_Zf; f.Zh = o_Zh; f.idx = (int32_t *)o_idx;
    check(spc_forward_f64(&dims, &f, NULL), "spc_forward_f64");

    /* ---- backward: set_gcm_tendencies (Zf from the forward pass) ------------------------------------------------ */
    spc_backward_args b;
    memset(&b, 0, sizeof(b));
    b.T = d_T; b.SH = d_SH; b.QL = d_QL; b.QI = d_QI; b.U = d_U; b.V = d_V; b.A = d_A; b.Zf = o_Zf; b.zf = d_zf;
    b.t_d = d_t_d; b.qt_d = d_qt_d; b.ql_d = d_ql_d; b.ql_ice_d = d_qi_d; b.u_d = d_u_d; b.v_d = d_v_d; b.A_prof = d_A_prof;
    b.factor = factor; b.dt = dt;
    b.f_T = o_fT; b.f_SH = o_fSH; b.f_QL = o_fQL; b.f_QI = o_fQI; b.f_U = o_fU; b.f_V = o_fV; b.f_A = o_fA; b.start_index = (int32_t *)o_si;
    check(spc_backward_f64(&dims, &b, NULL), "spc_backward_f64");

    /* ---- one helper on its own: sputils.interp(Zf, h, u_d) per column (what f_U is taken from) ------------------- */
    spc_interp_args ia;
    memset(&ia, 0, sizeof(ia));
    ia.n_rows = N; ia.n_x = NG; ia.n_xp = NL; ia.pitch_x = NG; ia.pitch_xp = 0; ia.pitch_fp = NL; ia.pitch_out = NG;
    ia.x = o_Zf; ia.xp = d_zf; ia.fp = d_u_d; ia.out = o_int;
    check(spc_interp_f64(&ia, NULL), "spc_interp_f64");
    if (hipDeviceSynchronize() != 0) { fprintf(stderr, "device error\n"); return 2; }

    /* ---- results back ----------------------------------------------------------------------------------------------- */
    static double h_fu[N][NL], h_fthl[N][NL], h_qlref[N][NL], h_fps[N], h_Zf[N][NG], h_fT[N][NG], h_fU[N][NG], h_fA[N][NG], h_int[N][NG];
    static int32_t h_idx[N][NG], h_si[N];
#define BACK(h, d) if (hipMemcpy(h, d, sizeof(h), 2) != 0) { fprintf(stderr, "copy back failed\n"); return 2; }
    BACK(h_fu, o_fu); BACK(h_fthl, o_fthl); BACK(h_qlref, o_qlref); BACK(h_fps, o_fps); BACK(h_Zf, o_Zf); BACK(h_fT, o_fT);
    BACK(h_fU, o_fU); BACK(h_fA, o_fA); BACK(h_int, o_int); BACK(h_idx, o_idx); BACK(h_si, o_si);
    printf("f_u   checksum % .17g\n", checksum(&h_fu[0][0], N * NL));
    printf("f_thl checksum % .17g\n", checksum(&h_fthl[0][0], N * NL));
    printf("f_T   checksum % .17g\n", checksum(&h_fT[0][0], N * NG));
    printf("f_U   checksum % .17g\n", checksum(&h_fU[0][0], N * NG));
    printf("start_index:");
    for (int c = 0; c < N; ++c) printf(" %d", h_si[c]);
    printf("\nidx[0]:");
    for (int k = 0; k < NG; ++k) printf(" %d", h_idx[0][k]);
    /* closure between the fused kernel and the standalone helper: f_U = factor * (interp(Zf, h, u_d) - U) / dt, masked */
    int bad = 0;
    for (int c = 0; c < N; ++c)
        for (int k = 0; k < NG; ++k) {
            double w = factor * (h_int[c][k] - U[c][k]) / dt;
            if (k < h_si[c]) w *= 0.0;
            bad += !(w == h_fU[c][k]);
        }
    printf("\nK3 vs spc_interp_f64 + host arithmetic: %d of %d elements differ\n", bad, N * NG);

    if (argc > 1) {              /* inputs and outputs, raw, for the parity test */
        FILE *fp = fopen(argv[1], "wb");
        if (!fp) { perror(argv[1]); return 4; }
#define W(x) fwrite(x, sizeof(x), 1, fp)
        W(U); W(V); W(T); W(SH); W(QL); W(QI); W(A); W(Pf); W(Ph); W(Zgf); W(Zgh); W(zf); W(zh); W(u_d); W(v_d); W(thl_d); W(qt_d);
        W(ql_d); W(qi_d); W(t_d); W(A_prof); W(ps_d);
        W(h_fu); W(h_fthl); W(h_qlref); W(h_fps); W(h_Zf); W(h_fT); W(h_fU); W(h_fA); W(h_int);
        fclose(fp);
    }
    /* (device buffers are released with the process) */
    (void)hipFree;
    return bad ? 5 : 0;
}
