/* Host sweep of spc_pow_pos (sp_coupler_amd/csrc/spc_pow.h: the device's own source, same IEEE operations; compile with
 * -ffp-contract=off) against powl in 80-bit arithmetic, and of the C library's pow next to it.
 * usage: pow_accuracy <points per exponent> ; prints one line per (function, exponent, range).
 *        pow_accuracy <points> d : spc_div_pref0_markstein against the division instead (exit 1 on any difference).
 *        pow_accuracy <stride> f : the fp32 variant's spc_powf_pos (spc_powf.h) on EVERY stride-th float of [1e-4, 1.2] and of
 *                                  the whole positive range, against pow() in double (2^-29 float ulp: exact for this purpose),
 *                                  with the C library's powf next to it.
 *        pow_accuracy <points> p : spc_pow_pos with the reciprocal -2 ... +2 ulp off the host's (the device refines v_rcp_f64
 *                                  instead of dividing): the worst case over the five is the bound claimed for the device.
 *        (built with -shared -fPIC the file also exports spc_powf_host(x, y, out, n) for the GPU-side bit comparison) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
/* the device's reciprocal (v_rcp_f64 + two Newton steps) is within 1 ulp of 1 / x but not the host's correctly rounded quotient:
 * mode `p` pushes the host's reciprocal spc_rcp_ulps units in the last place off, a superset of what the device can return */
static int spc_rcp_ulps = 0;
static double spc_rcp_perturbed(double x)
{
    double r = 1.0 / x;
    for (int i = 0; i < spc_rcp_ulps; ++i) r = nextafter(r, INFINITY);
    for (int i = 0; i > spc_rcp_ulps; --i) r = nextafter(r, -INFINITY);
    return r;
}
#define SPC_POW_RCP_HOST(x) spc_rcp_perturbed(x)
#include "../../sp_coupler_amd/csrc/spc_pow.h"
#include "../../sp_coupler_amd/csrc/spc_powf.h"
#include <string.h>

static unsigned long long st = 88172645463325252ull;
static double urand(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) * (1.0 / 9007199254740992.0); }

static double ulp_of(double v) { int e; frexp(v, &e); return ldexp(1.0, e - 53); }

static void sweep(const char *what, double (*fn)(double, double), double y, double lo, double hi, long n, int logu)
{
    double worst = 0, wx = 0;
    long a50 = 0, a55 = 0, a60 = 0, a100 = 0;
    for (long i = 0; i < n; ++i) {
        const double u = urand();
        const double x = logu ? exp(log(lo) + u * (log(hi) - log(lo))) : lo + u * (hi - lo);
        const double got = fn(x, y);
        const long double want = powl((long double)x, (long double)y);
        const double err = (double)(fabsl((long double)got - want) / (long double)ulp_of((double)want));
        if (err > worst) { worst = err; wx = x; }
        a50 += err > 0.5; a55 += err > 0.55; a60 += err > 0.6; a100 += err > 1.0;
    }
    printf("%-10s y=%+.16f x in [%.3g, %.3g] %s: %ld points, worst %.4f ulp (x = %.17g), > 0.5 ulp %.3e, > 0.55 ulp %.3e, > 0.6 ulp %.3e, > 1 ulp %.3e\n",
           what, y, lo, hi, logu ? "log-uniform" : "uniform", n, worst, wx, (double)a50 / n, (double)a55 / n, (double)a60 / n, (double)a100 / n);
}

/* spc_div_pref0_markstein(p) against p / 1e5: random mantissas over the whole valid exponent window, the pressures of the
 * atmosphere, and arguments next to multiples of 1e5 ulp-steps (quotients next to representable numbers and midpoints) */
static long check_division(long n)
{
    long bad = 0, total = 0;
    for (long i = 0; i < n; ++i) {
        const double m = 1.0 + urand();                                /* [1, 2) */
        const int e = (int)(urand() * 1800.0) - 900;
        const double a = ldexp(m, e), b = 10.0 + urand() * 1.2e5;
        bad += spc_div_pref0_markstein(a) != a / 1e5; bad += spc_div_pref0_markstein(b) != b / 1e5;
        total += 2;
    }
    for (long i = 0; i < n / 4; ++i) {                                 /* p = RN(k ulp-steps times 1e5) and its neighbours */
        const double q = ldexp(1.0 + urand(), (int)(urand() * 60.0) - 30);
        const double p0 = q * 1e5;
        double p = p0;
        for (int s = 0; s < 4; ++s) { bad += spc_div_pref0_markstein(p) != p / 1e5; p = nextafter(p, INFINITY); ++total; }
        p = p0;
        for (int s = 0; s < 4; ++s) { p = nextafter(p, 0.0); bad += spc_div_pref0_markstein(p) != p / 1e5; ++total; }
    }
    const double edge[4] = {0x1p-900, 0x1p+900, 1e5, 101325.0};
    for (int i = 0; i < 4; ++i) { bad += spc_div_pref0_markstein(edge[i]) != edge[i] / 1e5; ++total; }
    printf("division   spc_div_pref0_markstein(p) vs p / 1e5: %ld arguments, %ld differ\n", total, bad);
    return bad;
}


static float f_of(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static unsigned u_of(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static double ulpf_of(double v) { int e; frexp(v, &e); if (e < -125) e = -125; return ldexp(1.0, e - 24); }

static void sweepf(const char *what, float (*fn)(float, float), float y, float lo, float hi, unsigned stride)
{
    double worst = 0; float wx = 0;
    long n = 0, a50 = 0, a51 = 0, a100 = 0;
    for (unsigned u = u_of(lo); u <= u_of(hi); u += stride) {
        const float x = f_of(u);
        const double want = pow((double)x, (double)y);
        const double got = (double)fn(x, y);
        const double err = fabs(got - want) / ulpf_of(want);
        if (err > worst) { worst = err; wx = x; }
        a50 += err > 0.5; a51 += err > 0.501; a100 += err > 1.0; ++n;
    }
    printf("%-10s y=%+.9g x in [%.3g, %.3g] every %u-th float: %ld points, worst %.5f ulp (x = %.9g), > 0.5 ulp %.3e, > 0.501 ulp %.3e, > 1 ulp %.3e\n",
           what, (double)y, (double)lo, (double)hi, stride, n, worst, (double)wx, (double)a50 / n, (double)a51 / n, (double)a100 / n);
}

static int float_mode(unsigned stride)
{
    const float rd = 287.04f, cp = 1004.f;
    const float ys[2] = {(-rd) / cp, rd / cp};                       /* the kernels' own float exponents */
    for (int k = 0; k < 2; ++k) {
        sweepf("spc_powf", spc_powf_pos, ys[k], 1e-4f, 1.2f, stride);
        sweepf("spc_powf", spc_powf_pos, ys[k], 1.4e-45f, 3.4e38f, stride * 16u + 1u);
        sweepf("libm powf", powf, ys[k], 1e-4f, 1.2f, stride * 4u + 1u);
    }
    return 0;
}

/* the header's function on a vector, for tests/test_sputils_gpu.py (this file built with -shared): the GPU box compares the
 * device's fp32 exner operator with it bit for bit */
void spc_powf_host(const float *x, float y, float *out, long n)
{
    for (long i = 0; i < n; ++i) out[i] = spc_powf_pos(x[i], y);
}

int main(int argc, char **argv)
{
    if (argc > 2 && argv[2][0] == 'f') return float_mode((unsigned)atol(argv[1]));
    if (argc > 2 && argv[2][0] == 'p') {                              /* reciprocal -2 ... +2 ulp off: the device's envelope */
        const long n = atol(argv[1]);
        const double rd = 287.04, cp = 1004.;
        const double ys[2] = {(-rd) / cp, rd / cp};
        for (spc_rcp_ulps = -2; spc_rcp_ulps <= 2; ++spc_rcp_ulps) {
            char what[32];
            snprintf(what, sizeof(what), "rcp%+dulp", spc_rcp_ulps);
            for (int k = 0; k < 2; ++k) {
                sweep(what, spc_pow_pos, ys[k], 1e-6, 1.2, n, 0);
                sweep(what, spc_pow_pos, ys[k], 1e-300, 1e300, n / 4, 1);
            }
        }
        spc_rcp_ulps = 0;
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'd') return check_division(atol(argv[1])) != 0;
    const long n = argc > 1 ? atol(argv[1]) : 20000000;
    const double rd = 287.04, cp = 1004.;
    const double ys[2] = {(-rd) / cp, rd / cp};                      /* sputils.py:34 / :29 as doubles */
    for (int k = 0; k < 2; ++k) {
        sweep("spc_pow", spc_pow_pos, ys[k], 1e-6, 1.2, n, 0);       /* p / pref0 of the atmosphere: 0.1 Pa .. 1.2e5 Pa */
        sweep("spc_pow", spc_pow_pos, ys[k], 1e-8, 2.0, n, 1);
        sweep("spc_pow", spc_pow_pos, ys[k], 1e-300, 1e300, n / 4, 1);
        sweep("libm pow", pow, ys[k], 1e-6, 1.2, n / 4, 0);
    }
    return 0;
}
