/* Host sweep of spc_pow_pos (sp_coupler_amd/csrc/spc_pow.h: the device's own source, same IEEE operations; compile with
 * -ffp-contract=off) against powl in 80-bit arithmetic, and of the C library's pow next to it.
 * usage: pow_accuracy <points per exponent> ; prints one line per (function, exponent, range).
 *        pow_accuracy <points> d : spc_div_pref0_markstein against the division instead (exit 1 on any difference). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../sp_coupler_amd/csrc/spc_pow.h"

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

int main(int argc, char **argv)
{
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
