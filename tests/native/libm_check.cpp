// libm_check.cpp — host build of maray_amd/csrc/maray_libm.h compared bit for
// bit against the system libm (glibc): test infrastructure.
// Usage: libm_check <samples per range> ; exits non-zero on any mismatch.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "maray_libm.h"

static uint64_t rng_state = 0x6d61726179ull;
static uint64_t rng()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double uni(double lo, double hi) { return lo + (hi - lo) * ((rng() >> 11) * 0x1p-53); }
static double from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static bool same(double a, double b) { return (a != a && b != b) || bits(a) == bits(b); }

static double ref_step_sin(double x) { return sin(x) >= 0.0 ? 1.0 : 0.0; }
static long n_deferred = 0;
static double fast_step_sin(double x)
{
    float d = 0.0f;
    const double r = maray_libm_step_sin_fast(x, &d);
    if (d != 0.0f) { n_deferred++; return ref_step_sin(x); }   // deferred inputs are recomputed by the exact routine
    return r;
}
static double bounded_step_sin(double x) { return fabs(x) < 105414350.0 ? maray_libm_step_sin_bounded(x) : ref_step_sin(x); }
static double bounded_sin(double x) { return fabs(x) < 105414350.0 ? maray_libm_sin_bounded(x) : sin(x); }
typedef double (*fn)(double);
static long check(const char *name, fn mine, fn ref, double x, long &bad)
{
    double a = mine(x), b = ref(x);
    if (!same(a, b)) {
        if (bad < 10) printf("MISMATCH %s(%a) = %a, libm %a\n", name, x, a, b);
        bad++;
    }
    return 1;
}

int main(int argc, char **argv)
{
    long n = argc > 1 ? atol(argv[1]) : 1000000;
    long bad_sin = 0, bad_exp = 0, bad_log = 0, total = 0;
    const double specials[] = {0.0, -0.0, 1.0, -1.0, 0.5, 2.0, INFINITY, -INFINITY, NAN, 0x1p-1022, 0x1p-1074, -0x1p-1074,
                               0x1.fffffffffffffp1023, -0x1.fffffffffffffp1023, 0.126, 0.855469, 2.426265, 105414350.0,
                               105414357.85, 1e22, 0x1p-26, 0x1p-27, 709.782712893384, 709.79, -745.13, -745.14, -708.4,
                               1024.0, -1024.0, 0x1p-54, 0x1p-55, 0.9375, 1.0647, 3.141592653589793, 6.283185307179586,
                               1.5707963267948966, 0x1.921fb54442d18p+1, 1e300, 1e-300, 22.0, 355.0, 0x1.6ac5b262ca1ffp+849};
    for (double s : specials)
        for (int k = -3; k <= 3; k++) {
            double x = from_bits(bits(s) + (uint64_t)(int64_t)k);
            total += check("sin", maray_libm_sin, sin, x, bad_sin);
            total += check("step_sin", maray_libm_step_sin, ref_step_sin, x, bad_sin);
            total += check("step_sin_fast", fast_step_sin, ref_step_sin, x, bad_sin);
            total += check("step_sin_bounded", bounded_step_sin, ref_step_sin, x, bad_sin);
            total += check("sin_bounded", bounded_sin, sin, x, bad_sin);
            total += check("exp", maray_libm_exp, exp, x, bad_exp);
            total += check("log", maray_libm_log, log, x, bad_log);
        }
    // sin: every branch of __sin
    const double sin_ranges[][2] = {{-0x1p-25, 0x1p-25}, {-0.13, 0.13}, {-0.86, 0.86}, {-2.43, 2.43}, {-30, 30}, {-1e4, 1e4},
                                    {-1.06e8, 1.06e8}, {-1e15, 1e15}};
    for (auto &r : sin_ranges)
        for (long i = 0; i < n; i++) {
            const double x = uni(r[0], r[1]);
            total += check("sin", maray_libm_sin, sin, x, bad_sin);
            total += check("step_sin", maray_libm_step_sin, ref_step_sin, x, bad_sin);
            total += check("step_sin_fast", fast_step_sin, ref_step_sin, x, bad_sin);
            total += check("step_sin_bounded", bounded_step_sin, ref_step_sin, x, bad_sin);
            total += check("sin_bounded", bounded_sin, sin, x, bad_sin);
        }
    for (long i = 0; i < n; i++) {   // random bit patterns: all exponents incl. the __branred range, inf, NaN
        double x = from_bits(rng());
        total += check("sin", maray_libm_sin, sin, x, bad_sin);
        total += check("step_sin", maray_libm_step_sin, ref_step_sin, x, bad_sin);
            total += check("step_sin_fast", fast_step_sin, ref_step_sin, x, bad_sin);
            total += check("step_sin_bounded", bounded_step_sin, ref_step_sin, x, bad_sin);
            total += check("sin_bounded", bounded_sin, sin, x, bad_sin);
        total += check("exp", maray_libm_exp, exp, x, bad_exp);
        total += check("log", maray_libm_log, log, x, bad_log);
    }
    for (long i = 0; i < n; i++) {   // multiples of pi/2 and their neighbourhoods
        double x = (double)(rng() % 100000000) * 1.5707963267948966;
        const double xx = from_bits(bits(x) + (rng() % 5) - 2);
        total += check("sin", maray_libm_sin, sin, xx, bad_sin);
        total += check("step_sin", maray_libm_step_sin, ref_step_sin, xx, bad_sin);
        total += check("step_sin_fast", fast_step_sin, ref_step_sin, xx, bad_sin);
        total += check("step_sin_bounded", bounded_step_sin, ref_step_sin, xx, bad_sin);
    }
    if (argc > 2) {   // exhaustive: the doubles next to EVERY multiple of pi/2 inside reduce_sincos's range
        const long double hpi = 1.57079632679489661923132169163975144L;
        long walked = 0;
        for (long kk = 1; kk <= 67108864; kk++) {
            const double c = (double)((long double)kk * hpi);
            if (c >= 105414350.0) break;
            for (int d = -2; d <= 2; d++) {
                const double x = from_bits(bits(c) + (uint64_t)(int64_t)d);
                total += check("step_sin_bounded", bounded_step_sin, ref_step_sin, x, bad_sin);
                total += check("step_sin_bounded", bounded_step_sin, ref_step_sin, -x, bad_sin);
                walked++;
            }
        }
        printf("walked %ld doubles next to multiples of pi/2\n", walked);
    }
    const double exp_ranges[][2] = {{-1e-10, 1e-10}, {-1, 1}, {-40, 40}, {-745.2, -708}, {709, 710}, {-1100, 1100}};
    for (auto &r : exp_ranges)
        for (long i = 0; i < n; i++) total += check("exp", maray_libm_exp, exp, uni(r[0], r[1]), bad_exp);
    const double log_ranges[][2] = {{0.93, 1.07}, {0.5, 2}, {1e-3, 1e3}, {0, 1e-300}, {1e300, 1.7e308}, {-1, 1e-310}};
    for (auto &r : log_ranges)
        for (long i = 0; i < n; i++) total += check("log", maray_libm_log, log, uni(r[0], r[1]), bad_log);
    for (long i = 0; i < n; i++) {   // positive random bit patterns for log, incl. subnormals
        double x = from_bits(rng() >> 1);
        total += check("log", maray_libm_log, log, x, bad_log);
    }
    printf("step_sin_fast deferred %ld inputs to the exact routine\n", n_deferred);
    printf("checked %ld values: sin mismatches %ld, exp mismatches %ld, log mismatches %ld\n", total, bad_sin, bad_exp, bad_log);
    return (bad_sin || bad_exp || bad_log) ? 1 : 0;
}
