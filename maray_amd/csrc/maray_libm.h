// maray_libm.h — bit-exact sin / exp / log of the reference's platform libm
// (product code; compiled for gfx950 by hipcc and hiprtc, and for the host by
// the CPU test-suite to prove bit-equality with the system libm).
//
// The reference evaluates Sin/Exp/Ln with Rust's f64::sin/exp/ln
// (src/lib.rs:648-650), i.e. the libm of the host it runs on.  On this image
// (and on the GPU box) that is glibc 2.35, x86_64, whose IFUNC resolvers pick
// the FMA variants (__sin_fma, __exp_fma, __log_fma) on every AVX2+FMA CPU.
// This header restates those three algorithms operation for operation — the
// same tables, the same polynomial order, and fma() exactly where that build
// contracts a*b+c — so that results are identical bit for bit:
//
//   sin  sysdeps/ieee754/dbl-64/s_sin.c (IBM Accurate Mathematical Library):
//        __sin, do_sin, do_cos, do_sincos, reduce_sincos, TAYLOR_SIN; the
//        large-argument reduction of branred.c (__branred; compiled without
//        FMA in that libm).
//   exp  sysdeps/ieee754/dbl-64/e_exp.c   (ARM optimized-routines, N = 128)
//   log  sysdeps/ieee754/dbl-64/e_log.c   (ARM optimized-routines, N = 128)
//
// errno / fenv side effects are not reproduced (the reference never reads
// them).  NaN results are returned as the default NaN of the executing
// hardware; NaN payloads are not part of the parity contract.
//
// Everything must be compiled with -ffp-contract=off: only the fma() calls
// written below may be fused.
#pragma once

#if defined(__HIPCC_RTC__) || defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define MR_FN __device__ inline
#define MR_NOINLINE __device__ __attribute__((noinline))
#define MR_TABLE static __device__ const
#else
#define MR_FN static inline
#define MR_NOINLINE static __attribute__((noinline))
#define MR_TABLE static const
#endif

#include "maray_libm_tables.h"

MR_FN double mr_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
MR_FN unsigned long long mr_bits(double x) { return __builtin_bit_cast(unsigned long long, x); }
MR_FN double mr_from_bits(unsigned long long u) { return __builtin_bit_cast(double, u); }
MR_FN double mr_fabs(double x) { return __builtin_fabs(x); }
MR_FN double mr_copysign(double x, double s) { return __builtin_copysign(x, s); }
MR_FN double mr_nan() { return __builtin_nan(""); }

// ------------------------------------------------------------------- sin ----
// constants of s_sin.c / usncs.h / branred.h
#define MR_SN3 (-0x1.5555555555515p-3)
#define MR_SN5 (0x1.11110e829872fp-7)
#define MR_CS2 (0x1.0000000000000p-1)
#define MR_CS4 (-0x1.5555555555535p-5)
#define MR_CS6 (0x1.6c16bedd9e239p-10)
#define MR_S1 (-0x1.5555555555555p-3)
#define MR_S2 (0x1.1111111110ecep-7)
#define MR_S3 (-0x1.a01a019db08b8p-13)
#define MR_S4 (0x1.71de27b9a7ed9p-19)
#define MR_S5 (-0x1.addffc2fcdf59p-26)
#define MR_BIG (0x1.8p45)
#define MR_HP0 (0x1.921fb54442d18p+0)
#define MR_HP1 (0x1.1a62633145c07p-54)
#define MR_HPINV (0x1.45f306dc9c883p-1)
#define MR_TOINT (0x1.8p52)
#define MR_MP1 (0x1.921fb58000000p+0)
#define MR_MP2 (-0x1.dde973c000000p-27)
#define MR_PP3 (-0x1.cb3b398000000p-55)
#define MR_PP4 (-0x1.d747f23e32ed7p-83)

// TAYLOR_SIN(xx, x, dx): x + (((POLY(xx)*x - 0.5*dx) * xx) + dx), contracted as in __sin_fma.
MR_FN double mr_taylor_sin(double xx, double x, double dx)
{
    double p = mr_fma(xx, MR_S5, MR_S4);
    p = mr_fma(xx, p, MR_S3);
    p = mr_fma(xx, p, MR_S2);
    p = mr_fma(xx, p, MR_S1);
    const double t = mr_fma(xx, mr_fma(p, x, -(0.5 * dx)), dx);
    return x + t;
}

// do_sin(x, dx), including its |x| < 0.126 Taylor branch and the final copysign.
MR_FN double mr_do_sin(double x, double dx)
{
    if (mr_fabs(x) < 0.126) return mr_taylor_sin(x * x, x, dx);
    if (x <= 0) dx = -dx;
    const double ax = mr_fabs(x);
    const double u = MR_BIG + ax;
    const double xr = ax - (u - MR_BIG);
    const int k = (int)(unsigned)mr_bits(u) * 4;
    const double xx = xr * xr;
    const double s = xr + mr_fma(xr * xx, mr_fma(xx, MR_SN5, MR_SN3), dx);
    const double c = mr_fma(xr, dx, xx * mr_fma(xx, mr_fma(xx, MR_CS6, MR_CS4), MR_CS2));
    const double sn = mr_sincostab[k], ssn = mr_sincostab[k + 1], cs = mr_sincostab[k + 2], ccs = mr_sincostab[k + 3];
    const double cor = mr_fma(s, cs, mr_fma(-c, sn, mr_fma(s, ccs, ssn)));
    return mr_copysign(sn + cor, x);
}

// do_cos(x, dx)
MR_FN double mr_do_cos(double x, double dx)
{
    if (x < 0) dx = -dx;
    const double ax = mr_fabs(x);
    const double u = MR_BIG + ax;
    const double xr = (ax - (u - MR_BIG)) + dx;
    const int k = (int)(unsigned)mr_bits(u) * 4;
    const double xx = xr * xr;
    const double s = mr_fma(xr * xx, mr_fma(xx, MR_SN5, MR_SN3), xr);
    const double c = xx * mr_fma(xx, mr_fma(xx, MR_CS6, MR_CS4), MR_CS2);
    const double sn = mr_sincostab[k], ssn = mr_sincostab[k + 1], cs = mr_sincostab[k + 2], ccs = mr_sincostab[k + 3];
    const double cor = mr_fma(-s, sn, mr_fma(-c, cs, mr_fma(-s, ssn, ccs)));
    return cs + cor;
}

// do_sincos(a, da, n)
MR_FN double mr_do_sincos(double a, double da, int n)
{
    const double r = (n & 1) ? mr_do_cos(a, da) : mr_do_sin(a, da);
    return (n & 2) ? -r : r;
}

// __branred (branred.c): x -> (a, da, quadrant) for |x| >= 105414350; no FMA in this routine.
MR_FN void mr_branred_half(double xh, double *b_out, double *bb_out, double *sum_out)
{
    const double tm24 = 0x1p-24, big = 0x1.8p52, big1 = 0x1.8p54;
    int k = (int)((mr_bits(xh) >> 52) & 2047);
    k = (k - 450) / 24;
    if (k < 0) k = 0;
    double gor = mr_from_bits((unsigned long long)(0x63f00000u - (unsigned)((k * 24) << 20)) << 32);
    double r[6];
    for (int i = 0; i < 6; i++) {
        r[i] = xh * mr_toverp[k + i] * gor;
        gor *= tm24;
    }
    double sum = 0;
    for (int i = 0; i < 3; i++) {
        const double s = (r[i] + big) - big;
        sum += s;
        r[i] -= s;
    }
    double t = 0;
    for (int i = 0; i < 6; i++) t += r[5 - i];
    double bb = (((((r[0] - t) + r[1]) + r[2]) + r[3]) + r[4]) + r[5];
    double s = (t + big) - big;
    sum += s;
    t -= s;
    const double b = t + bb;
    bb = (t - b) + bb;
    s = (sum + big1) - big1;
    sum -= s;
    *b_out = b; *bb_out = bb; *sum_out = sum;
}

MR_FN int mr_branred(double x, double *a, double *aa)
{
    const double split = 0x1.0000002p+27, hp0 = MR_HP0, hp1 = MR_HP1, mp1 = MR_MP1, mp2 = -0x1.dde9740000000p-27;
    x *= 0x1p-600;
    double t = x * split;
    const double x1 = t - (t - x);
    const double x2 = x - x1;
    double b1, bb1, sum1, b2, bb2, sum2;
    mr_branred_half(x1, &b1, &bb1, &sum1);
    mr_branred_half(x2, &b2, &bb2, &sum2);
    double sum = sum1 + sum2;
    double b = b1 + b2;
    double bb = (mr_fabs(b1) > mr_fabs(b2)) ? (b1 - b) + b2 : (b2 - b) + b1;
    if (b > 0.5) { b -= 1.0; sum += 1.0; }
    else if (b < -0.5) { b += 1.0; sum -= 1.0; }
    double s = b + (bb + bb1 + bb2);
    t = ((b - s) + bb) + (bb1 + bb2);
    b = s * split;
    const double t1 = b - (b - s);
    const double t2 = s - t1;
    b = s * hp0;
    bb = (((t1 * mp1 - b) + t1 * mp2) + t2 * mp1) + (t2 * mp2 + s * hp1 + t * hp0);
    s = b + bb;
    t = (b - s) + bb;
    *a = s;
    *aa = t;
    return ((int)sum) & 3;
}

// The rare tail of __sin: |x| >= 105414350 (__branred), inf and NaN.  Kept out of
// line so that the common path stays small when sin is inlined many times.
MR_NOINLINE double mr_sin_huge(double x)
{
    const unsigned k = (unsigned)(mr_bits(x) >> 32) & 0x7fffffffu;
    if (k < 0x7ff00000u) {                                           // |x| < 2^1024: __branred
        double a, da;
        const int n = mr_branred(x, &a, &da);
        return mr_do_sincos(a, da, n);
    }
    return mr_nan();                                                 // x / x for inf and NaN
}

// Hook: a translation unit may route the rare tail elsewhere (the hiprtc pixel
// kernels defer such pixels to the interpreter kernel instead of paying for a
// call site per Sin op).
#ifndef MR_SIN_HUGE
#define MR_SIN_HUGE(x) mr_sin_huge(x)
#endif

// __sin (s_sin.c), FMA variant.  The three mid-range branches of the original
// (do_sin(x,0) | copysign(do_cos(hp0-|x|, hp1), x) | do_sincos(reduce_sincos(x)))
// are expressed as one do_sincos(a, da, n) call on the same (a, da, n): the
// arithmetic per branch is unchanged, only the control flow is shared.
MR_FN double maray_libm_sin(double x)
{
    const unsigned k = (unsigned)(mr_bits(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e500000u) return x;                                   // |x| < 2^-26
    if (k >= 0x419921FBu) return MR_SIN_HUGE(x);                      // |x| >= 105414350
    double a, da;
    int n;
    bool sign_of_x = false;
    if (k < 0x3feb6000u) {                                           // |x| < 0.855469: do_sin(x, 0)
        a = x; da = 0.0; n = 0;
    } else if (k < 0x400368fdu) {                                    // |x| < 2.426265: copysign(do_cos(hp0 - |x|, hp1), x)
        a = MR_HP0 - mr_fabs(x); da = MR_HP1; n = 1; sign_of_x = true;
    } else {                                                         // reduce_sincos
        const double t = mr_fma(x, MR_HPINV, MR_TOINT);
        const double xn = t - MR_TOINT;
        n = (int)(unsigned)mr_bits(t) & 3;
        const double y = mr_fma(-xn, MR_MP2, mr_fma(-xn, MR_MP1, x));
        const double t2 = mr_fma(-xn, MR_PP3, y);
        const double db = mr_fma(-MR_PP3, xn, y - t2);
        a = mr_fma(-xn, MR_PP4, t2);
        da = db + mr_fma(-xn, MR_PP4, t2 - a);
    }
    const double r = mr_do_sincos(a, da, n);
    return sign_of_x ? mr_copysign(r, x) : r;
}

// step(sin(x)) = (sin(x) >= 0 ? 1 : 0) without computing sin's magnitude: the
// result only depends on the sign __sin would return, which is fixed by the
// quadrant n and the sign of the reduced argument a — both computed here by the
// very same operations as in maray_libm_sin:
//   do_cos(a, da) = cs + cor is positive (|a| <= pi/4 + eps);
//   do_sin(a, da) returns copysign(., a) on its table path, and a + t with
//   |t| < |a| on its Taylor path unless a is tiny, where it is evaluated in full;
//   sin(x) = x for |x| < 2^-26 (so -0.0 >= 0 gives 1, like step(-0.0)).
MR_FN double maray_libm_step_sin(double x)
{
    const unsigned k = (unsigned)(mr_bits(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e500000u) return x >= 0.0 ? 1.0 : 0.0;
    if (k >= 0x419921FBu) return MR_SIN_HUGE(x) >= 0.0 ? 1.0 : 0.0;
    if (k < 0x400368fdu) return (mr_bits(x) >> 63) ? 0.0 : 1.0;       // do_sin(x,0) and copysign(do_cos(..), x): sign of x
    const double t = mr_fma(x, MR_HPINV, MR_TOINT);
    const double xn = t - MR_TOINT;
    const int n = (int)(unsigned)mr_bits(t) & 3;
    const double y = mr_fma(-xn, MR_MP2, mr_fma(-xn, MR_MP1, x));
    const double t2 = mr_fma(-xn, MR_PP3, y);
    const double a = mr_fma(-xn, MR_PP4, t2);
    bool neg;
    if (n & 1) neg = false;
    else if (mr_fabs(a) >= 0x1p-20) neg = (mr_bits(a) >> 63) != 0;
    else {                                                           // tiny reduced argument: full Taylor value
        const double db = mr_fma(-MR_PP3, xn, y - t2);
        const double da = db + mr_fma(-xn, MR_PP4, t2 - a);
        const double r = mr_taylor_sin(a * a, a, da);
        neg = !(r >= 0.0);
    }
    if (n & 2) neg = !neg;
    return neg ? 0.0 : 1.0;
}

// Branch-free form of maray_libm_step_sin for straight-line (hiprtc) kernels:
// same value on every input it decides; the inputs it does not decide — the
// huge / inf / NaN tail, and a reduced argument too small for sign(a) to be
// provably the sign of do_sin's Taylor value (|a| < 2^-70; |da| <= 2^-52|a| +
// 2^-75 by construction of reduce_sincos, and no double below 2^27 comes that
// close to a multiple of pi/2) — bump *defer (a count kept in f32) and are
// re-evaluated by the exact routine above in the interpreter kernel.
MR_FN double maray_libm_step_sin_fast(double x, float *defer)
{
    const unsigned k = (unsigned)(mr_bits(x) >> 32) & 0x7fffffffu;
    const double t = mr_fma(x, MR_HPINV, MR_TOINT);
    const double xn = t - MR_TOINT;
    const unsigned n = (unsigned)mr_bits(t);
    const double y = mr_fma(-xn, MR_MP2, mr_fma(-xn, MR_MP1, x));
    const double t2 = mr_fma(-xn, MR_PP3, y);
    const double a = mr_fma(-xn, MR_PP4, t2);
    const bool odd = (n & 1u) != 0;
    const bool neg_red = (odd ? false : (mr_bits(a) >> 63) != 0) != ((n & 2u) != 0);
    const bool small = k < 0x400368fdu;                 // sin(x) has the sign of x (and sin(+-0) = +-0 >= 0)
    const bool undecided = (k >= 0x419921FBu) | (!small & !odd & !(mr_fabs(a) >= 0x1p-70));
    *defer += undecided ? 1.0f : 0.0f;   // an FP add chain: cannot be reassociated or sunk away from this op
    const bool one = small ? (x >= 0.0) : !neg_red;
    return one ? 1.0 : 0.0;
}

// step(sin(x)) for an argument the lowering has PROVEN finite with
// |x| < 105414350 (MARAY_AUX_SIN_BOUNDED): pure and branch-free.  No huge tail
// exists here, and the Taylor-path caveat of maray_libm_step_sin is void: inside
// this range every double is farther than 2^-70 from every multiple of pi/2
// (tests/native/libm_check.cpp walks all 6.7e7 multiples and their neighbouring
// doubles against glibc), so sign(a) is the sign of do_sin(a, da).
//
// The small-argument branches of __sin need no special case here: for
// |x| < 2.426265 glibc returns a value with the sign of x (do_sin(x,0) /
// copysign(do_cos(..), x); sin(x) = x below 2^-26), and the reduction below
// yields exactly that sign (n = 0, a = x for |x| < pi/4; n odd or a = x -+ pi
// beyond; x = -0.0 reduces to a = +0.0, matching step(-0.0) = 1).
MR_FN double maray_libm_step_sin_bounded(double x)
{
    const double t = mr_fma(x, MR_HPINV, MR_TOINT);
    const double xn = t - MR_TOINT;
    const unsigned n = (unsigned)mr_bits(t);
    const double y = mr_fma(-xn, MR_MP2, mr_fma(-xn, MR_MP1, x));
    const double t2 = mr_fma(-xn, MR_PP3, y);
    const double a = mr_fma(-xn, MR_PP4, t2);
    // bit 31 of s = sign of __sin(x): sign(a) in the sine quadrants, flipped in quadrants 2 and 3
    unsigned s = (n & 1u) ? 0u : (unsigned)(mr_bits(a) >> 32);
    s ^= n << 30;
    return (int)s >= 0 ? 1.0 : 0.0;
}

// sin(x) under the same precondition: maray_libm_sin without its huge tail.
MR_FN double maray_libm_sin_bounded(double x)
{
    const unsigned k = (unsigned)(mr_bits(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e500000u) return x;
    double a, da;
    int n;
    bool sign_of_x = false;
    if (k < 0x3feb6000u) {
        a = x; da = 0.0; n = 0;
    } else if (k < 0x400368fdu) {
        a = MR_HP0 - mr_fabs(x); da = MR_HP1; n = 1; sign_of_x = true;
    } else {
        const double t = mr_fma(x, MR_HPINV, MR_TOINT);
        const double xn = t - MR_TOINT;
        n = (int)(unsigned)mr_bits(t) & 3;
        const double y = mr_fma(-xn, MR_MP2, mr_fma(-xn, MR_MP1, x));
        const double t2 = mr_fma(-xn, MR_PP3, y);
        const double db = mr_fma(-MR_PP3, xn, y - t2);
        a = mr_fma(-xn, MR_PP4, t2);
        da = db + mr_fma(-xn, MR_PP4, t2 - a);
    }
    const double r = mr_do_sincos(a, da, n);
    return sign_of_x ? mr_copysign(r, x) : r;
}

// ------------------------------------------------------------------- exp ----
MR_FN double mr_exp_specialcase(double tmp, unsigned long long sbits, unsigned long long ki)
{
    if ((ki & 0x80000000ull) == 0) {
        // k > 0, the exponent of scale might have overflowed by <= 460.
        sbits -= 1009ull << 52;
        const double scale = mr_from_bits(sbits);
        return 0x1p1009 * mr_fma(scale, tmp, scale);
    }
    // k < 0, need special care in the subnormal range.
    sbits += 1022ull << 52;
    const double scale = mr_from_bits(sbits);
    const double st = scale * tmp;
    double y = scale + st;
    if (y < 1.0) {
        // Round y to the right precision before scaling it into the subnormal range.
        const double lo = scale - y + st;
        const double hi = 1.0 + y;
        const double lo2 = 1.0 - hi + y + lo;
        y = (hi + lo2) - 1.0;
        if (y == 0.0) y = 0.0;   // avoid -0.0 with downward rounding (and always +0.0 here)
    }
    return 0x1p-1022 * y;
}

MR_FN double maray_libm_exp(double x)
{
    const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52;
    const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    unsigned abstop = (unsigned)(mr_bits(x) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x3fu) {
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;            // tiny: |x| < 2^-54
        if (abstop >= 0x409u) {                                        // |x| >= 1024
            if (mr_bits(x) == 0xfff0000000000000ull) return 0.0;       // exp(-inf)
            if (abstop >= 0x7ffu) return 1.0 + x;                      // inf / NaN
            if (mr_bits(x) >> 63) return 0x1p-767 * 0x1p-767;          // __math_uflow(0) = +0
            return 0x1p769 * 0x1p769;                                  // __math_oflow(0) = +inf
        }
        abstop = 0;                                                    // large x: handled in specialcase
    }
    // exp(x) = 2^(k/N) * exp(r), with exp(r) in [2^(-1/2N), 2^(1/2N)]
    double kd = mr_fma(x, InvLn2N, Shift);
    const unsigned long long ki = mr_bits(kd);
    kd -= Shift;
    double r = mr_fma(kd, NegLn2hiN, x);
    r = mr_fma(kd, NegLn2loN, r);
    const unsigned idx = 2u * (unsigned)(ki % 128u);
    const unsigned long long top = ki << (52 - 7);
    const double tail = mr_from_bits(mr_exp_tab[idx]);
    const unsigned long long sbits = mr_exp_tab[idx + 1] + top;
    const double r2 = r * r;
    double tmp = mr_fma(mr_fma(C3, r, C2), r2, tail + r);
    tmp = mr_fma(r2 * r2, mr_fma(r, C5, C4), tmp);
    if (abstop == 0) return mr_exp_specialcase(tmp, sbits, ki);
    const double scale = mr_from_bits(sbits);
    return mr_fma(scale, tmp, scale);
}

// ------------------------------------------------------------------- log ----
MR_FN double maray_libm_log(double x)
{
    const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    const double A0 = -0x1.0000000000001p-1, A1 = 0x1.555555551305bp-2, A2 = -0x1.fffffffeb4590p-3,
                 A3 = 0x1.999b324f10111p-3, A4 = -0x1.55575e506c89fp-3;
    const double B0 = -0x1.0000000000000p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3,
                 B3 = 0x1.999999995dd0cp-3, B4 = -0x1.55555556745a7p-3, B5 = 0x1.24924a344de30p-3,
                 B6 = -0x1.fffffa4423d65p-4, B7 = 0x1.c7184282ad6cap-4, B8 = -0x1.999eb43b068ffp-4,
                 B9 = 0x1.78182f7afd085p-4, B10 = -0x1.5521375d145cdp-4;
    unsigned long long ix = mr_bits(x);
    const unsigned top = (unsigned)(ix >> 48);
    if (ix - 0x3fee000000000000ull < 0x3090000000000ull) {            // 1 - 2^-4 <= x < 1 + 0x1.09p-4
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0;
        const double r2 = r * r;
        const double r3 = r * r2;
        double p = mr_fma(r3, B10, mr_fma(r2, B9, mr_fma(r, B8, B7)));
        p = mr_fma(p, r3, mr_fma(r2, B6, mr_fma(r, B5, B4)));
        p = mr_fma(p, r3, mr_fma(r2, B3, mr_fma(r, B2, B1)));
        // Worst-case error is around 0.507 ULP.
        const double rw = mr_fma(r, 0x1p27, r);                        // r + r*2^27
        const double rhi = mr_fma(-0x1p27, r, rw);                     // (r + w) - w
        const double rlo = r - rhi;
        const double rhi2 = rhi * rhi;
        const double hi = mr_fma(rhi2, B0, r);                         // r + rhi*rhi*B0
        double lo = mr_fma(rhi2, B0, r - hi);                          // r - hi + w
        lo = mr_fma(B0 * rlo, rhi + r, lo);
        const double y = mr_fma(p, r3, lo);
        return hi + y;
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
        if (ix * 2 == 0) return -__builtin_inf();                      // log(+-0) = -inf
        if (ix == 0x7ff0000000000000ull) return x;                     // log(inf) = inf
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return mr_nan();   // x < 0 or NaN
        ix = mr_bits(x * 0x1p52);                                      // subnormal: normalise
        ix -= 52ull << 52;
    }
    // x = 2^k z; where z is in range [OFF,2*OFF) and exact.
    const unsigned long long tmp = ix - 0x3fe6000000000000ull;
    const unsigned i = (unsigned)(tmp >> (52 - 7)) % 128u;
    const int k = (int)((long long)tmp >> 52);
    const unsigned long long iz = ix - (tmp & (0xfffull << 52));
    const double invc = mr_log_tab[2 * i], logc = mr_log_tab[2 * i + 1];
    const double z = mr_from_bits(iz);
    const double r = mr_fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w = mr_fma(kd, Ln2hi, logc);
    const double hi = r + w;
    const double lo = mr_fma(kd, Ln2lo, (w - hi) + r);
    const double r2 = r * r;
    const double q = mr_fma(mr_fma(r, A4, A3), r2, mr_fma(r, A2, A1));
    const double y = mr_fma(r * r2, q, mr_fma(r2, A0, lo));
    return y + hi;
}
