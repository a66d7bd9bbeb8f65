// device_math.h — scalar f64 semantics of the tape ops on gfx950 (product code).
//
// One definition of every op of `Expr::eval2` (src/lib.rs:623-670) shared by the
// tape interpreter kernels and the hiprtc-specialised kernels.  Compiled with
// -ffp-contract=off: a*b+c is never fused unless written as fma().
#pragma once

#include "maray_libm.h"

#define MARAY_DEV __device__ __forceinline__

// Booleans of the specialised kernels: one bit per lane of the wavefront, held in an SGPR pair.
typedef unsigned long long mr_mask;
#define MR_ALL (~0ull)
#define MR_NONE 0ull
#define mr_ballot(c) __builtin_amdgcn_ballot_w64(c)            /* per-lane condition -> mask (v_cmp writes it) */

// mask -> the f64 it stands for, per lane: {hi, 0} with hi = m[lane] ? on : off, one v_cndmask_b32 on the mask.  (Inline asm rather than __builtin_amdgcn_inverse_ballot_w64: a process
// that imported PyTorch first compiles with PyTorch's bundled ROCm 7.0 hiprtc, which lacks that builtin.)
MARAY_DEV double mr_mask_f64(mr_mask m, unsigned on, unsigned off)
{
    // The mask goes through VCC: with an "s" operand the compiler may hand over EXEC itself (a mask that is ballot(true)
    // IS exec), which v_cndmask cannot encode as its selector -- the upper 32 lanes then came out wrong.
    unsigned hi;
    asm("s_mov_b64 vcc, %3\n\tv_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(hi) : "v"(off), "v"(on), "s"(m) : "vcc");
    return __builtin_bit_cast(double, (unsigned long long)hi << 32);
}
MARAY_DEV double mr_pos(mr_mask m) { return mr_mask_f64(m, 0x3ff00000u, 0u); }               /* +1.0 : +0.0 */
// v on the lanes of m, +0.0 on the others (two v_cndmask_b32 on the mask)
MARAY_DEV double mr_sel0(mr_mask m, double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned lo, hi;
    asm("s_mov_b64 vcc, %4\n\tv_cndmask_b32_e32 %0, 0, %2, vcc\n\tv_cndmask_b32_e32 %1, 0, %3, vcc"
        : "=&v"(lo), "=v"(hi) : "v"((unsigned)u), "v"((unsigned)(u >> 32)), "s"(m) : "vcc");
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
MARAY_DEV double mr_neg01(mr_mask m) { return mr_mask_f64(m, 0xbff00000u, 0x80000000u); }    /* -1.0 : -0.0 */

// Neg Abs Recip Sqrt: IEEE-754 exact (:640-643).  1.0/a and sqrt lower to the
// correctly rounded f64 expansions (no fast-math).
MARAY_DEV double mr_neg(double a) { return -a; }
MARAY_DEV double mr_abs(double a) { return __builtin_fabs(a); }
MARAY_DEV double mr_recip(double a) { return 1.0 / a; }
// Sqrt: the compiler's correctly rounded expansion scales its argument by 2^256 when it is below 2^-767 (and the root
// back by 2^-128) so that the Newton steps keep their precision: a compare, two selects and two v_ldexp_f64 per lane for
// a case pixels do not meet.  Here the wavefront asks once whether any lane needs the scaling (zeros do not: the final
// class select returns them) and otherwise runs the same steps on the bare argument -- identical bits, 14 instructions
// instead of 18.
MARAY_DEV bool mr_sqrt_needs_scaling(double a)          // wave-uniform
{
    return __builtin_expect(mr_ballot(a < 0x1p-767) != 0ull, 0) && mr_ballot(a < 0x1p-767 && a != 0.0) != 0ull;
}
MARAY_DEV double mr_sqrt_unscaled(double a)
{
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, a);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, a);
    g = __builtin_fma(d, h, g);
    return __builtin_amdgcn_class(a, 0x260) ? a : g;         // +-0 and +inf are their own roots
}
MARAY_DEV double mr_sqrt(double a) { return mr_sqrt_needs_scaling(a) ? __builtin_sqrt(a) : mr_sqrt_unscaled(a); }
// Step (:644-647): NaN -> 0, -0.0 -> 1.
MARAY_DEV double mr_step(double a) { return a >= 0.0 ? 1.0 : 0.0; }
// f64::max / f64::min (:655-658): NaN-ignoring = v_max_f64 / v_min_f64
// (quiet NaN operand -> the other operand; -0 < +0).
MARAY_DEV double mr_max(double a, double b) { return __builtin_fmax(a, b); }
MARAY_DEV double mr_min(double a, double b) { return __builtin_fmin(a, b); }
// Sin Exp Ln (:648-650): the platform libm of the reference host = glibc 2.35
// x86_64 FMA variants, reproduced bit for bit (maray_libm.h).
MARAY_DEV double mr_sin(double a) { return maray_libm_sin(a); }
MARAY_DEV double mr_stepsin(double a) { return maray_libm_step_sin(a); }
MARAY_DEV double mr_stepsin_fast(double a, float *defer) { return maray_libm_step_sin_fast(a, defer); }
MARAY_DEV double mr_stepsin_bounded(double a) { return maray_libm_step_sin_bounded(a); }
MARAY_DEV bool mr_stepsin_bounded_b(double a) { return maray_libm_step_sin_bounded(a) != 0.0; }
MARAY_DEV double mr_sin_bounded(double a) { return maray_libm_sin_bounded(a); }
MARAY_DEV double mr_exp(double a) { return maray_libm_exp(a); }
MARAY_DEV double mr_ln(double a) { return maray_libm_log(a); }

// value -> lane mask, and the one test a region needs: is any lane's bit set
MARAY_DEV mr_mask mr_ge0(double a) { return mr_ballot(a >= 0.0); }       // Step
MARAY_DEV mr_mask mr_gek(double a, double k) { return mr_ballot(a >= k); }      // Step(a + (-k)), see jit_emit.hpp
MARAY_DEV mr_mask mr_lek(double a, double k) { return mr_ballot(a <= k); }      // Step(-(a + (-k)))
MARAY_DEV mr_mask mr_ne0(double a) { return mr_ballot(a != 0.0); }
MARAY_DEV mr_mask mr_ne1(double a) { return mr_ballot(a != 1.0); }
MARAY_DEV mr_mask mr_stepsin_bounded_m(double a) { return mr_ballot(maray_libm_step_sin_bounded(a) != 0.0); }
// The same with the six constants of the argument reduction read from k[0..5] (scalar loads from the kernel's constant
// table: one cache line) instead of being 64-bit literals: gfx950 has no 64-bit literal operand, and a literal costs two
// s_mov on the scalar unit each time a leaf is entered -- 12 of a chess leaf's ~50 scalar instructions.
// k = {2/pi, 1.5 * 2^52, mp1, mp2, pp3, pp4} (maray_libm.h MR_HPINV .. MR_PP4): the operations are those of
// maray_libm_step_sin_bounded, operand for operand.
MARAY_DEV mr_mask mr_stepsin_bounded_mk(double x, const __attribute__((address_space(4))) double *k)
{
    const double hpinv = k[0], toint = k[1], mp1 = k[2], mp2 = k[3], pp3 = k[4], pp4 = k[5];
    const double t = mr_fma(x, hpinv, toint);
    const double xn = t - toint;
    const unsigned n = (unsigned)mr_bits(t);
    const double y = mr_fma(-xn, mp2, mr_fma(-xn, mp1, x));
    const double t2 = mr_fma(-xn, pp3, y);
    const double a = mr_fma(-xn, pp4, t2);
    unsigned s_ = (n & 1u) ? 0u : (unsigned)(mr_bits(a) >> 32);
    s_ ^= n << 30;
    return mr_ballot((int)s_ >= 0);
}
// maray_libm_step_sin_fast the same way (k[6] = 2^-70, the bound below which the sign of the reduced argument is not
// taken for the sign of the sine)
MARAY_DEV double mr_stepsin_fast_k(double x, float *defer, const __attribute__((address_space(4))) double *k)
{
    const double hpinv = k[0], toint = k[1], mp1 = k[2], mp2 = k[3], pp3 = k[4], pp4 = k[5], tiny = k[6];
    const unsigned kx = (unsigned)(mr_bits(x) >> 32) & 0x7fffffffu;
    const double t = mr_fma(x, hpinv, toint);
    const double xn = t - toint;
    const unsigned n = (unsigned)mr_bits(t);
    const double y = mr_fma(-xn, mp2, mr_fma(-xn, mp1, x));
    const double t2 = mr_fma(-xn, pp3, y);
    const double a = mr_fma(-xn, pp4, t2);
    const bool odd = (n & 1u) != 0;
    const bool neg_red = (odd ? false : (mr_bits(a) >> 63) != 0) != ((n & 2u) != 0);
    const bool small = kx < 0x400368fdu;
    const bool undecided = (kx >= 0x419921FBu) | (!small & !odd & !(mr_fabs(a) >= tiny));
    *defer += undecided ? 1.0f : 0.0f;
    const bool one = small ? (x >= 0.0) : !neg_red;
    return one ? 1.0 : 0.0;
}
MARAY_DEV bool mr_any(mr_mask m) { return m != MR_NONE; }
// a y value known to be +0.0 or 1.0 (the same on every lane) as a lane mask: one scalar load and compare on its high word
// (the high word is 0 or 0x3ff00000: bit 20 spread over a mask by integer arithmetic.  Written as a select, the back end may
// keep the boolean in a VGPR across blocks, and the mask then reaches v_cndmask's scalar operand from a vector register.)
MARAY_DEV mr_mask mr_ym(const __attribute__((address_space(4))) unsigned *yw, unsigned k) { return (mr_mask)0 - (mr_mask)((yw[2u * k + 1u] >> 20) & 1u); }

// Rust `f64 as u8` (src/render.rs:92-94): saturating, NaN -> 0, truncation.
MARAY_DEV unsigned mr_cast_u8(double v)
{
    // v_cvt_u32_f64 truncates toward zero, saturates (negative -> 0, >= 2^32 -> 0xffffffff) and turns NaN into 0:
    // Rust's `as u32`; the min makes it `as u8`.  Inline asm: the C cast is undefined out of range, the instruction is not.
    unsigned u;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(u) : "v"(v));
    return u < 255u ? u : 255u;
}
// Rust `f64 as u32` (src/textures.rs:32-33): saturating, NaN -> 0.
MARAY_DEV unsigned mr_cast_u32(double v)
{
    unsigned u;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(u) : "v"(v));      // see mr_cast_u8
    return u;
}

struct MarayTex {
    const unsigned char *rgb;
    unsigned w, h;
};

// textures::fun_color_channel (src/textures.rs:27-36): id = image*5 + channel.
MARAY_DEV double mr_app(const MarayTex *tex, unsigned id, double x, double y)
{
    const MarayTex t = tex[id / 5u];
    const unsigned sel = id % 5u;
    if (x < 0.0 || y < 0.0) return 0.0;              // :30
    const unsigned xi = mr_cast_u32(x), yi = mr_cast_u32(y);   // :32-33
    if (xi >= t.w || yi >= t.h) return 0.0;          // :34
    return (double)t.rgb[((size_t)yi * t.w + xi) * 3u + sel];  // :35
}
// The three channels of one texel (the specialised kernels, jit_emit.hpp): a scene samples a texture as
// App(channel(i, 0), u, v), App(channel(i, 1), u, v), App(channel(i, 2), u, v) (examples/test6.rs:5-10), three calls of
// fun_color_channel with the same coordinates.  The coordinate tests, the two casts and the address are the same for the
// three: done once (mr_texel), and a channel is one byte load off that address (mr_texch).  No early return: a lane
// outside the image reads a byte that is certainly there (`safe`: the descriptor table) and selects 0.0 -- the byte loads of
// a wavefront issue back to back instead of behind two divergent branches each.
typedef const __attribute__((address_space(1))) unsigned char *mr_gbytes;      // global memory, said so: global_load, not flat_load
struct mr_tx { mr_gbytes p; bool ok; };
MARAY_DEV mr_tx mr_texel(const MarayTex &t, const void *safe, double x, double y)
{
    const unsigned xi = mr_cast_u32(x), yi = mr_cast_u32(y);                       // src/textures.rs:32-33
    mr_tx r;
    r.ok = !(x < 0.0) & !(y < 0.0) & (xi < t.w) & (yi < t.h);                     // :30, :34 (a NaN coordinate is not < 0: it casts to 0)
    r.p = r.ok ? (mr_gbytes)t.rgb + ((size_t)yi * t.w + xi) * 3u : (mr_gbytes)safe;
    return r;
}
MARAY_DEV double mr_texch(const mr_tx &t, unsigned sel) { const unsigned b = t.p[sel]; return t.ok ? (double)b : 0.0; }   // :35

// fun_image_width / fun_image_height (src/textures.rs:40-50).
MARAY_DEV double mr_texdim(const MarayTex *tex, unsigned id)
{
    const MarayTex t = tex[id / 5u];
    return (id % 5u) == 3u ? (double)t.w : (double)t.h;
}

#ifdef MR_VEC4
// ---- two rows per lane (the specialised PIXEL kernel's busy tiles) ------------------------------------------------
// A wavefront that owns the same 64 pixels of two neighbouring rows of one guard rectangle enters the same shapes for
// both (one set of guard bits covers the rectangle's 32 rows): the branch-table dispatch, the shape's constants
// (scalar loads and their waits) and everything that depends on x alone are paid once for 128 pixels.  A value that
// depends on the row -- it read Y or a y value somewhere -- is a pair (mr_p: .a = row r, .b = row r + 1), a boolean
// a pair of lane masks (mr_pm); values of x and constants alone stay single and widen where an operation mixes them
// (the converting constructors).  The generated text is the same as for one row; the emitter types each variable.
struct mr_p {
    double a, b;
    MARAY_DEV mr_p() {}
    MARAY_DEV mr_p(double s) : a(s), b(s) {}
    MARAY_DEV mr_p(double a_, double b_) : a(a_), b(b_) {}
};
struct mr_pm {
    mr_mask a, b;
    MARAY_DEV mr_pm() {}
    MARAY_DEV mr_pm(mr_mask s) : a(s), b(s) {}
    MARAY_DEV mr_pm(mr_mask a_, mr_mask b_) : a(a_), b(b_) {}
};
#define MR_PAIR1(f, v) mr_p(f((v).a), f((v).b))
MARAY_DEV mr_p operator+(const mr_p &x, const mr_p &y) { return mr_p(x.a + y.a, x.b + y.b); }
MARAY_DEV mr_p operator*(const mr_p &x, const mr_p &y) { return mr_p(x.a * y.a, x.b * y.b); }
MARAY_DEV mr_pm operator&(const mr_pm &x, const mr_pm &y) { return mr_pm(x.a & y.a, x.b & y.b); }
MARAY_DEV mr_pm operator|(const mr_pm &x, const mr_pm &y) { return mr_pm(x.a | y.a, x.b | y.b); }
MARAY_DEV mr_pm &operator|=(mr_pm &x, const mr_pm &y) { x.a |= y.a; x.b |= y.b; return x; }
MARAY_DEV mr_pm operator~(const mr_pm &x) { return mr_pm(~x.a, ~x.b); }
MARAY_DEV bool mr_any(const mr_pm &m) { return (m.a | m.b) != MR_NONE; }
MARAY_DEV bool mr_covered(mr_mask m) { return m == MR_ALL; }                       // every lane has its 1: a reduction's OR may stop
MARAY_DEV bool mr_covered(const mr_pm &m) { return (m.a & m.b) == MR_ALL; }
MARAY_DEV mr_p mr_pos(const mr_pm &m) { return mr_p(mr_pos(m.a), mr_pos(m.b)); }
MARAY_DEV mr_p mr_neg01(const mr_pm &m) { return mr_p(mr_neg01(m.a), mr_neg01(m.b)); }
MARAY_DEV mr_p mr_sel0(const mr_pm &m, const mr_p &v) { return mr_p(mr_sel0(m.a, v.a), mr_sel0(m.b, v.b)); }
MARAY_DEV mr_pm mr_ge0(const mr_p &v) { return mr_pm(mr_ge0(v.a), mr_ge0(v.b)); }
MARAY_DEV mr_pm mr_gek(const mr_p &v, double k) { return mr_pm(mr_gek(v.a, k), mr_gek(v.b, k)); }
MARAY_DEV mr_pm mr_lek(const mr_p &v, double k) { return mr_pm(mr_lek(v.a, k), mr_lek(v.b, k)); }
MARAY_DEV mr_pm mr_ne0(const mr_p &v) { return mr_pm(mr_ne0(v.a), mr_ne0(v.b)); }
MARAY_DEV mr_pm mr_ne1(const mr_p &v) { return mr_pm(mr_ne1(v.a), mr_ne1(v.b)); }
MARAY_DEV mr_pm mr_stepsin_bounded_m(const mr_p &v) { return mr_pm(mr_stepsin_bounded_m(v.a), mr_stepsin_bounded_m(v.b)); }
MARAY_DEV mr_pm mr_stepsin_bounded_mk(const mr_p &v, const __attribute__((address_space(4))) double *k) { return mr_pm(mr_stepsin_bounded_mk(v.a, k), mr_stepsin_bounded_mk(v.b, k)); }
MARAY_DEV mr_p mr_stepsin_fast_k(const mr_p &v, float *defer, const __attribute__((address_space(4))) double *k) { return mr_p(mr_stepsin_fast_k(v.a, defer, k), mr_stepsin_fast_k(v.b, defer, k)); }
MARAY_DEV mr_p mr_stepsin_fast(const mr_p &v, float *defer) { return mr_p(mr_stepsin_fast(v.a, defer), mr_stepsin_fast(v.b, defer)); }
MARAY_DEV mr_p mr_stepsin(const mr_p &v) { return MR_PAIR1(mr_stepsin, v); }
MARAY_DEV mr_p mr_neg(const mr_p &v) { return MR_PAIR1(mr_neg, v); }
MARAY_DEV mr_p mr_abs(const mr_p &v) { return MR_PAIR1(mr_abs, v); }
MARAY_DEV mr_p mr_recip(const mr_p &v) { return MR_PAIR1(mr_recip, v); }
MARAY_DEV mr_p mr_sqrt(const mr_p &v) { return MR_PAIR1(mr_sqrt, v); }
MARAY_DEV mr_p mr_sin(const mr_p &v) { return MR_PAIR1(mr_sin, v); }
MARAY_DEV mr_p mr_sin_bounded(const mr_p &v) { return MR_PAIR1(mr_sin_bounded, v); }
MARAY_DEV mr_p mr_exp(const mr_p &v) { return MR_PAIR1(mr_exp, v); }
MARAY_DEV mr_p mr_ln(const mr_p &v) { return MR_PAIR1(mr_ln, v); }
MARAY_DEV mr_p mr_max(const mr_p &x, const mr_p &y) { return mr_p(mr_max(x.a, y.a), mr_max(x.b, y.b)); }
MARAY_DEV mr_p mr_min(const mr_p &x, const mr_p &y) { return mr_p(mr_min(x.a, y.a), mr_min(x.b, y.b)); }
struct mr_tx2 { mr_tx a, b; };
MARAY_DEV mr_tx2 mr_texel(const MarayTex &t, const void *safe, const mr_p &x, const mr_p &y)
{
    mr_tx2 r;
    r.a = mr_texel(t, safe, x.a, y.a); r.b = mr_texel(t, safe, x.b, y.b);
    return r;
}
MARAY_DEV mr_p mr_texch(const mr_tx2 &t, unsigned sel) { return mr_p(mr_texch(t.a, sel), mr_texch(t.b, sel)); }
MARAY_DEV mr_p mr_app(const MarayTex *tex, unsigned id, const mr_p &x, const mr_p &y) { return mr_p(mr_app(tex, id, x.a, y.a), mr_app(tex, id, x.b, y.b)); }

// ---- four pixels per lane (the specialised PIXEL kernel) --------------------------------------------------------
// A wavefront owns a whole 256-pixel tile: every value is four f64 per lane (mr_d), every boolean four lane masks
// (mr_m).  The scalar unit's work per tile -- guard-bit tests, region branches, constant and y-value loads -- is
// then paid once per 256 pixels instead of once per 64, a lane's four RGB8 pixels are 12 contiguous bytes (one
// global_store_dwordx3, no cross-lane packing), and four independent dependence chains are in flight per wave.
// Element e of lane l is pixel x0 + 4 l + e (RGB8 only) or x0 + 64 e + l (when f64 planes are stored too).
// Plain structs with element-wise operators: the same generated text serves both widths.
struct mr_d {
    double a, b, c, d;
    MARAY_DEV mr_d() {}
    MARAY_DEV mr_d(double s) : a(s), b(s), c(s), d(s) {}
    MARAY_DEV mr_d(double a_, double b_, double c_, double d_) : a(a_), b(b_), c(c_), d(d_) {}
};
struct mr_m {
    mr_mask a, b, c, d;
    MARAY_DEV mr_m() {}
    MARAY_DEV mr_m(mr_mask s) : a(s), b(s), c(s), d(s) {}
    MARAY_DEV mr_m(mr_mask a_, mr_mask b_, mr_mask c_, mr_mask d_) : a(a_), b(b_), c(c_), d(d_) {}
};
#define MR_EACH1(f, v) mr_d(f((v).a), f((v).b), f((v).c), f((v).d))
MARAY_DEV mr_d operator+(const mr_d &x, const mr_d &y) { return mr_d(x.a + y.a, x.b + y.b, x.c + y.c, x.d + y.d); }
MARAY_DEV mr_d operator*(const mr_d &x, const mr_d &y) { return mr_d(x.a * y.a, x.b * y.b, x.c * y.c, x.d * y.d); }
MARAY_DEV mr_m operator&(const mr_m &x, const mr_m &y) { return mr_m(x.a & y.a, x.b & y.b, x.c & y.c, x.d & y.d); }
MARAY_DEV mr_m operator|(const mr_m &x, const mr_m &y) { return mr_m(x.a | y.a, x.b | y.b, x.c | y.c, x.d | y.d); }
MARAY_DEV mr_m operator~(const mr_m &x) { return mr_m(~x.a, ~x.b, ~x.c, ~x.d); }
MARAY_DEV bool mr_any(const mr_m &m) { return (m.a | m.b | m.c | m.d) != MR_NONE; }
MARAY_DEV mr_d mr_pos(const mr_m &m) { return mr_d(mr_pos(m.a), mr_pos(m.b), mr_pos(m.c), mr_pos(m.d)); }
MARAY_DEV mr_d mr_sel0(const mr_m &m, const mr_d &v) { return mr_d(mr_sel0(m.a, v.a), mr_sel0(m.b, v.b), mr_sel0(m.c, v.c), mr_sel0(m.d, v.d)); }
MARAY_DEV mr_d mr_neg01(const mr_m &m) { return mr_d(mr_neg01(m.a), mr_neg01(m.b), mr_neg01(m.c), mr_neg01(m.d)); }
MARAY_DEV mr_m mr_ge0(const mr_d &v) { return mr_m(mr_ge0(v.a), mr_ge0(v.b), mr_ge0(v.c), mr_ge0(v.d)); }
MARAY_DEV mr_m mr_gek(const mr_d &v, double k) { return mr_m(mr_gek(v.a, k), mr_gek(v.b, k), mr_gek(v.c, k), mr_gek(v.d, k)); }
MARAY_DEV mr_m mr_lek(const mr_d &v, double k) { return mr_m(mr_lek(v.a, k), mr_lek(v.b, k), mr_lek(v.c, k), mr_lek(v.d, k)); }
MARAY_DEV mr_m mr_ne0(const mr_d &v) { return mr_m(mr_ne0(v.a), mr_ne0(v.b), mr_ne0(v.c), mr_ne0(v.d)); }
MARAY_DEV mr_m mr_ne1(const mr_d &v) { return mr_m(mr_ne1(v.a), mr_ne1(v.b), mr_ne1(v.c), mr_ne1(v.d)); }
MARAY_DEV mr_m mr_stepsin_bounded_m(const mr_d &v) { return mr_m(mr_stepsin_bounded_m(v.a), mr_stepsin_bounded_m(v.b), mr_stepsin_bounded_m(v.c), mr_stepsin_bounded_m(v.d)); }
MARAY_DEV mr_d mr_neg(const mr_d &v) { return MR_EACH1(mr_neg, v); }
MARAY_DEV mr_d mr_abs(const mr_d &v) { return MR_EACH1(mr_abs, v); }
MARAY_DEV mr_d mr_recip(const mr_d &v) { return MR_EACH1(mr_recip, v); }
MARAY_DEV mr_d mr_sqrt(const mr_d &v)           // one question for the four elements: their Newton chains interleave
{
    const double least = __builtin_fmin(__builtin_fmin(v.a, v.b), __builtin_fmin(v.c, v.d));      // (NaNs drop out: they need no scaling)
    if (__builtin_expect(mr_ballot(least < 0x1p-767) != 0ull, 0)) return MR_EACH1(mr_sqrt, v);     // some element is tiny, zero or negative: ask per element
    return MR_EACH1(mr_sqrt_unscaled, v);
}
MARAY_DEV mr_d mr_sin(const mr_d &v) { return MR_EACH1(mr_sin, v); }
MARAY_DEV mr_d mr_sin_bounded(const mr_d &v) { return MR_EACH1(mr_sin_bounded, v); }
MARAY_DEV mr_d mr_exp(const mr_d &v) { return MR_EACH1(mr_exp, v); }
MARAY_DEV mr_d mr_ln(const mr_d &v) { return MR_EACH1(mr_ln, v); }
MARAY_DEV mr_d mr_stepsin(const mr_d &v) { return MR_EACH1(mr_stepsin, v); }
MARAY_DEV mr_d mr_stepsin_fast(const mr_d &v, float *defer)
{
    return mr_d(mr_stepsin_fast(v.a, defer), mr_stepsin_fast(v.b, defer), mr_stepsin_fast(v.c, defer), mr_stepsin_fast(v.d, defer));
}
MARAY_DEV mr_d mr_max(const mr_d &x, const mr_d &y) { return mr_d(mr_max(x.a, y.a), mr_max(x.b, y.b), mr_max(x.c, y.c), mr_max(x.d, y.d)); }
MARAY_DEV mr_d mr_min(const mr_d &x, const mr_d &y) { return mr_d(mr_min(x.a, y.a), mr_min(x.b, y.b), mr_min(x.c, y.c), mr_min(x.d, y.d)); }
struct mr_tx4 { mr_tx a, b, c, d; };
MARAY_DEV mr_tx4 mr_texel(const MarayTex &t, const void *safe, const mr_d &x, const mr_d &y)
{
    mr_tx4 r;
    r.a = mr_texel(t, safe, x.a, y.a); r.b = mr_texel(t, safe, x.b, y.b); r.c = mr_texel(t, safe, x.c, y.c); r.d = mr_texel(t, safe, x.d, y.d);
    return r;
}
MARAY_DEV mr_d mr_texch(const mr_tx4 &t, unsigned sel) { return mr_d(mr_texch(t.a, sel), mr_texch(t.b, sel), mr_texch(t.c, sel), mr_texch(t.d, sel)); }
MARAY_DEV mr_d mr_app(const MarayTex *tex, unsigned id, const mr_d &x, const mr_d &y)
{
    return mr_d(mr_app(tex, id, x.a, y.a), mr_app(tex, id, x.b, y.b), mr_app(tex, id, x.c, y.c), mr_app(tex, id, x.d, y.d));
}

#endif
