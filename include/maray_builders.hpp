// maray_builders.hpp — header-only C++17 mirror of the reference's scene-authoring
// functions, enough to write every configuration of §8(d) as
// a real `.maray` file.  Each function restates the Rust function of the same
// name; expressions are immutable shared trees, `save()` writes the bincode
// encoding of `([u32;2], [Expr;3])` in the current tag numbering
// (src/lib.rs:101-149, :1216-1224), which `maray_scene_open` / `maray::open` read.
//
//   constructors / algebra   src/lib.rs:836-966
//   chess, set_unit_square   src/lib.rs:969-980
//   p2_*                     src/lib.rs:983-1075
//   quad_*, triangles, uv    src/lib.rs:1078-1131
//   Grid2::cell              src/grid.rs:10-32
//   Sd2                      src/sd.rs:6-48
//   textures ids             src/textures.rs:14-23
//   Expr::subst2 / scale     src/lib.rs:709-735, 804-806
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace maray_build {

enum Tag : uint32_t {
    Arc = 0, X, Y, Tau, E, Var, Nat, Neg, Abs, Recip, Sqrt, Step, Sin, Exp, Ln, Add, Mul, Max, Min, Let, Decor, App,
    Encoded = 0xFFFFFFFFu       // not a variant of the reference's Expr: an expression already in its bincode form (see encoded())
};

struct Node;
using Expr = std::shared_ptr<const Node>;
using Point2 = std::array<Expr, 2>;
using Point3 = std::array<Expr, 3>;
using Point4 = std::array<Expr, 4>;
using Color = std::array<Expr, 3>;

struct Node {
    Tag tag;
    uint64_t u = 0;                                 // Var id / Nat value / App id
    Expr a, b;
    std::vector<std::pair<uint64_t, Expr>> vars;    // Let
    std::vector<uint8_t> raw;                       // Encoded
};

inline Expr mk(Tag t, Expr a = nullptr, Expr b = nullptr, uint64_t u = 0)
{
    auto n = std::make_shared<Node>();
    n->tag = t; n->a = std::move(a); n->b = std::move(b); n->u = u;
    return n;
}

// ---- constructors (src/lib.rs:836-966) ---------------------------------------------------
inline Expr app(uint32_t id, Expr a, Expr b) { return mk(App, a, b, id); }
inline Expr x() { return mk(X); }
inline Expr y() { return mk(Y); }
inline Expr var_id(uint64_t id) { return mk(Var, nullptr, nullptr, id); }
// var (src/lib.rs:845-850): the id is the name's hash -- Rust's `str::hash` feeds the bytes and a 0xff terminator to
// fnv::FnvHasher (FNV-1a, 64 bit)
inline Expr var(const std::string &name)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (unsigned char c : name) { h ^= c; h *= 0x100000001b3ull; }
    h ^= 0xffu; h *= 0x100000001b3ull;
    return var_id(h);
}
inline Expr tau() { return mk(Tau); }
inline Expr e() { return mk(E); }
inline Expr nat(uint64_t a) { return mk(Nat, nullptr, nullptr, a); }
inline Expr neg(Expr a) { return mk(Neg, a); }
inline Expr abs(Expr a) { return mk(Abs, a); }
inline Expr recip(Expr a) { return mk(Recip, a); }
inline Expr sqrt(Expr a) { return mk(Sqrt, a); }
inline Expr step(Expr a) { return mk(Step, a); }
inline Expr sin(Expr a) { return mk(Sin, a); }
inline Expr exp(Expr a) { return mk(Exp, a); }
inline Expr ln(Expr a) { return mk(Ln, a); }
inline Expr max(Expr a, Expr b) { return mk(Max, a, b); }
inline Expr min(Expr a, Expr b) { return mk(Min, a, b); }
inline Expr add(Expr a, Expr b) { return mk(Add, a, b); }
inline Expr mul(Expr a, Expr b) { return mk(Mul, a, b); }
inline Expr sub(Expr a, Expr b) { return add(a, neg(b)); }
inline Expr div(Expr a, Expr b) { return mul(a, recip(b)); }
// the operators of src/lib.rs:151-194 (Expr with Expr, Expr with a natural number, unary minus); found through Node's namespace
inline Expr operator/(const Expr &a, const Expr &b) { return div(a, b); }
inline Expr operator/(const Expr &a, uint64_t b) { return div(a, nat(b)); }
inline Expr operator*(const Expr &a, const Expr &b) { return mul(a, b); }
inline Expr operator*(const Expr &a, uint64_t b) { return mul(a, nat(b)); }
inline Expr operator+(const Expr &a, const Expr &b) { return add(a, b); }
inline Expr operator+(const Expr &a, uint64_t b) { return add(a, nat(b)); }
inline Expr operator-(const Expr &a, const Expr &b) { return sub(a, b); }
inline Expr operator-(const Expr &a, uint64_t b) { return sub(a, nat(b)); }
inline Expr operator-(const Expr &a) { return neg(a); }
inline Expr let_(std::vector<std::pair<uint64_t, Expr>> vars, Expr body)
{
    auto n = std::make_shared<Node>();
    n->tag = Let; n->vars = std::move(vars); n->a = std::move(body);
    return n;
}
// An expression that exists as bytes: what `Expr::simplify(mem).compress(mem)` returns when the two passes are done by
// libmaray_hip (maray_scene_simplify_ex / maray_scene_compress work on encoded scenes).  `save` writes the bytes as they
// are, so `mul(encoded(shape_bytes), nat(255))` is examples/chess.rs:43-45.  Opaque to subst2 / scale / translate.
inline Expr encoded(std::vector<uint8_t> bincode_of_one_expr)
{
    auto n = std::make_shared<Node>();
    n->tag = Encoded; n->raw = std::move(bincode_of_one_expr);
    return n;
}
inline Expr pi() { return div(tau(), nat(2)); }
inline Expr rad_45() { return div(tau(), nat(8)); }
inline Expr rad_90() { return div(tau(), nat(4)); }
inline Expr half() { return div(nat(1), nat(2)); }
inline Expr square(Expr a) { return mul(a, a); }
inline Expr lerp(Expr a, Expr b, Expr t) { return add(a, mul(sub(b, a), t)); }
inline Expr set_inv(Expr a) { return sub(nat(1), a); }
inline Expr set_and(Expr a, Expr b) { return min(a, b); }
inline Expr set_or(Expr a, Expr b) { return max(a, b); }
inline Expr set_xor(Expr a, Expr b) { return set_or(set_and(a, set_inv(b)), set_and(b, set_inv(a))); }
inline Expr step_at(Expr a, Expr x_) { return step(sub(x_, a)); }
inline Expr step_pos(Expr a) { return set_inv(step(neg(a))); }
inline Expr step_pos_at(Expr a, Expr x_) { return step_pos(sub(x_, a)); }
inline Expr pos(Expr cond, Expr a, Expr b) { return lerp(b, a, step_pos(cond)); }
inline Expr range(Expr a, Expr b, Expr x_) { return mul(step_at(a, x_), set_inv(step_at(b, x_))); }
inline Expr range_incl(Expr a, Expr b, Expr x_) { return mul(step_at(a, x_), set_inv(step_pos_at(b, x_))); }
inline Expr clamp(Expr a, Expr b, Expr x_) { return pos(sub(x_, a), pos(sub(x_, b), b, x_), a); }
inline Expr clamp_unit(Expr x_) { return clamp(nat(0), nat(1), x_); }
inline Expr clamp_u8(Expr x_) { return clamp(nat(0), nat(255), x_); }
inline Expr ge(Expr a, Expr b) { return step(sub(a, b)); }
inline Expr gt(Expr a, Expr b) { return step_pos(sub(a, b)); }
inline Expr le(Expr a, Expr b) { return set_inv(gt(a, b)); }
inline Expr lt(Expr a, Expr b) { return set_inv(ge(a, b)); }
inline Expr eq(Expr a, Expr b) { return set_and(ge(a, b), le(a, b)); }
inline Expr cos(Expr a) { return sin(add(a, rad_90())); }
inline Expr unit_to_rad(Expr a) { return mul(a, tau()); }
inline Expr rad_to_unit(Expr a) { return div(a, tau()); }

inline Expr chess(uint64_t n)   // src/lib.rs:969-973
{
    Expr sx = step(sin(mul(mul(div(nat(n), nat(2)), tau()), x())));
    Expr sy = step(sin(mul(mul(div(nat(n), nat(2)), tau()), y())));
    return set_xor(sx, sy);
}
inline Expr set_unit_square(Expr f)   // src/lib.rs:975-980
{
    return set_and(set_and(range(nat(0), nat(1), x()), range(nat(0), nat(1), y())), f);
}

// ---- 2-D points (src/lib.rs:983-1075) ---------------------------------------------------
inline Point2 p2_neg(Point2 a) { return {neg(a[0]), neg(a[1])}; }
inline Point2 p2_abs(Point2 a) { return {abs(a[0]), abs(a[1])}; }
inline Point2 p2_add(Point2 a, Point2 b) { return {add(a[0], b[0]), add(a[1], b[1])}; }
inline Point2 p2_sub(Point2 a, Point2 b) { return {sub(a[0], b[0]), sub(a[1], b[1])}; }
inline Point2 p2_mul(Point2 a, Point2 b) { return {mul(a[0], b[0]), mul(a[1], b[1])}; }
inline Point2 p2_div(Point2 a, Point2 b) { return {div(a[0], b[0]), div(a[1], b[1])}; }
inline Point2 p2_max(Point2 a, Point2 b) { return {max(a[0], b[0]), max(a[1], b[1])}; }
inline Point2 p2_scale(Point2 a, Expr b) { return p2_mul(a, {b, b}); }
inline Expr p2_dot(Point2 a, Point2 b) { return add(mul(a[0], b[0]), mul(a[1], b[1])); }
inline Expr p2_len(Point2 a) { return sqrt(p2_dot(a, a)); }
inline Point2 p2_lerp(Point2 a, Point2 b, Expr t) { return {lerp(a[0], b[0], t), lerp(a[1], b[1], t)}; }
inline Point2 p2_pos(Expr cond, Point2 a, Point2 b) { return p2_lerp(b, a, step_pos(cond)); }
inline Point2 p2_circle(Expr ang) { return {cos(ang), sin(ang)}; }
inline Point2 p2_qbez(Point2 a, Point2 b, Point2 c, Expr t) { return p2_lerp(p2_lerp(a, b, t), p2_lerp(b, c, t), t); }
inline Point2 p2_cbez(Point2 a, Point2 b, Point2 c, Point2 d, Expr t) { return p2_lerp(p2_qbez(a, b, c, t), p2_qbez(b, c, d, t), t); }   // :1061-1065
inline Point2 p2_spiral(Expr ang) { return p2_scale(p2_circle(ang), rad_to_unit(ang)); }                                               // :1035-1037
inline Point4 p4_same(Expr v) { return {v, v, v, v}; }                                                                                   // :1141-1151
inline Point2 p4_xy(const Point4 &p) { return {p[0], p[1]}; }
inline Point2 p4_zw(const Point4 &p) { return {p[2], p[3]}; }

// ---- quads, triangles, uv (src/lib.rs:1078-1131) ------------------------------------------------
inline Point3 to_barycentric(const std::array<Point2, 3> &tri, Point2 p)
{
    Expr px = p[0], py = p[1];
    Expr x1 = tri[0][0], y1 = tri[0][1], x2 = tri[1][0], y2 = tri[1][1], x3 = tri[2][0], y3 = tri[2][1];
    auto den = [&] { return add(mul(sub(y2, y3), sub(x1, x3)), mul(sub(x3, x2), sub(y1, y3))); };
    Expr l1 = div(add(mul(sub(y2, y3), sub(px, x3)), mul(sub(x3, x2), sub(py, y3))), den());
    Expr l2 = div(add(mul(sub(y3, y1), sub(px, x3)), mul(sub(x1, x3), sub(py, y3))), den());
    Expr l3 = sub(sub(nat(1), l1), l2);
    return {l1, l2, l3};
}
inline Expr inside_triangle(const std::array<Point2, 3> &tri, Point2 p)
{
    Point3 b = to_barycentric(tri, p);
    return set_and(set_and(step(b[0]), step(b[1])), step(b[2]));
}
inline Point2 to_uv(const std::array<Point2, 3> &tri, const std::array<Point2, 3> &uv, Point2 p)
{
    Point3 b = to_barycentric(tri, p);
    return p2_add(p2_add(p2_scale(uv[0], b[0]), p2_scale(uv[1], b[1])), p2_scale(uv[2], b[2]));
}
inline Point2 quad_pos(const std::array<Point2, 4> &q, Point2 uv)
{
    return p2_lerp(p2_lerp(q[0], q[1], uv[0]), p2_lerp(q[2], q[3], uv[0]), uv[1]);
}
struct TriPair { std::array<Point2, 3> tri[2], uv[2]; };
inline TriPair quad_to_tri(const std::array<Point2, 4> &q, const std::array<Point2, 4> &uv)
{
    TriPair r;
    r.tri[0] = {q[0], q[1], q[2]}; r.uv[0] = {uv[0], uv[1], uv[2]};
    r.tri[1] = {q[1], q[2], q[3]}; r.uv[1] = {uv[1], uv[2], uv[3]};
    return r;
}

// Grid2::cell (src/grid.rs:10-32)
struct Grid2 {
    uint64_t nx, ny;
    std::pair<std::array<Point2, 4>, std::array<Point2, 4>> cell(uint64_t i, uint64_t j, const std::array<Point2, 4> &quad) const
    {
        Expr w = nat(nx), h = nat(ny);
        Expr fx = div(nat(i), w), fy = div(nat(j), h), gx = div(nat(i + 1), w), gy = div(nat(j + 1), h);
        std::array<Point2, 4> uv = {Point2{fx, fy}, Point2{gx, fy}, Point2{fx, gy}, Point2{gx, gy}};
        return {{quad_pos(quad, uv[0]), quad_pos(quad, uv[1]), quad_pos(quad, uv[2]), quad_pos(quad, uv[3])}, uv};
    }
};

// Sd2 (src/sd.rs:6-48): signed distance of a circle / box / rounded box as an Expr
inline Expr sd_circle(Expr r) { return sub(p2_len({x(), y()}), r); }
inline Expr sd_box(Point2 half_size)
{
    Point2 d = p2_sub(p2_abs({x(), y()}), half_size);
    return add(p2_len(p2_max(d, {nat(0), nat(0)})), min(max(d[0], d[1]), nat(0)));
}

inline Expr sd_rounded_box(Point2 b, std::array<Expr, 4> r)
{
    Point2 rxy = p2_pos(x(), {r[0], r[1]}, {r[2], r[3]});
    Expr rx = pos(y(), rxy[0], rxy[1]);
    Point2 q = p2_add(p2_sub(p2_abs({x(), y()}), b), {rx, rx});
    return sub(add(min(max(q[0], q[1]), nat(0)), p2_len(p2_max(q, {nat(0), nat(0)}))), rx);
}
inline Expr sd_inside(Expr sd) { return step(neg(sd)); }    // Sd2::inside (src/sd.rs:51-53)
inline Expr sd_outside(Expr sd) { return step(sd); }        // Sd2::outside (:56-58)

// textures ids (src/textures.rs:14-23)
inline uint32_t channel(uint32_t img, uint32_t ch) { return img * 5u + ch; }
inline uint32_t image_width(uint32_t img) { return img * 5u + 3u; }
inline uint32_t image_height(uint32_t img) { return img * 5u + 4u; }

// Expr::subst2 (src/lib.rs:709-735): does not descend into Let
inline Expr subst2(const Expr &ex, const Point2 &p)
{
    switch (ex->tag) {
    case Arc: return subst2(ex->a, p);
    case X: return p[0];
    case Y: return p[1];
    case Tau: case E: case Var: case Nat: case Let: case Encoded: return ex;
    case App: return app((uint32_t)ex->u, subst2(ex->a, p), subst2(ex->b, p));
    default: return mk(ex->tag, subst2(ex->a, p), ex->b ? subst2(ex->b, p) : nullptr, ex->u);
    }
}
inline Expr scale(const Expr &ex, Point2 s) { return subst2(ex, p2_div({x(), y()}, s)); }     // :804-806
inline Expr translate(const Expr &ex, Point2 off) { return subst2(ex, p2_sub({x(), y()}, off)); }   // :799-801
inline Expr scale_at(const Expr &ex, Point2 off, Point2 s) { return translate(scale(translate(ex, p2_neg(off)), s), off); }   // :809-811
inline Expr rotate(const Expr &ex, Expr rad)   // :814-821
{
    const Expr sn = sin(rad), cs = cos(rad);
    const Point2 id = {x(), y()};
    return subst2(ex, {p2_dot({cs, neg(sn)}, id), p2_dot({sn, cs}, id)});
}
inline Expr rotate_at(const Expr &ex, Point2 off, Expr rad) { return translate(rotate(translate(ex, p2_neg(off)), rad), off); }   // :824-826
inline Point2 p2_subst(const Point2 &p, const Point2 &off) { return {subst2(p[0], off), subst2(p[1], off)}; }                    // :1067-1070
// from_barycentric (:1127-1138)
inline Point2 from_barycentric(const std::array<Point2, 3> &tri, const Point3 &l)
{
    return {add(add(mul(l[0], tri[0][0]), mul(l[1], tri[1][0])), mul(l[2], tri[2][0])),
            add(add(mul(l[0], tri[0][1]), mul(l[1], tri[1][1])), mul(l[2], tri[2][1]))};
}

// Expr::var_range (:738-764): the half-open range of variable ids an expression mentions or defines; empty = {0, 0}
inline std::array<uint64_t, 2> var_range_union(std::array<uint64_t, 2> a, std::array<uint64_t, 2> b)   // :830-834
{
    if (a[1] - a[0] == 0) return b;
    if (b[1] - b[0] == 0) return a;
    return {std::min(a[0], b[0]), std::max(a[1], b[1])};
}
inline std::array<uint64_t, 2> var_range(const Expr &ex)
{
    switch (ex->tag) {
    case Arc: case Decor: return var_range(ex->a);
    case X: case Y: case Tau: case E: case Nat: case Encoded: return {0, 0};
    case Var: return {ex->u, ex->u + 1};
    case Let: {
        std::array<uint64_t, 2> r = var_range(ex->a);            // (the body)
        for (const auto &v : ex->vars) { r = var_range_union(r, {v.first, v.first + 1}); r = var_range_union(r, var_range(v.second)); }
        return r;
    }
    default: return ex->b ? var_range_union(var_range(ex->a), var_range(ex->b)) : var_range(ex->a);
    }
}
// Expr::var_offset (:767-796): every variable id, used or defined, moved by off
inline Expr var_offset(const Expr &ex, int64_t off)
{
    switch (ex->tag) {
    case Arc: return var_offset(ex->a, off);
    case X: case Y: case Tau: case E: case Nat: case Encoded: return ex;
    case Var: return var_id((uint64_t)((int64_t)ex->u + off));
    case Let: {
        std::vector<std::pair<uint64_t, Expr>> vars;
        for (const auto &v : ex->vars) vars.emplace_back((uint64_t)((int64_t)v.first + off), var_offset(v.second, off));
        return let_(std::move(vars), var_offset(ex->a, off));
    }
    case App: return app((uint32_t)ex->u, var_offset(ex->a, off), var_offset(ex->b, off));
    default: return mk(ex->tag, var_offset(ex->a, off), ex->b ? var_offset(ex->b, off) : nullptr, ex->u);
    }
}

// ---- save (src/lib.rs:1216-1224) ---------------------------------------------------------------
inline void encode_expr(const Expr &ex, std::vector<uint8_t> &out)
{
    auto u32 = [&](uint32_t v) { uint8_t b[4]; memcpy(b, &v, 4); out.insert(out.end(), b, b + 4); };
    auto u64 = [&](uint64_t v) { uint8_t b[8]; memcpy(b, &v, 8); out.insert(out.end(), b, b + 8); };
    if (ex->tag == Encoded) { out.insert(out.end(), ex->raw.begin(), ex->raw.end()); return; }
    u32(ex->tag);
    switch (ex->tag) {
    case X: case Y: case Tau: case E: break;
    case Var: case Nat: u64(ex->u); break;
    case Let:
        u64(ex->vars.size());
        for (auto &v : ex->vars) { u64(v.first); encode_expr(v.second, out); }
        encode_expr(ex->a, out);
        break;
    case App: u32((uint32_t)ex->u); encode_expr(ex->a, out); encode_expr(ex->b, out); break;
    case Decor: encode_expr(ex->a, out); u64(0); break;
    default:
        encode_expr(ex->a, out);
        if (ex->b) encode_expr(ex->b, out);
    }
}
inline std::vector<uint8_t> encode(uint32_t w, uint32_t h, const Color &c)
{
    std::vector<uint8_t> out;
    uint8_t b[8]; memcpy(b, &w, 4); memcpy(b + 4, &h, 4);
    out.insert(out.end(), b, b + 8);
    for (const Expr &e_ : c) encode_expr(e_, out);
    return out;
}
inline bool save(const std::string &path, uint32_t w, uint32_t h, const Color &c)
{
    std::vector<uint8_t> bytes = encode(w, h, c);
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    fclose(f);
    return ok;
}

}   // namespace maray_build
