// simplify.cpp — `Expr::simplify` of the reference, restated on the arena data model (product code, host side).
//
// SURVEY.md §8(f) N4, first half: the authoring-time simplifier (`src/lib.rs:601-604` = `constant_reduction::run`
// at the root, `src/constant_reduction.rs:9-185`, then `simplify::run`, `src/simplify.rs:129-327`, which recurses
// through `Expr::simplify` again).  It is a fixed list of rewrite rules over trees of naturals, not a normaliser: the
// restatement keeps the reference's rules, their order and their quirks (prime factors tried up to 17 only, a rule
// that fires for `Neg` on the left but not on the right, ...), because its results are data -- `.maray` files are
// written after it -- and the reference's own tests (`src/lib.rs:1288-1516`, `:1694-1720`) pin them.
// Not on the render path: nothing here runs per pixel.  `compress` (the Let-introducing pass): compress.cpp.
#include <cstdint>
#include <utility>

#include "expr.hpp"
#include "maray_hip.h"

namespace maray {

namespace {

struct Simp {
    Scene &s;
    bool merge_divisors = false;           // MARAY_SIMPLIFY_MERGE_DIVISORS (maray_hip.h): `(a/p) * 1/q` becomes `a / (p*q)`
    explicit Simp(Scene &s_) : s(s_) {}

    // ---- builders (src/lib.rs:836-870: sub = a + -b, div = a * 1/b) ----
    int32_t mk(uint8_t tag, int32_t a = -1, int32_t b = -1, uint64_t u = 0) { Node n; n.tag = tag; n.a = a; n.b = b; n.u = u; return s.add(n); }
    int32_t nat(uint64_t n) { return mk(T_NAT, -1, -1, n); }
    int32_t neg(int32_t a) { return mk(T_NEG, a); }
    int32_t recip(int32_t a) { return mk(T_RECIP, a); }
    int32_t add(int32_t a, int32_t b) { return mk(T_ADD, a, b); }
    int32_t mul(int32_t a, int32_t b) { return mk(T_MUL, a, b); }
    int32_t sub(int32_t a, int32_t b) { return add(a, neg(b)); }
    int32_t div(int32_t a, int32_t b) { return mul(a, recip(b)); }

    const Node &N(int32_t e) const { return s.nodes[e]; }
    uint8_t tag(int32_t e) const { return s.nodes[e].tag; }

    bool equal(int32_t x, int32_t y) const {                   // derived PartialEq of Expr
        if (x == y) return true;
        const Node &p = N(x), &q = N(y);
        if (p.tag != q.tag) return false;
        switch (p.tag) {
        case T_X: case T_Y: case T_TAU: case T_E: return true;
        case T_VAR: case T_NAT: return p.u == q.u;
        case T_APP: return p.app == q.app && equal(p.a, q.a) && equal(p.b, q.b);
        case T_LET: {
            const Ctx &c = s.ctxs[p.ctx], &d = s.ctxs[q.ctx];
            if (c.ids != d.ids || c.defs.size() != d.defs.size()) return false;
            for (size_t i = 0; i < c.defs.size(); i++) if (!equal(c.defs[i], d.defs[i])) return false;
            return equal(p.a, q.a);
        }
        case T_DECOR: {
            const auto &t = s.toklists[p.toks], &u = s.toklists[q.toks];
            if (t.size() != u.size()) return false;
            for (size_t i = 0; i < t.size(); i++) {
                if (t[i].kind != u[i].kind || t[i].str != u[i].str) return false;
                if (t[i].kind == 0 && !equal(t[i].expr, u[i].expr)) return false;
            }
            return equal(p.a, q.a);
        }
        default:
            if (is_binary(p.tag)) return equal(p.a, q.a) && equal(p.b, q.b);
            return equal(p.a, q.a);                            // unary, Arc
        }
    }

    // ---- getters (src/lib.rs:404-530) ----
    bool get_nat(int32_t e, uint64_t &n) const { if (tag(e) != T_NAT) return false; n = N(e).u; return true; }
    int32_t get_neg(int32_t e) const { return tag(e) == T_NEG ? N(e).a : -1; }
    int32_t get_recip(int32_t e) const { return tag(e) == T_RECIP ? N(e).a : -1; }
    bool get_add(int32_t e, int32_t &a, int32_t &b) const { if (tag(e) != T_ADD) return false; a = N(e).a; b = N(e).b; return true; }
    bool get_mul(int32_t e, int32_t &a, int32_t &b) const { if (tag(e) != T_MUL) return false; a = N(e).a; b = N(e).b; return true; }
    bool get_sub(int32_t e, int32_t &a, int32_t &b) const {   // a + -(b)
        if (tag(e) != T_ADD || tag(N(e).b) != T_NEG) return false;
        a = N(e).a; b = N(N(e).b).a; return true;
    }
    bool get_div(int32_t e, int32_t &a, int32_t &b) const {   // a * 1/(b)
        if (tag(e) != T_MUL || tag(N(e).b) != T_RECIP) return false;
        a = N(e).a; b = N(N(e).b).a; return true;
    }

    // ---- signed rational cases (src/simplify.rs:4-127) ----
    enum Kind { NONE, NAT, DIV };
    struct Case { bool pos; Kind kind; uint64_t a, b; };

    Case from_expr(int32_t e) const {
        uint64_t n, m;
        int32_t p, q;
        switch (tag(e)) {
        case T_NAT: return {true, NAT, N(e).u, 0};
        case T_NEG: {
            const int32_t a = N(e).a;
            if (get_nat(a, n)) return {false, NAT, n, 0};
            if (tag(a) == T_RECIP) return get_nat(N(a).a, n) ? Case{false, DIV, 1, n} : Case{false, NONE, 0, 0};
            if (get_mul(a, p, q) && get_nat(p, n) && tag(q) == T_RECIP)
                return get_nat(N(q).a, m) ? Case{false, DIV, n, m} : Case{false, NONE, 0, 0};
            return {false, NONE, 0, 0};
        }
        case T_RECIP: return get_nat(N(e).a, n) ? Case{true, DIV, 1, n} : Case{true, NONE, 0, 0};
        case T_MUL:
            if (get_nat(N(e).a, n) && tag(N(e).b) == T_RECIP) {
                if (!get_nat(N(N(e).b).a, m)) return {true, NONE, 0, 0};
                if (n == m && m != 0) return {true, NAT, 1, 0};
                for (uint64_t pr : {2u, 3u, 5u, 7u, 11u, 13u, 17u}) if (n % pr == 0 && m % pr == 0) { n /= pr; m /= pr; }
                return {true, DIV, n, m};
            }
            return {true, NONE, 0, 0};
        default: return {true, NONE, 0, 0};
        }
    }
    int32_t to_expr(const Case &c) {                           // None -> -1
        int32_t r;
        if (c.kind == NONE) return -1;
        if (c.kind == NAT) r = nat(c.a);
        else if (c.a == 1) r = recip(nat(c.b));
        else if (c.b == 1) r = nat(c.a);
        else r = div(nat(c.a), nat(c.b));
        return c.pos ? r : neg(r);
    }
    int32_t normalize(int32_t e) { return to_expr(from_expr(e)); }   // the reference unwraps: callers pass rationals only

    int32_t add_nat(bool sa, uint64_t a, bool sb, uint64_t b) {
        if (sa && sb) return nat(a + b);
        if (!sa && !sb) return neg(nat(a + b));
        if (!sa) { std::swap(a, b); }                           // now: +a, -b
        return a >= b ? nat(a - b) : neg(nat(b - a));
    }
    int32_t add_div(bool sa, uint64_t a0, uint64_t a1, bool sb, uint64_t b0, uint64_t b1) {
        if (sa && sb)
            return a1 == b1 ? div(nat(a0 + b0), nat(a1)) : simplify(div(nat(a0 * b1 + a1 * b0), nat(a1 * b1)));
        if (!sa && !sb)
            return simplify(neg(add(div(nat(a0), nat(a1)), div(nat(b0), nat(b1)))));
        if (!sa) { std::swap(a0, b0); std::swap(a1, b1); }      // now: +a0/a1, -b0/b1
        if (a1 == b1) return simplify(a0 >= b0 ? div(nat(a0 - b0), nat(a1)) : neg(div(nat(b0 - a0), nat(a1))));
        const uint64_t d1 = a0 * b1, d2 = a1 * b0;
        return simplify(d1 >= d2 ? div(nat(d1 - d2), nat(a1 * b1)) : neg(div(nat(d2 - d1), nat(a1 * b1))));
    }
    int32_t mul_nat(bool sa, uint64_t a, bool sb, uint64_t b) { return sa == sb ? nat(a * b) : neg(nat(a * b)); }
    int32_t mul_div(bool sa, uint64_t a0, uint64_t a1, bool sb, uint64_t b0, uint64_t b1) {
        const int32_t r = div(nat(a0 * b0), nat(a1 * b1));
        return sa == sb ? r : neg(r);
    }

    // ---- constant_reduction::run: common prime factors cancelled in a few fixed shapes, at the root only ----
    // Returns e or a copy with changed naturals (the reference mutates in place; nodes may be shared here).
    static void cancel(std::initializer_list<uint64_t *> v) {
        static const unsigned primes[] = {2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 5, 5, 5, 7, 7, 7, 11, 11, 13, 17};
        for (unsigned p : primes) {
            bool all = true;
            for (uint64_t *x : v) all = all && (*x % p == 0);
            if (all) for (uint64_t *x : v) *x /= p;
        }
    }
    int32_t constant_reduction(int32_t e) {
        int32_t a, b;
        if (!get_mul(e, a, b)) return e;
        uint64_t k, m;
        int32_t a1, a2, b1, b2, p, q;
        // a1/k * k
        if (get_div(a, a1, a2) && get_nat(a2, m) && get_nat(b, k)) {
            cancel({&m, &k});
            a = div(a1, nat(m)); b = nat(k);
        }
        // (k * a2) / k  and  (a1 * k) / k        (b = 1/k)
        if (get_mul(a, a1, a2) && tag(b) == T_RECIP && get_nat(N(b).a, k)) {
            if (get_nat(a1, m)) { cancel({&m, &k}); a1 = nat(m); }
            if (get_nat(a2, m)) { cancel({&m, &k}); a2 = nat(m); }
            a = mul(a1, a2); b = recip(nat(k));
        }
        // k * (b11/k1 - b21/k2)  and  k * (b11/k1 + b21/k2)
        for (int pass = 0; pass < 2; pass++) {
            uint64_t k1, k2;
            int32_t n1, d1, n2, d2;
            const bool shape = pass == 0 ? get_sub(b, b1, b2) : get_add(b, b1, b2);
            if (get_nat(a, k) && shape && get_div(b1, n1, d1) && get_div(b2, n2, d2) && get_nat(d1, k1) && get_nat(d2, k2)) {
                cancel({&k, &k1, &k2});
                a = nat(k);
                b = pass == 0 ? sub(div(n1, nat(k1)), div(n2, nat(k2))) : add(div(n1, nat(k1)), div(n2, nat(k2)));
            }
        }
        // k * (b11 * (p1/k1 - p2/k2) - b21/k3)
        if (get_nat(a, k) && get_sub(b, b1, b2)) {
            int32_t f, inner, n3, d3, i1, i2, pn1, pd1, pn2, pd2;
            uint64_t k1, k2, k3;
            if (get_mul(b1, f, inner) && get_div(b2, n3, d3) && get_sub(inner, i1, i2) && get_nat(d3, k3) &&
                get_div(i1, pn1, pd1) && get_div(i2, pn2, pd2) && get_nat(pd1, k1) && get_nat(pd2, k2)) {
                cancel({&k, &k1, &k2, &k3});
                a = nat(k);
                b = sub(mul(f, sub(div(pn1, nat(k1)), div(pn2, nat(k2)))), div(n3, nat(k3)));
            }
        }
        // k * (b11 * (p1/k1 - p2/k2) - b21 * (q1/k3 - q2/k4))
        if (get_nat(a, k) && get_sub(b, b1, b2)) {
            int32_t f1, in1, f2, in2, i1, i2, j1, j2, pn1, pd1, pn2, pd2, qn1, qd1, qn2, qd2;
            uint64_t k1, k2, k3, k4;
            if (get_mul(b1, f1, in1) && get_mul(b2, f2, in2) && get_sub(in1, i1, i2) && get_sub(in2, j1, j2) &&
                get_div(i1, pn1, pd1) && get_div(i2, pn2, pd2) && get_div(j1, qn1, qd1) && get_div(j2, qn2, qd2) &&
                get_nat(pd1, k1) && get_nat(pd2, k2) && get_nat(qd1, k3) && get_nat(qd2, k4)) {
                cancel({&k, &k1, &k2, &k3, &k4});
                a = nat(k);
                b = sub(mul(f1, sub(div(pn1, nat(k1)), div(pn2, nat(k2)))), mul(f2, sub(div(qn1, nat(k3)), div(qn2, nat(k4)))));
            }
        }
        (void)p; (void)q;
        return (a == N(e).a && b == N(e).b) ? e : mul(a, b);
    }

    // The rules are not confluent: `(a/b) * 1/c` becomes `(a * 1/c) / b`, whose numerator is the quotient a/c again,
    // and so on for ever unless a is a natural (the reference overflows its stack on such input; examples/chess.rs
    // does not produce it).  Here that is an error, not a crash.
    int depth = 0;
    bool runaway = false;                  // reported by scene_simplify once the recursion has unwound by itself
    int32_t simplify(int32_t e) {
        if (runaway) return e;
        if (++depth > 50000) { runaway = true; depth--; return e; }
        const int32_t r = run(constant_reduction(e));
        depth--;
        return r;
    }

    // ---- simplify::run (src/simplify.rs:129-327) ----
    int32_t unary(uint8_t t, int32_t e, int32_t a) { return a == N(e).a ? e : mk(t, a); }
    int32_t binary(uint8_t t, int32_t e, int32_t a, int32_t b) { return (a == N(e).a && b == N(e).b) ? e : mk(t, a, b); }

    int32_t run(int32_t e) {
        uint64_t n, m;
        int32_t p, q, r, t;
        switch (tag(e)) {
        case T_ARC: {
            const int32_t a = simplify(N(e).a);
            return equal(a, N(e).a) ? e : mk(T_ARC, a);
        }
        case T_X: case T_Y: case T_TAU: case T_E: case T_NAT: case T_VAR: return e;
        case T_NEG: {
            const int32_t a = simplify(N(e).a);
            if (get_neg(a) >= 0) return simplify(get_neg(a));
            if (get_nat(a, n) && n == 0) return nat(0);
            if (get_sub(a, p, q)) return sub(q, p);
            return unary(T_NEG, e, a);
        }
        case T_ABS: return unary(T_ABS, e, simplify(N(e).a));
        case T_RECIP: {
            const int32_t a = simplify(N(e).a);
            if (get_div(a, p, q)) return simplify(div(q, p));
            if (get_neg(a) >= 0) return simplify(neg(recip(get_neg(a))));
            if (get_recip(a) >= 0) return simplify(get_recip(a));
            if (get_nat(a, n) && n == 1) return nat(1);
            return unary(T_RECIP, e, a);
        }
        case T_SQRT: return unary(T_SQRT, e, simplify(N(e).a));
        case T_STEP: {
            const int32_t a = simplify(N(e).a);
            if (get_nat(a, n)) return nat(1);
            if (get_neg(a) >= 0 && get_nat(get_neg(a), n)) return nat(n == 0 ? 1 : 0);
            const Case c = from_expr(a);
            if (c.kind == DIV) return nat(c.pos ? 1 : 0);
            return unary(T_STEP, e, a);
        }
        case T_SIN: {
            const int32_t a = simplify(N(e).a);
            if (tag(a) == T_TAU) return nat(0);
            if (get_add(a, p, q)) {
                if (tag(p) == T_TAU) return mk(T_SIN, q);
                if (tag(q) == T_TAU) return mk(T_SIN, p);
            }
            return unary(T_SIN, e, a);
        }
        case T_EXP: {
            const int32_t a = simplify(N(e).a);
            if (get_nat(a, n)) {
                if (n == 0) return nat(1);
                if (n == 1) return mk(T_E);
            }
            return unary(T_EXP, e, a);
        }
        case T_LN: return unary(T_LN, e, simplify(N(e).a));
        case T_ADD: {
            const int32_t a = simplify(N(e).a), b = simplify(N(e).b);
            const Case ca = from_expr(a), cb = from_expr(b);
            if (ca.kind == NAT && ca.a == 0) return b;
            if (cb.kind == NAT && cb.a == 0) return a;
            if (ca.kind != NONE && cb.kind != NONE) {
                if (ca.kind == NAT && cb.kind == NAT) return normalize(add_nat(ca.pos, ca.a, cb.pos, cb.a));
                if (ca.kind == DIV && cb.kind == DIV) return normalize(add_div(ca.pos, ca.a, ca.b, cb.pos, cb.a, cb.b));
                if (ca.kind == NAT) return normalize(add_div(ca.pos, ca.a * cb.b, cb.b, cb.pos, cb.a, cb.b));
                return normalize(add_div(ca.pos, ca.a, ca.b, cb.pos, ca.b * cb.a, ca.b));
            }
            if (get_neg(a) >= 0 && get_neg(b) >= 0) return simplify(neg(add(get_neg(a), get_neg(b))));
            if (get_neg(a) >= 0) return simplify(sub(b, get_neg(a)));
            if (get_sub(a, p, q) && get_add(b, r, t)) {
                if (equal(q, r)) return add(p, t);
                if (equal(q, t)) return add(p, r);
            }
            if (get_sub(b, p, q) && (equal(q, a) || equal(q, b))) return p;
            return binary(T_ADD, e, a, b);
        }
        case T_MUL: {
            const int32_t a = simplify(N(e).a), b = simplify(N(e).b);
            const Case ca = from_expr(a), cb = from_expr(b);
            if (ca.kind == NAT && ca.a == 0) return nat(0);
            if (cb.kind == NAT && cb.a == 0) return nat(0);
            if (ca.kind == NAT && ca.a == 1 && ca.pos) return b;
            if (cb.kind == NAT && cb.a == 1 && cb.pos) return a;
            if (ca.kind == NAT && ca.a == 1 && !ca.pos) return simplify(neg(b));
            if (cb.kind == NAT && cb.a == 1 && !cb.pos) return simplify(neg(a));
            if (ca.kind != NONE && cb.kind != NONE) {
                if (ca.kind == NAT && cb.kind == NAT) return normalize(mul_nat(ca.pos, ca.a, cb.pos, cb.a));
                if (ca.kind == DIV && cb.kind == DIV) return normalize(mul_div(ca.pos, ca.a, ca.b, cb.pos, cb.a, cb.b));
                if (ca.kind == NAT) return normalize(mul_div(ca.pos, ca.a, 1, cb.pos, cb.a, cb.b));
                return normalize(mul_div(cb.pos, cb.a, 1, ca.pos, ca.a, ca.b));
            }
            if (get_neg(a) >= 0 && get_neg(b) >= 0) return simplify(mul(get_neg(a), get_neg(b)));
            if (get_neg(a) >= 0) return simplify(neg(mul(get_neg(a), b)));
            if (get_neg(b) >= 0) return simplify(neg(mul(a, get_neg(b))));
            if (get_recip(a) >= 0 && get_recip(b) >= 0) return simplify(recip(mul(get_recip(a), get_recip(b))));
            if (get_recip(a) >= 0) return simplify(div(b, get_recip(a)));
            {
                int32_t a0, a1, b0, b1;
                const bool da = get_div(a, a0, a1), db = get_div(b, b0, b1);
                if (da && db) return simplify(div(mul(a0, b0), mul(a1, b1)));
                // Not in the reference: there `(a0/a1) * 1/q` takes the rule below, becomes `(a0 * 1/q) / a1`, whose numerator is the
                // quotient a0/q again -- the two divisors change places for ever (src/simplify.rs:276-283; examples/chess.rs
                // builds such terms: DESIGN.md section 5.1).  Merging them ends it.
                if (da && merge_divisors && get_recip(b) >= 0) return simplify(div(a0, mul(a1, get_recip(b))));
                if (da) return simplify(div(mul(a0, b), a1));
                if (db) return simplify(div(mul(a, b0), b1));
            }
            if (get_recip(b) >= 0 && get_nat(get_recip(b), m) && get_mul(a, p, q) && get_nat(q, n))
                return simplify(mul(div(nat(n), nat(m)), p));
            if (get_mul(a, p, q) && get_nat(p, n) && get_nat(b, m)) return simplify(mul(nat(n * m), q));
            return binary(T_MUL, e, a, b);
        }
        case T_MAX: case T_MIN: {
            const int32_t a = simplify(N(e).a), b = simplify(N(e).b);
            if (get_nat(a, n) && get_nat(b, m)) return nat(tag(e) == T_MAX ? (n >= m ? n : m) : (n <= m ? n : m));
            return binary(tag(e), e, a, b);
        }
        case T_LET: return e;
        case T_DECOR: {
            const int32_t a = simplify(N(e).a);
            if (a == N(e).a) return e;
            Node d = N(e); d.a = a; return s.add(d);
        }
        case T_APP: {
            const int32_t a = simplify(N(e).a), b = simplify(N(e).b);
            if (a == N(e).a && b == N(e).b) return e;
            Node d = N(e); d.a = a; d.b = b; return s.add(d);
        }
        default: return e;
        }
    }
};

}   // namespace

// Expr::simplify applied to the three channels (src/lib.rs:601-604).
void scene_simplify(Scene &s, uint32_t flags)
{
    Simp z(s);
    z.merge_divisors = (flags & MARAY_SIMPLIFY_MERGE_DIVISORS) != 0;
    int32_t out[3];
    for (int c = 0; c < 3; c++) out[c] = z.simplify(s.color[c]);
    if (z.runaway)
        throw Error{MARAY_E_LIMIT, "simplify: the rewrite rules do not terminate on this expression (src/simplify.rs:276-283 keeps swapping the divisors of (a/b)/c; "
                                 "maray_scene_simplify_ex with MARAY_SIMPLIFY_MERGE_DIVISORS merges them instead)"};
    for (int c = 0; c < 3; c++) s.color[c] = out[c];
}

}   // namespace maray
