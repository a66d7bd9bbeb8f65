// lower.cpp — Expr -> tape lowering (product code).
//
// One-time replacement for what the reference does per pixel in
// `Expr::eval2` + `Cache` (src/lib.rs:623-670, src/cache.rs:6-42):
//
//  1. fix_color (src/var_fixer.rs:74-82), as every renderer does first
//     (src/render.rs:14,51,117).
//  2. Symbolic evaluation of R, G, B into one hash-consed DAG: Let replaces
//     the context (src/lib.rs:659-662), Var resolves to the first matching
//     definition of the current context (src/cache.rs:32-33) or NaN (:40),
//     Arc/Decor are transparent (:633,:663).  The reference's Cache is keyed by
//     id alone; it is value-transparent iff every id resolves to one DAG node
//     at all of its use sites — checked here, MARAY_E_ALIASED otherwise.
//  3. Constant folding with IEEE-exact ops only (neg abs recip sqrt step add
//     mul max min); sin/exp/ln are never evaluated on the host.
//  4. Dependence classes: Y-only ops go to the ROW section (once per row),
//     everything that depends on X to the PIXEL section.
//  5. Sethi-Ullman ordered DFS schedule (keeps few values live), linear-scan
//     slot allocation, ACC forwarding of results consumed by the next op.
#include "lower.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <exception>
#include <functional>
#include <pthread.h>
#include <unordered_map>
#include <unordered_set>

#include "maray_hip.h"

namespace maray {

namespace {

enum : uint8_t { D_CONST = 100, D_X = 101, D_Y = 102, D_XMAX = 103, D_XMIN = 104, D_YMAX = 105, D_YMIN = 106 };   // leaf kinds; ops use MARAY_OP_*  (XMIN..YMAX: the rectangle of pixels a guard bounds)
enum : uint8_t { DEP_X = 1, DEP_Y = 2 };

struct DNode {
    uint8_t op;
    uint8_t dep;
    uint32_t aux;
    int32_t a, b;
    double cval;
    uint32_t inst = 0;     // 0 = shared (hash-consed); k > 0 = private copy owned by row region k
};

inline uint64_t bits_of(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }

// f64::max / f64::min (src/lib.rs:655-658): NaN-ignoring; for +0 vs -0 the
// IEEE 754-2019 maximumNumber/minimumNumber choice, which is what v_max_f64 /
// v_min_f64 compute on gfx950.
inline double rs_max(double a, double b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return std::signbit(a) ? b : a;
    return a > b ? a : b;
}
inline double rs_min(double a, double b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return std::signbit(a) ? a : b;
    return a < b ? a : b;
}

// Hash-consing table of the DAG: open addressing over node ids (the key of a slot is read from the node it names).  A
// large scene interns 60,000 nodes three or four times over (the scene, its row bounds over x and over y, the private
// copies): in a node-based map that was a sixth of a lowering.
struct Dag {
    std::vector<DNode> n;
    std::vector<int32_t> table;          // node id + 1, 0 = empty; size a power of two, at most half full
    uint32_t folded = 0;
    bool commute = true;
    bool fuse = true;

    static uint64_t hash(const DNode &d) {
        uint64_t h = ((uint64_t)d.op | ((uint64_t)d.aux << 8) | ((uint64_t)d.inst << 32)) * 0x9e3779b97f4a7c15ULL;
        h ^= ((((uint64_t)(uint32_t)d.a << 32) | (uint32_t)d.b) + 0x7f4a7c15ULL + (h << 6) + (h >> 2));
        h *= 0xff51afd7ed558ccdULL;
        h ^= ((d.op == D_CONST ? bits_of(d.cval) : 0) + (h << 6) + (h >> 2));
        return h ^ (h >> 29);
    }
    static bool same(const DNode &x, const DNode &y) {
        return x.op == y.op && x.aux == y.aux && x.inst == y.inst && x.a == y.a && x.b == y.b && (x.op != D_CONST || bits_of(x.cval) == bits_of(y.cval));
    }
    void reserve(size_t nodes) {
        n.reserve(nodes);
        size_t want = 1024;
        while (want < 2 * nodes) want *= 2;
        if (want > table.size()) rehash(want);
    }
    void rehash(size_t size) {
        table.assign(size, 0);
        for (size_t id = 0; id < n.size(); id++) {
            size_t at = hash(n[id]) & (size - 1);
            while (table[at]) at = (at + 1) & (size - 1);
            table[at] = (int32_t)id + 1;
        }
    }
    int32_t intern(const DNode &d) {
        if (2 * (n.size() + 1) > table.size()) rehash(table.empty() ? 1024 : 2 * table.size());
        const size_t mask = table.size() - 1;
        size_t at = hash(d) & mask;
        for (; table[at]; at = (at + 1) & mask)
            if (same(n[table[at] - 1], d)) return table[at] - 1;
        n.push_back(d);
        table[at] = (int32_t)n.size();
        return (int32_t)n.size() - 1;
    }
    int32_t konst(double v) {
        if (v != v) v = NAN;   // one canonical NaN
        return intern(DNode{D_CONST, 0, 0, -1, -1, v});
    }
    int32_t leaf(uint8_t kind) { return intern(DNode{kind, (uint8_t)(kind == D_X ? DEP_X : kind == D_Y ? DEP_Y : 0), 0, -1, -1, 0.0}); }

    int32_t unary(uint8_t op, int32_t a) {
        if (n[a].op == D_CONST) {
            double v = n[a].cval;
            switch (op) {
            case MARAY_OP_NEG: folded++; return konst(-v);
            case MARAY_OP_ABS: folded++; return konst(std::fabs(v));
            case MARAY_OP_RECIP: folded++; return konst(1.0 / v);
            case MARAY_OP_SQRT: folded++; return konst(std::sqrt(v));
            case MARAY_OP_STEP: folded++; return konst(v >= 0.0 ? 1.0 : 0.0);
            default: break;   // sin / exp / ln are evaluated on the device only
            }
        }
        if (fuse && op == MARAY_OP_STEP && n[a].op == MARAY_OP_SIN)   // Step(Sin(t)): only the sign of sin is observable here
            return intern(DNode{MARAY_OP_STEPSIN, n[n[a].a].dep, 0, n[a].a, -1, 0.0});
        return intern(DNode{op, n[a].dep, 0, a, -1, 0.0});
    }
    int32_t binary(uint8_t op, int32_t a, int32_t b) {
        if (n[a].op == D_CONST && n[b].op == D_CONST) {
            double x = n[a].cval, y = n[b].cval;
            folded++;
            switch (op) {
            case MARAY_OP_ADD: return konst(x + y);
            case MARAY_OP_MUL: return konst(x * y);
            case MARAY_OP_MAX: return konst(rs_max(x, y));
            default: return konst(rs_min(x, y));
            }
        }
        if (commute && a > b) std::swap(a, b);   // + * max min are commutative in value: canonical operand order
        return intern(DNode{op, (uint8_t)(n[a].dep | n[b].dep), 0, a, b, 0.0});
    }
    int32_t app(uint32_t id, int32_t a, int32_t b) {
        if (id % 5u >= 3u) return intern(DNode{MARAY_OP_TEXDIM, 0, id, -1, -1, 0.0});   // width/height ignore a, b
        return intern(DNode{MARAY_OP_APP, (uint8_t)(n[a].dep | n[b].dep), id, a, b, 0.0});
    }
};

// uint64 -> node id, open addressing (values are stored + 1: 0 marks an empty slot)
struct FlatMap64 {
    std::vector<std::pair<uint64_t, int32_t>> t;
    size_t used = 0;
    static size_t mix(uint64_t k) { k *= 0x9e3779b97f4a7c15ULL; return (size_t)(k ^ (k >> 29)); }
    void grow(size_t size) {
        std::vector<std::pair<uint64_t, int32_t>> old(size);
        old.swap(t);
        for (const auto &e : old) {
            if (!e.second) continue;
            size_t at = mix(e.first) & (size - 1);
            while (t[at].second) at = (at + 1) & (size - 1);
            t[at] = e;
        }
    }
    void reserve(size_t n) { size_t want = 1024; while (want < 2 * n) want *= 2; if (want > t.size()) grow(want); }
    const int32_t *find(uint64_t k) const {
        if (t.empty()) return nullptr;
        const size_t mask = t.size() - 1;
        for (size_t at = mix(k) & mask; t[at].second; at = (at + 1) & mask)
            if (t[at].first == k) return &t[at].second;
        return nullptr;
    }
    void put(uint64_t k, int32_t v) {               // k is not in the table
        if (2 * (used + 1) > t.size()) grow(t.empty() ? 1024 : 2 * t.size());
        const size_t mask = t.size() - 1;
        size_t at = mix(k) & mask;
        while (t[at].second) at = (at + 1) & mask;
        t[at] = {k, v + 1};
        used++;
    }
};

// ---- symbolic evaluation -------------------------------------------------------
struct SymEval {
    const Scene &s;
    Dag &g;
    FlatMap64 memo;                                     // (expr, ctx) -> dag node
    FlatMap64 var_memo;                                 // (ctx, def index) -> dag node
    std::unordered_set<uint64_t> in_progress;
    std::unordered_map<uint64_t, int32_t> id_node;      // Var id -> dag node at its use sites (Cache transparency)
    int depth = 0;

    SymEval(const Scene &s_, Dag &g_) : s(s_), g(g_) { memo.reserve(s_.nodes.size()); }

    int32_t var(uint64_t id, int32_t ctx) {
        int32_t r = -1;
        if (ctx >= 0) {
            const Ctx &c = s.ctxs[ctx];
            for (size_t i = 0; i < c.ids.size(); i++) {
                if (c.ids[i] != id) continue;                       // first match wins (src/cache.rs:32-33)
                uint64_t k = ((uint64_t)(uint32_t)ctx << 32) | (uint32_t)i;
                if (const int32_t *hit = var_memo.find(k)) { r = *hit - 1; break; }
                if (!in_progress.insert(k).second)
                    throw Error{MARAY_E_CYCLE, "Let variable " + std::to_string(id) + " is defined in terms of itself"};
                r = eval(c.defs[i], ctx);                            // definitions see their own Let's ctx (src/cache.rs:34)
                in_progress.erase(k);
                var_memo.put(k, r);
                break;
            }
        }
        if (r < 0) r = g.konst(NAN);                                 // unknown variable (src/cache.rs:40)
        auto ins = id_node.emplace(id, r);
        if (!ins.second && ins.first->second != r)
            throw Error{MARAY_E_ALIASED, "variable id " + std::to_string(id) +
                        " resolves to different values in different scopes; the reference's id-keyed Cache makes the result order dependent"};
        return r;
    }

    int32_t eval(int32_t e, int32_t ctx) {
        uint64_t key = ((uint64_t)(uint32_t)e << 32) | (uint32_t)(ctx + 1);
        if (const int32_t *hit = memo.find(key)) return *hit - 1;
        if (++depth > 2000000) throw Error{MARAY_E_LIMIT, "expression too deep"};
        const Node &n = s.nodes[e];
        int32_t r;
        switch (n.tag) {
        case T_ARC: case T_DECOR: r = eval(n.a, ctx); break;         // :633, :663
        case T_X: r = g.leaf(D_X); break;
        case T_Y: r = g.leaf(D_Y); break;
        case T_TAU: r = g.konst(6.283185307179586); break;           // :636
        case T_E: r = g.konst(2.718281828459045); break;             // :637
        case T_VAR: r = var(n.u, ctx); break;                        // :638
        case T_NAT: r = g.konst((double)n.u); break;                 // :639 `n as f64`
        case T_LET: r = eval(n.a, n.ctx); break;                     // :659-662 ctx replaced
        case T_APP: {
            int32_t a = eval(n.a, ctx), b = eval(n.b, ctx);
            r = g.app(n.app, a, b);
            break;
        }
        default:
            if (is_binary(n.tag)) {
                int32_t a = eval(n.a, ctx), b = eval(n.b, ctx);
                static const uint8_t ops[] = {MARAY_OP_ADD, MARAY_OP_MUL, MARAY_OP_MAX, MARAY_OP_MIN};
                r = g.binary(ops[n.tag - T_ADD], a, b);
            } else {
                static const uint8_t ops[] = {MARAY_OP_NEG, MARAY_OP_ABS, MARAY_OP_RECIP, MARAY_OP_SQRT,
                                              MARAY_OP_STEP, MARAY_OP_SIN, MARAY_OP_EXP, MARAY_OP_LN};
                r = g.unary(ops[n.tag - T_NEG], eval(n.a, ctx));
            }
        }
        depth--;
        memo.put(key, r);
        return r;
    }
};

// ---- interval analysis ----------------------------------------------------------------
// Conservative bounds of every DAG node over the pixel domain x, y in
// [0, MARAY_DOMAIN_MAX): used to prove that the argument of a Sin stays inside
// glibc's reduce_sincos range (so the specialised kernels need no huge-argument
// path for that op).  Bounds are widened by one ulp per operation; `nan` marks
// values that may be NaN.
struct Ival {
    double lo, hi;
    bool nan;
};

inline double down(double v) { return std::nextafter(v, -INFINITY); }
inline double up(double v) { return std::nextafter(v, INFINITY); }

Ival ival_mul(const Ival &a, const Ival &b)
{
    Ival r{INFINITY, -INFINITY, a.nan || b.nan};
    const double xs[2] = {a.lo, a.hi}, ys[2] = {b.lo, b.hi};
    for (double x : xs) for (double y : ys) {
        const double p = x * y;
        if (p != p) { r.nan = true; r.lo = -INFINITY; r.hi = INFINITY; continue; }   // 0 * inf
        r.lo = std::min(r.lo, p); r.hi = std::max(r.hi, p);
    }
    // 0 * inf in the interior: one factor can be zero while the other can be infinite
    const bool a_zero = a.lo <= 0 && a.hi >= 0, b_zero = b.lo <= 0 && b.hi >= 0;
    const bool a_inf = std::isinf(a.lo) || std::isinf(a.hi), b_inf = std::isinf(b.lo) || std::isinf(b.hi);
    if ((a_zero && b_inf) || (b_zero && a_inf)) { r.nan = true; r.lo = -INFINITY; r.hi = INFINITY; }
    r.lo = down(r.lo); r.hi = up(r.hi);
    if (!r.nan) {      // the margin must not cross zero for factors of known sign
        if ((a.lo >= 0 && b.lo >= 0) || (a.hi <= 0 && b.hi <= 0)) r.lo = std::max(r.lo, 0.0);
        if ((a.lo >= 0 && b.hi <= 0) || (a.hi <= 0 && b.lo >= 0)) r.hi = std::min(r.hi, 0.0);
    }
    return r;
}

// (`have`: the intervals of the first nodes of a DAG that has grown since -- a node's interval depends on its operands' only)
std::vector<Ival> intervals(const Dag &g, const std::vector<Ival> *have = nullptr)
{
    const double dmax = (double)MARAY_DOMAIN_MAX - 1.0;
    std::vector<Ival> v;
    v.reserve(g.n.size());
    if (have) v = *have;
    const size_t first = v.size();
    v.resize(g.n.size());
    for (size_t i = first; i < g.n.size(); i++) {        // children are interned before their parents
        const DNode &d = g.n[i];
        const Ival a = d.a >= 0 ? v[d.a] : Ival{0, 0, false};
        const Ival b = d.b >= 0 ? v[d.b] : Ival{0, 0, false};
        Ival r{-INFINITY, INFINITY, true};
        switch (d.op) {
        case D_CONST: r = (d.cval != d.cval) ? Ival{-INFINITY, INFINITY, true} : Ival{d.cval, d.cval, false}; break;
        case D_X: case D_Y: case D_XMAX: case D_XMIN: case D_YMAX: case D_YMIN: r = Ival{0.0, dmax, false}; break;
        case MARAY_OP_MOV: r = a; break;
        case MARAY_OP_NEG: r = Ival{-a.hi, -a.lo, a.nan}; break;
        case MARAY_OP_ABS:
            r = (a.lo >= 0) ? a : (a.hi <= 0 ? Ival{-a.hi, -a.lo, a.nan} : Ival{0.0, std::max(-a.lo, a.hi), a.nan});
            break;
        case MARAY_OP_RECIP:
            if (a.lo > 0 || a.hi < 0) r = Ival{down(1.0 / a.hi), up(1.0 / a.lo), a.nan};
            else r = Ival{-INFINITY, INFINITY, a.nan};
            break;
        case MARAY_OP_SQRT: r = Ival{a.lo > 0 ? down(std::sqrt(a.lo)) : 0.0, a.hi > 0 ? up(std::sqrt(a.hi)) : 0.0, a.nan || a.lo < 0}; break;
        case MARAY_OP_STEP: case MARAY_OP_STEPSIN: r = Ival{0.0, 1.0, false}; break;
        case MARAY_OP_SIN: r = Ival{-1.0, 1.0, a.nan || std::isinf(a.lo) || std::isinf(a.hi)}; break;
        case MARAY_OP_EXP: r = Ival{std::max(0.0, down(std::exp(a.lo) * (1 - 0x1p-50))), up(std::exp(a.hi) * (1 + 0x1p-50)), a.nan}; break;
        case MARAY_OP_LN:
            r = Ival{a.lo > 0 ? down(std::log(a.lo) - 0x1p-40) : -INFINITY, a.hi > 0 ? up(std::log(a.hi) + 0x1p-40) : -INFINITY, a.nan || a.lo < 0};
            break;
        case MARAY_OP_ADD: {
            const double lo = a.lo + b.lo, hi = a.hi + b.hi;
            r = Ival{lo != lo ? -INFINITY : down(lo), hi != hi ? INFINITY : up(hi), a.nan || b.nan || lo != lo || hi != hi ||
                     (std::isinf(a.lo) && std::isinf(b.hi)) || (std::isinf(a.hi) && std::isinf(b.lo))};
            // the outward step is a margin, not a need (rounding is monotone: fl(a + b) >= fl(a.lo + b.lo)); it must not
            // carry a sum of non-negative terms below zero (0 + 0 -> -4.9e-324 would make the sqrt above it "may be NaN")
            if (a.lo >= 0 && b.lo >= 0) r.lo = std::max(r.lo, 0.0);
            if (a.hi <= 0 && b.hi <= 0) r.hi = std::min(r.hi, 0.0);
            break;
        }
        case MARAY_OP_MUL:
            r = ival_mul(a, b);
            // a square: never below zero, whatever the corner products say (p2_len = sqrt(a*a + b*b), src/lib.rs:1027-1029,
            // is how every signed-distance shape measures; the corner rule alone gives lo * hi < 0 for an operand that
            // changes sign, and the sqrt above it would count as a possible NaN)
            if (d.a == d.b && !a.nan && !r.nan) {
                const double m = a.lo > 0 ? a.lo : (a.hi < 0 ? -a.hi : 0.0);
                r.lo = std::max(r.lo, m > 0 ? down(m * m) : 0.0);
            }
            break;
        case MARAY_OP_MAX: r = Ival{std::max(a.lo, b.lo), std::max(a.hi, b.hi), a.nan && b.nan}; if (a.nan || b.nan) { r.lo = std::min(a.lo, b.lo); } break;
        case MARAY_OP_MIN: r = Ival{std::min(a.lo, b.lo), std::min(a.hi, b.hi), a.nan && b.nan}; if (a.nan || b.nan) { r.hi = std::max(a.hi, b.hi); } break;
        case MARAY_OP_APP: r = Ival{0.0, 255.0, false}; break;
        case MARAY_OP_TEXDIM: r = Ival{0.0, 4294967295.0, false}; break;
        default: break;
        }
        v[i] = r;
    }
    return v;
}

// ---- section scheduling + encoding --------------------------------------------------
// ---- monotonicity in x and row bounds -----------------------------------------------------
// For a fixed row, is a value a monotone function of the pixel's x *as computed in f64*?  Every
// rule below composes correctly rounded, NaN-free operations, each of which is monotone in the
// operand that varies (rounding is monotone), so the property holds for the rounded results and
// not just for the real-number formula.
enum Mono : uint8_t { M_CONSTX, M_INC, M_DEC, M_MONO /* monotone, direction depends on the row */, M_NONE };

inline Mono flip(Mono m) { return m == M_INC ? M_DEC : m == M_DEC ? M_INC : m; }
inline Mono join(Mono a, Mono b)     // both operands vary together (add / min / max)
{
    if (a == M_CONSTX) return b;
    if (b == M_CONSTX) return a;
    if (a == b && (a == M_INC || a == M_DEC)) return a;
    return M_NONE;
}

// Monotonicity in one coordinate (dep_bit / var_leaf = DEP_X / D_X or DEP_Y / D_Y) with everything else held fixed.
std::vector<Mono> monotonicity(const Dag &g, const std::vector<Ival> &iv, uint8_t dep_bit = DEP_X, uint8_t var_leaf = D_X)
{
    std::vector<Mono> m(g.n.size(), M_NONE);
    for (size_t i = 0; i < g.n.size(); i++) {
        const DNode &d = g.n[i];
        if (!(d.dep & dep_bit)) { m[i] = M_CONSTX; continue; }
        if (iv[i].nan) { m[i] = M_NONE; continue; }
        const Mono a = d.a >= 0 ? m[d.a] : M_NONE, b = d.b >= 0 ? m[d.b] : M_NONE;
        if (d.op == var_leaf) { m[i] = M_INC; continue; }
        switch (d.op) {
        case MARAY_OP_MOV: case MARAY_OP_STEP: m[i] = a; break;
        case MARAY_OP_NEG: m[i] = flip(a); break;
        case MARAY_OP_SQRT: m[i] = iv[d.a].lo >= 0 ? a : M_NONE; break;
        case MARAY_OP_ABS: m[i] = iv[d.a].lo >= 0 ? a : (iv[d.a].hi <= 0 ? flip(a) : M_NONE); break;
        case MARAY_OP_RECIP: m[i] = (iv[d.a].lo > 0 || iv[d.a].hi < 0) ? flip(a) : M_NONE; break;
        case MARAY_OP_ADD: case MARAY_OP_MIN: case MARAY_OP_MAX: m[i] = join(a, b); break;
        case MARAY_OP_MUL: {
            // one factor must be fixed along the row; its sign decides the direction
            const int32_t v = (b == M_CONSTX) ? d.a : (a == M_CONSTX ? d.b : -1);
            const int32_t c = (b == M_CONSTX) ? d.b : d.a;
            if (v < 0 && (a == M_INC || a == M_DEC) && (b == M_INC || b == M_DEC)) {
                // both factors vary: monotone when both are sign-definite and their MAGNITUDES move the same way
                // (0 <= f1 <= f2, 0 <= g1 <= g2  =>  f1 g1 <= f2 g2 in the reals, and rounding is monotone); the sign of
                // the product then says which way the value goes.  Covers squares of sign-definite terms.
                const Ival &fa = iv[d.a], &fb = iv[d.b];
                const int sa = fa.lo >= 0 ? 1 : (fa.hi <= 0 ? -1 : 0), sb = fb.lo >= 0 ? 1 : (fb.hi <= 0 ? -1 : 0);
                if (sa && sb) {
                    const Mono ma = sa > 0 ? a : flip(a), mb = sb > 0 ? b : flip(b);      // direction of |f|, |g|
                    if (ma == mb) { m[i] = sa * sb > 0 ? ma : flip(ma); break; }
                }
                m[i] = M_NONE;
                break;
            }
            if (v < 0 || m[v] == M_NONE) { m[i] = M_NONE; break; }
            if (iv[c].lo >= 0) m[i] = m[v];
            else if (iv[c].hi <= 0) m[i] = flip(m[v]);
            else m[i] = M_MONO;
            break;
        }
        default: m[i] = M_NONE;   // sin / exp / ln (libm monotonicity is not guaranteed), App
        }
    }
    return m;
}

// Builds, for boolean nodes, boolean expressions free of one coordinate that bound them over a span of it:
//   ub(v) == 0  =>  v == 0 for every value of the coordinate in [lo, hi]        (lb: == 1 => v == 1)
// A monotone boolean takes its extreme values at the ends of the span (monotone on the domain is monotone on any
// sub-interval).  Applied to x with lo / hi = XMIN / XMAX it gives the row guards; applied once more, to y with
// YMIN / YMAX, to those guards, it gives guards that hold over a rectangle of pixels.
// A map keyed by DAG node ids (dense, growing): a vector and a presence byte per id (the memo tables of the passes below were
// node-based hash maps: a tenth of a large scene's lowering)
template <class T> struct IdMap {
    std::vector<T> v;
    std::vector<uint8_t> has;
    std::vector<int32_t> keys;
    const T *find(int32_t k) const { return (size_t)k < has.size() && has[k] ? &v[k] : nullptr; }
    void put(int32_t k, const T &x) {
        if ((size_t)k >= has.size()) { const size_t n = std::max<size_t>(2 * has.size(), (size_t)k + 1024); v.resize(n); has.resize(n, 0); }
        if (!has[k]) keys.push_back(k);
        has[k] = 1; v[k] = x;
    }
    void clear() { for (int32_t k : keys) has[k] = 0; keys.clear(); }
    void reserve(size_t n) { if (n > has.size()) { v.resize(n); has.resize(n, 0); } }
};

struct RowBounds {
    struct B { int32_t ub, lb; bool lossy; };       // lossy: somewhere below, a sub-expression could only be bounded by "anything"
    Dag &g;
    const std::vector<uint8_t> &isbool;
    const std::vector<Mono> &mono;
    const std::vector<Ival> &range;                 // static interval of every node over the whole domain
    const uint8_t dep_bit, var_leaf;
    IdMap<int32_t> sub_lo, sub_hi;                   // subst(i, x0), subst(i, xmax)
    IdMap<B> memo;
    struct IV { int32_t lo, hi; bool ok() const { return lo >= 0; } };
    IdMap<IV> iv_memo;
    int32_t c_true, c_false, x0, xmax;

    RowBounds(Dag &g_, const std::vector<uint8_t> &b, const std::vector<Mono> &m, const std::vector<Ival> &rg, uint8_t dep_bit_, uint8_t var_leaf_,
              uint8_t lo_leaf, uint8_t hi_leaf)
        : g(g_), isbool(b), mono(m), range(rg), dep_bit(dep_bit_), var_leaf(var_leaf_) {
        sub_lo.reserve(2 * g.n.size()); sub_hi.reserve(2 * g.n.size()); memo.reserve(2 * g.n.size()); iv_memo.reserve(2 * g.n.size());
        c_true = g.konst(1.0); c_false = g.konst(0.0); x0 = g.leaf(lo_leaf); xmax = g.leaf(hi_leaf);
    }
    int32_t subst(int32_t i, int32_t xr) {          // i with the coordinate replaced by node xr
        const DNode d = g.n[i];
        if (!(d.dep & dep_bit)) return i;
        if (d.op == var_leaf) return xr;
        IdMap<int32_t> &sub_memo = xr == x0 ? sub_lo : sub_hi;          // (the coordinate is only ever replaced by one of the span's ends)
        if (xr != x0 && xr != xmax) throw Error{MARAY_E_INTERNAL, "subst: not an end of the span"};
        if (const int32_t *hit = sub_memo.find(i)) return *hit;
        int32_t r;
        if (d.b >= 0) {
            const int32_t a = subst(d.a, xr), b = subst(d.b, xr);
            r = d.op == MARAY_OP_APP ? g.app(d.aux, a, b) : g.binary(d.op, a, b);
        } else r = g.unary(d.op, subst(d.a, xr));
        sub_memo.put(i, r);
        return r;
    }
    int32_t b_not(int32_t a) { return g.binary(MARAY_OP_ADD, c_true, g.unary(MARAY_OP_NEG, a)); }

    // Interval of an arithmetic value over the span, as two expressions free of the coordinate: lo <= v <= hi for every
    // value of the coordinate in [x0, xmax].  Every op here is correctly rounded, hence monotone in each operand, so
    // interval arithmetic in the SAME arithmetic bounds the ROUNDED results (fl(a+b) <= fl(ha+hb) when a <= ha, b <= hb).
    // Monotone sub-trees take their end points (tight); a difference of two increasing terms -- an edge function
    // written as p - q -- gets [p(lo) - q(hi), p(hi) - q(lo)], which the end-point rule cannot bound at all.
    // Only nodes that the static analysis proves NaN-free are bounded; sin / exp / ln / App are not.
    IV ival(int32_t i) {
        const DNode d = g.n[i];
        if (!(d.dep & dep_bit)) return {i, i};
        if (d.op == var_leaf) return {x0, xmax};
        if (const IV *hit = iv_memo.find(i)) return *hit;
        IV r{-1, -1};
        const Mono m = (size_t)i < mono.size() ? mono[i] : M_NONE;
        const bool clean = (size_t)i < range.size() && !range[i].nan;
        if (m == M_INC) r = {subst(i, x0), subst(i, xmax)};
        else if (m == M_DEC) r = {subst(i, xmax), subst(i, x0)};
        else if (clean) {
            auto rng = [&](int32_t k) { return (size_t)k < range.size() ? range[k] : Ival{-INFINITY, INFINITY, true}; };
            switch (d.op) {
            case MARAY_OP_MOV: r = ival(d.a); break;
            case MARAY_OP_NEG: { const IV a = ival(d.a); if (a.ok()) r = {g.unary(MARAY_OP_NEG, a.hi), g.unary(MARAY_OP_NEG, a.lo)}; break; }
            case MARAY_OP_STEP: { const IV a = ival(d.a); if (a.ok()) r = {g.unary(MARAY_OP_STEP, a.lo), g.unary(MARAY_OP_STEP, a.hi)}; break; }
            case MARAY_OP_SQRT: { const IV a = ival(d.a); if (a.ok() && rng(d.a).lo >= 0) r = {g.unary(MARAY_OP_SQRT, a.lo), g.unary(MARAY_OP_SQRT, a.hi)}; break; }
            case MARAY_OP_RECIP: {      // decreasing on either side of 0
                const IV a = ival(d.a);
                if (a.ok() && (rng(d.a).lo > 0 || rng(d.a).hi < 0)) r = {g.unary(MARAY_OP_RECIP, a.hi), g.unary(MARAY_OP_RECIP, a.lo)};
                break;
            }
            case MARAY_OP_ABS: {
                const IV a = ival(d.a);
                if (!a.ok()) break;
                if (rng(d.a).lo >= 0) r = a;
                else if (rng(d.a).hi <= 0) r = {g.unary(MARAY_OP_NEG, a.hi), g.unary(MARAY_OP_NEG, a.lo)};
                else      // may change sign over the domain: over THIS span it is at least its distance from zero, max(lo, -hi, +0)
                    r = {g.binary(MARAY_OP_MAX, g.binary(MARAY_OP_MAX, a.lo, g.unary(MARAY_OP_NEG, a.hi)), c_false),
                         g.binary(MARAY_OP_MAX, g.unary(MARAY_OP_ABS, a.lo), g.unary(MARAY_OP_ABS, a.hi))};
                break;
            }
            case MARAY_OP_ADD: case MARAY_OP_MIN: case MARAY_OP_MAX: {
                const IV a = ival(d.a), b = ival(d.b);
                if (a.ok() && b.ok()) r = {g.binary(d.op, a.lo, b.lo), g.binary(d.op, a.hi, b.hi)};
                break;
            }
            case MARAY_OP_MUL: {
                const IV a = ival(d.a), b = ival(d.b);
                if (!a.ok() || !b.ok()) break;
                const Ival ra = rng(d.a), rb = rng(d.b);
                auto mul = [&](int32_t p, int32_t q) { return g.binary(MARAY_OP_MUL, p, q); };
                if (d.a == d.b && !(ra.lo >= 0) && !(ra.hi <= 0)) {
                    // a square of something that may change sign over the domain: [m * m, max(lo * lo, hi * hi)] with
                    // m = max(lo, -hi, +0) the operand's distance from zero over the span (fl(v * v) >= fl(m * m) for
                    // |v| >= m; 0 when the span straddles zero) -- the corner rule below would answer lo * hi < 0
                    const int32_t m = g.binary(MARAY_OP_MAX, g.binary(MARAY_OP_MAX, a.lo, g.unary(MARAY_OP_NEG, a.hi)), c_false);
                    r = {mul(m, m), g.binary(MARAY_OP_MAX, mul(a.lo, a.lo), mul(a.hi, a.hi))};
                    fired_sq++;
                }
                else if (ra.lo >= 0 && rb.lo >= 0) r = {mul(a.lo, b.lo), mul(a.hi, b.hi)};
                else if (ra.hi <= 0 && rb.hi <= 0) r = {mul(a.hi, b.hi), mul(a.lo, b.lo)};
                else if (ra.lo >= 0 && rb.hi <= 0) r = {mul(a.hi, b.lo), mul(a.lo, b.hi)};
                else if (ra.hi <= 0 && rb.lo >= 0) r = {mul(a.lo, b.hi), mul(a.hi, b.lo)};
                else {      // a sign is unknown: the extremes are among the four corner products
                    const int32_t p0 = mul(a.lo, b.lo), p1 = mul(a.lo, b.hi), p2 = mul(a.hi, b.lo), p3 = mul(a.hi, b.hi);
                    r = {g.binary(MARAY_OP_MIN, g.binary(MARAY_OP_MIN, p0, p1), g.binary(MARAY_OP_MIN, p2, p3)),
                         g.binary(MARAY_OP_MAX, g.binary(MARAY_OP_MAX, p0, p1), g.binary(MARAY_OP_MAX, p2, p3))};
                }
                break;
            }
            default: break;
            }
            if (r.ok() && d.op < MARAY_OP_COUNT) fired[d.op]++;
        }
        else if (r.ok()) fired_mono++;
        iv_memo.put(i, r);
        return r;
    }
    // which rules produced an interval (MARAY_TRACE_LOWER=1 prints them: what a fuzz family reaches)
    uint32_t fired[MARAY_OP_COUNT] = {}, fired_mono = 0, fired_sq = 0;
    void trace(const char *pass) const {
        static const char *names[] = {"MOV", "NEG", "ABS", "RECIP", "SQRT", "STEP", "ADD", "MUL", "MAX", "MIN"};
        static const uint8_t ops[] = {MARAY_OP_MOV, MARAY_OP_NEG, MARAY_OP_ABS, MARAY_OP_RECIP, MARAY_OP_SQRT, MARAY_OP_STEP, MARAY_OP_ADD,
                                      MARAY_OP_MUL, MARAY_OP_MAX, MARAY_OP_MIN};
        fprintf(stderr, "maray lower: interval rules fired, bounds over %s: end points of monotone values %u, squares %u", pass, fired_mono, fired_sq);
        for (size_t k = 0; k < sizeof ops; k++) fprintf(stderr, ", %s %u", names[k], fired[ops[k]]);
        fprintf(stderr, "\n");
    }

    // For an arithmetic value v: an expression free of the coordinate with  zub == 0  =>  v is +0.0 (that very bit
    // pattern) for every value of the coordinate in the span.  A boolean is its own witness; a product with a finite,
    // sign-clear factor keeps it (+0 * c = +0); max, min and sum of sign-clear values combine their operands' (all
    // operands +0 => +0).  c_true = no such statement.  `clear[i]`: the static analysis proves i NaN-free with its
    // sign bit clear, `finite[i]`: and below +inf.
    IdMap<int32_t> z_memo;
    int32_t zub(int32_t i, const std::vector<uint8_t> &clear, const std::vector<uint8_t> &finite) {
        if (i < 0 || (size_t)i >= isbool.size()) return c_true;
        if (isbool[i]) return bounds(i).ub;
        const DNode d = g.n[i];
        if (!(d.dep & dep_bit) || d.a < 0 || d.b < 0 || !clear[i]) return c_true;
        if (const int32_t *hit = z_memo.find(i)) return *hit;
        int32_t r = c_true;
        const int32_t za = zub(d.a, clear, finite), zb = zub(d.b, clear, finite);
        switch (d.op) {
        case MARAY_OP_MUL: {
            const bool by_a = za != c_true && clear[d.b] && finite[d.b], by_b = zb != c_true && clear[d.a] && finite[d.a];
            if (by_a && by_b) r = g.binary(MARAY_OP_MIN, za, zb);
            else if (by_a) r = za;
            else if (by_b) r = zb;
            break;
        }
        case MARAY_OP_MAX: case MARAY_OP_ADD:
            if (za != c_true && zb != c_true && clear[d.a] && clear[d.b]) r = g.binary(MARAY_OP_MAX, za, zb);
            break;
        case MARAY_OP_MIN:
            if (clear[d.a] && clear[d.b]) {
                if (za != c_true && zb != c_true) r = g.binary(MARAY_OP_MIN, za, zb);
                else r = za != c_true ? za : zb;
            }
            break;
        default: break;
        }
        z_memo.put(i, r);
        return r;
    }

    B bounds(int32_t i) {
        if ((size_t)i >= isbool.size() || !isbool[i]) return {c_true, c_false, true};
        const DNode d = g.n[i];
        if (!(d.dep & dep_bit)) return {i, i, false};
        if (const B *hit = memo.find(i)) return *hit;
        B r{c_true, c_false, true};
        const Mono m = (size_t)i < mono.size() ? mono[i] : M_NONE;
        if (m == M_INC || m == M_DEC || m == M_MONO) {
            const int32_t at0 = subst(i, x0), atw = subst(i, xmax);
            if (m == M_INC) r = {atw, at0, false};
            else if (m == M_DEC) r = {at0, atw, false};
            else r = {g.binary(MARAY_OP_MAX, at0, atw), g.binary(MARAY_OP_MIN, at0, atw), false};
        } else if (d.op == MARAY_OP_MUL || d.op == MARAY_OP_MIN) {
            const B a = bounds(d.a), b = bounds(d.b);
            r = {g.binary(MARAY_OP_MIN, a.ub, b.ub), g.binary(MARAY_OP_MIN, a.lb, b.lb), a.lossy || b.lossy};
        } else if (d.op == MARAY_OP_MAX) {
            const B a = bounds(d.a), b = bounds(d.b);
            r = {g.binary(MARAY_OP_MAX, a.ub, b.ub), g.binary(MARAY_OP_MAX, a.lb, b.lb), a.lossy || b.lossy};
        } else if (d.op == MARAY_OP_ADD) {       // 1 + -(b) = NOT b
            const int32_t nb = g.n[d.a].op == MARAY_OP_NEG ? d.a : d.b;
            const B a = bounds(g.n[nb].a);
            r = {b_not(a.lb), b_not(a.ub), a.lossy};
        } else if (d.op == MARAY_OP_STEP) {      // not monotone as a whole: bound its argument by interval arithmetic
            const IV a = ival(d.a);
            if (a.ok()) r = {g.unary(MARAY_OP_STEP, a.hi), g.unary(MARAY_OP_STEP, a.lo), false};
        }
        memo.put(i, r);
        return r;
    }
};


constexpr size_t MIN_ROW_REGION = 12;           // smallest region worth a row-level SKIP op (guard = a y value)
constexpr size_t MAX_REGION = 0x1FFF;           // aux field


// Long chains of one boolean connective -- max(t1, max(t2, max(t3, ...))), the way a scene paints
// shape over shape -- are rebuilt as balanced trees over the same operands in the same order.
// On {+0.0, 1.0} max is OR and mul / min are AND: associative and commutative bit for bit, so every
// value is unchanged.  A balanced tree has sub-trees of 4, 16, ... shapes whose bound over a row is
// still selective, which lets one row-level SKIP op drop a whole group of shapes (a right-deep
// chain only has "this shape" and "all the remaining ones").
struct Rebalancer {
    Dag &g;
    const std::vector<uint8_t> &isbool;
    std::vector<uint32_t> users;
    std::vector<int32_t> remap;
    uint32_t rebuilt = 0;

    Rebalancer(Dag &g_, const std::vector<uint8_t> &b, const int32_t roots[3]) : g(g_), isbool(b), users(g_.n.size(), 0), remap(g_.n.size(), -2) {
        std::vector<uint8_t> seen(g.n.size(), 0);
        std::vector<int32_t> st(roots, roots + 3);
        for (int c = 0; c < 3; c++) users[roots[c]]++;
        while (!st.empty()) {
            const int32_t i = st.back(); st.pop_back();
            if (seen[i]) continue;
            seen[i] = 1;
            for (int32_t c : {g.n[i].a, g.n[i].b}) if (c >= 0) { users[c]++; st.push_back(c); }
        }
    }
    // 1: OR, 2: AND of two booleans; 3 / 4: max / min of anything else (NaN-ignoring max and min with +0 > -0 are
    // associative and commutative as well: the result is the largest / smallest non-NaN operand whatever the order,
    // a NaN only if all are); 0: anything else
    int cls(int32_t i) const {
        if (i < 0 || (size_t)i >= isbool.size()) return 0;
        const DNode &d = g.n[i];
        if (d.a < 0 || d.b < 0) return 0;
        const bool bools = isbool[i] && isbool[d.a] && isbool[d.b];
        if (d.op == MARAY_OP_MAX) return bools ? 1 : 3;
        if (d.op == MARAY_OP_MIN) return bools ? 2 : 4;
        if (d.op == MARAY_OP_MUL && bools) return 2;
        return 0;
    }
    void leaves(int32_t i, int c, bool top, std::vector<int32_t> &out) {
        if (cls(i) == c && (top || users[i] == 1)) {
            const DNode d = g.n[i];
            leaves(d.a, c, false, out);
            leaves(d.b, c, false, out);
        } else out.push_back(map(i));
    }
    int32_t tree(const std::vector<int32_t> &l, size_t lo, size_t hi, uint8_t op) {
        if (hi - lo == 1) return l[lo];
        const size_t mid = lo + (hi - lo + 1) / 2;
        return g.binary(op, tree(l, lo, mid, op), tree(l, mid, hi, op));
    }
    int32_t map(int32_t i) {
        if (i < 0 || (size_t)i >= remap.size()) return i;
        if (remap[i] != -2) return remap[i];
        int32_t r = i;
        const int c = cls(i);
        const DNode d = g.n[i];
        bool done = false;
        if (c) {
            std::vector<int32_t> l;
            leaves(i, c, true, l);
            if (l.size() >= 4) { r = tree(l, 0, l.size(), (c == 1 || c == 3) ? MARAY_OP_MAX : MARAY_OP_MIN); rebuilt++; done = true; }
        }
        if (!done && d.op < D_CONST) {
            const int32_t a = map(d.a), b = map(d.b);
            if (a != d.a || b != d.b) { DNode e = d; e.a = a; e.b = b; r = g.intern(e); }
        }
        remap[i] = r;
        return r;
    }
};

// Gives every row region its own copy of the x-dependent values it reads.
//
// Hash-consing merges what neighbouring shapes have in common (an edge shared by two triangles,
// `x * 1/w`, ...).  For a region that a y value can switch off for a whole row this is the wrong
// trade: shared values have to be computed ahead of the SKIP op, unconditionally, for every region
// of the image, and they stay live across regions.  Re-deriving them inside the region from X, y
// values and constants costs a few ops in the regions a row does enter and nothing in those it
// skips.  Same ops on the same operands, so every value is unchanged bit for bit.
struct Privatizer {
    Dag &g;
    std::vector<int32_t> &rowub;
    std::vector<uint8_t> is_root, seen;
    std::vector<int32_t> remap;                      // original node -> node after privatisation (-2: not yet)
    IdMap<int32_t> memo;                             // clones of the region being copied
    mutable std::vector<uint32_t> cone_stamp;        // x_cone: the walk a node was last counted in
    mutable uint32_t cone_epoch = 0;
    uint32_t inst = 0;
    size_t budget;

    Privatizer(Dag &g_, std::vector<int32_t> &ub) : g(g_), rowub(ub), is_root(g_.n.size(), 0), seen(g_.n.size(), 0),
                                                     remap(g_.n.size(), -2), budget(4 * g_.n.size() + 4096) {}

    bool x_op(int32_t i) const { return i >= 0 && g.n[i].op < D_CONST && (g.n[i].dep & DEP_X); }

    size_t x_cone(int32_t root, size_t cap) const {  // x-dependent ops below root (root included), counted up to cap
        if (cone_stamp.size() < g.n.size()) cone_stamp.resize(g.n.size(), 0);
        ++cone_epoch;
        size_t in = 0;
        std::vector<int32_t> st{root};
        while (!st.empty() && in <= cap) {
            const int32_t v = st.back(); st.pop_back();
            if (!x_op(v) || cone_stamp[v] == cone_epoch) continue;
            cone_stamp[v] = cone_epoch; in++;
            st.push_back(g.n[v].a); st.push_back(g.n[v].b);
        }
        return in;
    }
    void select(int32_t i) {                         // outermost bounded conjunctions, seen from the channel roots
        if (!x_op(i) || seen[i]) return;
        seen[i] = 1;
        if (rowub[i] >= 0 && g.n[i].op != MARAY_OP_MAX) {    // a shape (conjunction); groups of shapes (OR) stay shared structure
            const size_t sz = x_cone(i, MAX_REGION / 2);
            if (sz >= MIN_ROW_REGION && sz <= MAX_REGION / 2 && sz <= budget) { is_root[i] = 1; budget -= sz; return; }
        }
        select(g.n[i].a); select(g.n[i].b);
    }
    int32_t clone(int32_t i) {
        if (!x_op(i)) return i;                      // leaves, constants and y-only values stay shared
        if (const int32_t *hit = memo.find(i)) return *hit;
        DNode d = g.n[i];
        if (d.a >= 0) d.a = clone(d.a);
        if (d.b >= 0) d.b = clone(d.b);
        d.inst = inst;
        const int32_t r = g.intern(d);
        memo.put(i, r);
        return r;
    }
    int32_t map(int32_t i) {
        if (i < 0 || (size_t)i >= remap.size()) return i;
        if (is_root[i]) {                            // a copy per use: a shape two channels paint belongs to both their trees
            inst++; memo.clear();
            const int32_t r = clone(i);
            rowub.resize(g.n.size(), -1);
            rowub[r] = rowub[i];
            return r;
        }
        if (remap[i] != -2) return remap[i];
        int32_t r = i;
        if (x_op(i)) {
            DNode d = g.n[i];
            const int32_t a = map(d.a), b = map(d.b);
            if (a != d.a || b != d.b) {
                d.a = a; d.b = b; r = g.intern(d);
                rowub.resize(g.n.size(), -1);
                rowub[r] = rowub[i];                 // a group of shapes keeps its bound
            }
        }
        remap[i] = r;
        return r;
    }
};

// One scheduled tape op.
struct SItem {
    int32_t node;          // NODE: the computing node; OUT: the node read; SKIP: the guard
    int32_t out = -1;      // >= 0: OUT op of this output index
    int32_t target = -1;   // >= 0: SKIPZ / SKIPNZ op guarding the region that ends with node `target`
    uint8_t nz = 0;        // SKIP: 0 = SKIPZ (AND), 1 = SKIPNZ (OR)
};

struct Section {
    std::vector<SItem> sched;                   // NODE and SKIP items in schedule order
    std::vector<std::pair<int32_t, uint32_t>> outs;   // (node, output index)
    std::vector<uint64_t> ops;
    uint32_t n_slots = 0;
    uint32_t acc_operands = 0;
    uint32_t n_skips = 0;
};

constexpr size_t MIN_REGION = 6;                // smallest exclusive cone worth a SKIP op
struct Lowerer {
    const Dag &g;
    std::vector<uint8_t> sin_bounded;           // per node: Sin/StepSin argument proven inside reduce_sincos range
    std::vector<uint8_t> isbool;                // per node: value is provably +0.0 or 1.0
    bool regions = true;
    std::vector<int32_t> rowub;                 // per boolean node: y-only upper bound over a row (DAG node) or -1
    std::vector<int32_t> used_rowguards;        // row bounds referenced by SKIP ops, in schedule order
    std::vector<uint8_t> in_section;            // node belongs to the section being built
    std::vector<int32_t> need;                  // Sethi-Ullman label
    std::vector<uint8_t> visited;
    std::vector<uint32_t> users_off;            // consumers inside the section (an OUT counts as user -1), CSR: users of node i
    std::vector<int32_t> users_idx;             // are users_idx[users_off[i] .. users_off[i + 1])
    // membership by stamp instead of hash sets (the scheduler asks for the cone of every AND / OR it meets: chess lowers
    // in a tenth of the time): a node is in the set computed last iff its stamp equals the set's epoch
    static constexpr uint32_t STAMP_OUT = 0xFFFFFFFFu;      // stamp of a node outside the section, or scheduled
    // what reach() reads of a node, twelve bytes instead of the DAG's forty: its operands inside the section (-1: none or outside)
    // and the walk it was last seen in -- the walks are bound by memory, and a large scene's DAG does not fit the cache the nodes' links do
    struct Link { int32_t a, b; uint32_t stamp; };
    std::vector<Link> link;
    std::vector<uint32_t> cone_stamp;
    std::vector<std::pair<int32_t, int>> reach_stack;      // (reach(): node, operands walked)
    std::vector<int32_t> reach_buf;
    uint32_t seen_epoch = 0, cone_epoch = 0;
    bool in_cone(int32_t v) const { return cone_stamp[v] == cone_epoch; }
    std::unordered_map<uint64_t, uint32_t> const_index;
    std::vector<double> consts;
    std::vector<int32_t> yval_of;               // node -> y value index or -1
    int row_depth = 0;
    int32_t row_reentry = -1;

    explicit Lowerer(const Dag &g_) : g(g_), yval_of(g_.n.size(), -1) {}

    uint32_t const_ref(double v) {
        uint64_t b = bits_of(v);
        auto it = const_index.find(b);
        if (it != const_index.end()) return it->second;
        uint32_t i = (uint32_t)consts.size();
        if (i > MARAY_MAX_INDEX) throw Error{MARAY_E_LIMIT, "more than 16384 distinct constants"};
        consts.push_back(v);
        const_index.emplace(b, i);
        return i;
    }

    int32_t su(int32_t i) {
        if (need[i] >= 0) return need[i];
        const DNode &d = g.n[i];
        int32_t na = (d.a >= 0 && in_section[d.a]) ? su(d.a) : 0;
        int32_t nb = (d.b >= 0 && in_section[d.b] && d.b != d.a) ? su(d.b) : 0;
        int32_t r;
        if (!na && !nb) r = 1;
        else if (na == nb) r = na + 1;
        else r = std::max(na, nb);
        need[i] = r;
        return r;
    }

    void prepare(const Section &sec) {
        const size_t N = g.n.size();
        need.assign(N, -1);
        visited.assign(N, 0);
        row_depth = 0;
        used_rowguards.clear();
        cone_stamp.assign(N, 0);
        link.resize(N);
        for (size_t i = 0; i < N; i++) {
            const int32_t a = g.n[i].a, b = g.n[i].b;
            link[i] = Link{a >= 0 && in_section[a] ? a : -1, b >= 0 && in_section[b] ? b : -1, in_section[i] ? 0u : STAMP_OUT};
        }
        seen_epoch = cone_epoch = 0;
        users_off.assign(N + 1, 0);
        for (size_t i = 0; i < N; i++) {
            if (!in_section[i]) continue;
            for (int32_t c : {g.n[i].a, g.n[i].b}) if (c >= 0 && in_section[c]) users_off[c + 1]++;
        }
        for (auto &o : sec.outs) if (in_section[o.first]) users_off[o.first + 1]++;
        for (size_t i = 0; i < N; i++) users_off[i + 1] += users_off[i];
        users_idx.assign(users_off[N], 0);
        std::vector<uint32_t> fill(users_off.begin(), users_off.end() - 1);
        for (size_t i = 0; i < N; i++) {
            if (!in_section[i]) continue;
            for (int32_t c : {g.n[i].a, g.n[i].b}) if (c >= 0 && in_section[c]) users_idx[fill[c]++] = (int32_t)i;
        }
        for (auto &o : sec.outs) if (in_section[o.first]) users_idx[fill[o.first]++] = -1;
    }

    // Unvisited section nodes reachable from `root`, producers before consumers (the post-order of a depth-first walk: a node
    // after everything it reads).  The cones below walk it backwards and need exactly that -- every user of a node before the
    // node -- and are sets, so any such order gives the same cone; sorting the nodes by id (ascending ids are one such
    // order) was an eighth of a large scene's lowering.
    std::vector<int32_t> reach(int32_t root) {
        ++seen_epoch;
        // (stamp = STAMP_OUT while a node is outside the section or scheduled: one load answers "in the section, unscheduled, not seen in this walk")
        auto fresh = [&](int32_t v) { return v >= 0 && link[v].stamp < seen_epoch; };
        if (!fresh(root)) return {};
        reach_buf.clear();                  // (collected in a buffer that keeps its capacity: the result is allocated once, at its size)
        reach_stack.clear();
        reach_stack.push_back({root, 0});
        link[root].stamp = seen_epoch;
        while (!reach_stack.empty()) {
            auto &top = reach_stack.back();
            const int32_t v = top.first;
            if (top.second < 2) {
                const int32_t c = top.second++ == 0 ? link[v].a : link[v].b;
                if (fresh(c)) { link[c].stamp = seen_epoch; reach_stack.push_back({c, 0}); }
                continue;
            }
            reach_buf.push_back(v);
            reach_stack.pop_back();
        }
        return std::vector<int32_t>(reach_buf.begin(), reach_buf.end());
    }

    // The exclusive cone of `body` with respect to its consumer `n`: the nodes of reach(body) that
    // are used by nothing but n or other nodes of the cone, i.e. that become dead when n's value is
    // known without them.  Returns its size; in_cone() tells its members until the next cone is computed.
    size_t exclusive_cone(int32_t body, int32_t n, const std::vector<int32_t> &r) {
        ++cone_epoch;
        size_t size = 0;
        for (auto it = r.rbegin(); it != r.rend(); ++it) {      // consumers before producers
            const int32_t v = *it;
            bool excl = users_off[v + 1] > users_off[v];
            for (uint32_t k = users_off[v]; k < users_off[v + 1] && excl; k++) {
                const int32_t u = users_idx[k];
                if (u == n && v == body) continue;
                if (u < 0 || !in_cone(u)) excl = false;
            }
            if (excl) { cone_stamp[v] = cone_epoch; size++; }
        }
        return size;
    }

    // Nodes of r (= reach(v)) that feed nothing but v: they are dead when v's value is known.
    size_t self_cone(int32_t v, const std::vector<int32_t> &r) {
        ++cone_epoch;
        size_t size = 0;
        for (auto it = r.rbegin(); it != r.rend(); ++it) {
            const int32_t u0 = *it;
            if (u0 == v) { cone_stamp[v] = cone_epoch; size++; continue; }
            bool excl = users_off[u0 + 1] > users_off[u0];
            for (uint32_t k = users_off[u0]; k < users_off[u0 + 1] && excl; k++) {
                const int32_t u = users_idx[k];
                if (u < 0 || !in_cone(u)) excl = false;
            }
            if (excl) { cone_stamp[u0] = cone_epoch; size++; }
        }
        return size;
    }

    int region_kind(int32_t i) const {          // 1: AND (Mul/Min of booleans), 2: OR (Max of booleans)
        const DNode &d = g.n[i];
        if (!isbool[i] || d.a < 0 || d.b < 0 || d.a == d.b || !isbool[d.a] || !isbool[d.b]) return 0;
        if (d.op == MARAY_OP_MUL || d.op == MARAY_OP_MIN) return 1;
        if (d.op == MARAY_OP_MAX) return 2;
        return 0;
    }

    // Schedules, ahead of a region, what the region's cone reads but does not own.  Only the nodes the cone reads
    // directly are visited (each brings its own sub-tree, with the regions that sub-tree deserves): walking every
    // shared node bottom-up would schedule a shared shape's factors one by one and leave nothing to guard at its root.
    // (the cone is the one computed last: in_cone())
    void visit_shared(const std::vector<int32_t> &r, Section &sec) {
        std::vector<int32_t> frontier;
        for (int32_t v : r) {
            if (!in_cone(v)) continue;
            for (int32_t c : {g.n[v].a, g.n[v].b})
                if (c >= 0 && in_section[c] && !visited[c] && !in_cone(c)) frontier.push_back(c);
        }
        std::sort(frontier.begin(), frontier.end());
        frontier.erase(std::unique(frontier.begin(), frontier.end()), frontier.end());
        for (int32_t v : frontier) visit(v, sec);
    }

    void visit(int32_t i, Section &sec) {
        if (i < 0 || visited[i] || !in_section[i]) return;
        // Row-level short circuit: a y value proves this boolean 0 on the whole row -> skip all that only feeds it.
        // not inside a conjunction's row region: bounds of its factors would mostly be true there
        const bool reentry = i == row_reentry;       // the call below that schedules the region's own contents
        row_reentry = -1;
        if (regions && !reentry && !rowub.empty() && rowub[i] >= 0 && row_depth == 0) {
            const std::vector<int32_t> r = reach(i);
            const size_t cone = self_cone(i, r);
            if (cone >= MIN_ROW_REGION && cone <= MAX_REGION / 2) {
                visit_shared(r, sec);                                        // shared nodes stay unconditional
                const size_t mark = sec.sched.size();
                SItem sk; sk.node = rowub[i]; sk.target = i; sk.nz = 0;
                sec.sched.push_back(sk);
                const int conj = g.n[i].op != MARAY_OP_MAX;
                row_depth += conj;
                row_reentry = i;
                visit(i, sec);                                               // the usual schedule, wave-level regions included
                row_depth -= conj;
                if (sec.sched.size() - mark - 1 > MAX_REGION) sec.sched.erase(sec.sched.begin() + (long)mark);
                else used_rowguards.push_back(rowub[i]);
                return;
            }
        }
        const DNode &d = g.n[i];
        const int kind = regions ? region_kind(i) : 0;
        if (kind) {
            // orientation: the operand with the smaller exclusive cone guards the other one
            size_t best = 0;
            int32_t guard = -1, body = -1;
            std::vector<int32_t> r_body;
            for (int o = 0; o < 2; o++) {
                const int32_t bq = o ? d.a : d.b, gq = o ? d.b : d.a;
                if (!in_section[bq] || visited[bq] || g.n[bq].op >= D_CONST) continue;
                std::vector<int32_t> r = reach(bq);
                const size_t sz = exclusive_cone(bq, i, r);
                if (sz >= MIN_REGION && sz > best) { best = sz; guard = gq; body = bq; r_body.swap(r); }
            }
            if (guard >= 0 && g.n[guard].op != D_CONST) {
                visited[i] = 1; link[i].stamp = STAMP_OUT;
                visit(guard, sec);
                if (!visited[body]) {
                    // reach(body) now = what it was before the guard was scheduled, less the nodes scheduled since: a node that
                    // got scheduled took everything it reads with it, so what is left is still reached through unscheduled nodes
                    r_body.erase(std::remove_if(r_body.begin(), r_body.end(), [&](int32_t v) { return visited[v] != 0; }), r_body.end());
                    const std::vector<int32_t> &r = r_body;
                    const size_t cone = exclusive_cone(body, i, r);
                    if (cone >= MIN_REGION && in_cone(body)) {
                        visit_shared(r, sec);                                     // shared nodes: computed unconditionally
                        const size_t mark = sec.sched.size();
                        SItem sk; sk.node = guard; sk.target = i; sk.nz = kind == 2;
                        sec.sched.push_back(sk);
                        visit(body, sec);
                        SItem self; self.node = i;
                        sec.sched.push_back(self);
                        if (sec.sched.size() - mark - 1 > MAX_REGION) sec.sched.erase(sec.sched.begin() + (long)mark);
                        return;
                    }
                }
                visit(body, sec);
                SItem self; self.node = i;
                sec.sched.push_back(self);
                return;
            }
        }
        visited[i] = 1; link[i].stamp = STAMP_OUT;
        int32_t c0 = d.a, c1 = d.b;
        bool h0 = c0 >= 0 && in_section[c0], h1 = c1 >= 0 && in_section[c1];
        if (h0 && h1 && su(c1) > su(c0)) std::swap(c0, c1);   // heavier sub-tree first
        if (c0 >= 0) visit(c0, sec);
        if (c1 >= 0) visit(c1, sec);
        SItem self; self.node = i;
        sec.sched.push_back(self);
    }

    // scheduled: sec.sched holds the section's schedule already (the dry run that numbered the y values made it: the schedule
    // does not depend on their numbers, only the encoding does)
    void build(Section &sec, bool pixel, bool scheduled = false) {
        const size_t N = g.n.size();
        if (!scheduled) {
            prepare(sec);
            for (auto &o : sec.outs) if (in_section[o.first]) visit(o.first, sec);
        }

        // Final item list: OUT ops go right after the node they read (never inside a skipped
        // region: a node with an OUT is not exclusive to anything).
        std::vector<SItem> items;
        {
            std::unordered_map<int32_t, std::vector<uint32_t>> outs_of;
            for (auto &o : sec.outs) outs_of[o.first].push_back(o.second);
            items.reserve(sec.sched.size() + sec.outs.size());
            for (const SItem &it : sec.sched) {
                items.push_back(it);
                if (it.target >= 0) continue;
                auto f = outs_of.find(it.node);
                if (f != outs_of.end()) for (uint32_t k : f->second) { SItem o; o.node = it.node; o.out = (int32_t)k; items.push_back(o); }
            }
            for (auto &o : sec.outs) if (!in_section[o.first]) { SItem x; x.node = o.first; x.out = (int32_t)o.second; items.push_back(x); }
        }
        auto is_node = [](const SItem &it) { return it.out < 0 && it.target < 0; };
        std::vector<int32_t> item_pos(N, -1);
        for (size_t j = 0; j < items.size(); j++) if (is_node(items[j])) item_pos[items[j].node] = (int32_t)j;

        // ACC holds the result of the latest computing item; OUT and SKIP items do not disturb it
        // (a taken SKIP leaves its target's value in ACC, exactly what the target op would have left).
        std::vector<int32_t> acc_holder(items.size(), -1);
        {
            int32_t cur = -1;
            for (size_t j = 0; j < items.size(); j++) {
                acc_holder[j] = cur;
                if (is_node(items[j])) cur = items[j].node;
            }
        }
        std::vector<int32_t> last_use(N, -1), needs_slot(N, 0);
        auto note_use = [&](int32_t c, size_t j) {
            if (c < 0 || !in_section[c]) return;
            last_use[c] = (int32_t)j;
            if (acc_holder[j] != c) needs_slot[c] = 1;   // some use cannot be served by ACC
        };
        for (size_t j = 0; j < items.size(); j++) {
            if (!is_node(items[j])) note_use(items[j].node, j);
            else { note_use(g.n[items[j].node].a, j); note_use(g.n[items[j].node].b, j); }
        }

        // linear scan
        std::vector<int32_t> slot(N, -1);
        std::vector<uint32_t> free_slots;              // min-heap of free slot numbers
        uint32_t next_slot = 0;
        auto alloc = [&]() -> uint32_t {
            if (!free_slots.empty()) {
                std::pop_heap(free_slots.begin(), free_slots.end(), std::greater<uint32_t>());
                uint32_t s = free_slots.back(); free_slots.pop_back(); return s;
            }
            if (next_slot >= MARAY_MAX_SLOTS) throw Error{MARAY_E_LIMIT, "more than 4095 live values"};
            return next_slot++;
        };
        auto release = [&](int32_t c, size_t j) {
            if (c >= 0 && in_section[c] && slot[c] >= 0 && last_use[c] == (int32_t)j) {
                free_slots.push_back((uint32_t)slot[c]);
                std::push_heap(free_slots.begin(), free_slots.end(), std::greater<uint32_t>());
                slot[c] = -2;   // released
            }
        };
        auto opref = [&](int32_t c, size_t j) -> uint32_t {
            const DNode &d = g.n[c];
            if (d.op == D_CONST) return MARAY_REF(MARAY_K_CONST, const_ref(d.cval));
            if (d.op == D_X) return MARAY_REF(MARAY_K_SPEC, MARAY_SPEC_X);
            if (d.op == D_Y) return MARAY_REF(MARAY_K_SPEC, MARAY_SPEC_Y);
            if (d.op == D_XMAX) return MARAY_REF(MARAY_K_SPEC, MARAY_SPEC_XMAX);
            if (d.op == D_XMIN) return MARAY_REF(MARAY_K_SPEC, MARAY_SPEC_XMIN);
            if (d.op == D_YMAX) return MARAY_REF(MARAY_K_SPEC, MARAY_SPEC_YMAX);
            if (d.op == D_YMIN) return MARAY_REF(MARAY_K_SPEC, MARAY_SPEC_YMIN);
            if (in_section[c]) {
                if (acc_holder[j] == c) { sec.acc_operands++; return MARAY_REF(MARAY_K_SPEC, MARAY_SPEC_ACC); }
                if (slot[c] < 0) throw Error{MARAY_E_INTERNAL, "operand without a slot"};
                return MARAY_REF(MARAY_K_SLOT, (uint32_t)slot[c]);
            }
            if (pixel && yval_of[c] >= 0) {
                if ((uint32_t)yval_of[c] > MARAY_MAX_INDEX) throw Error{MARAY_E_LIMIT, "more than 16384 row values"};
                return MARAY_REF(MARAY_K_YVAL, (uint32_t)yval_of[c]);
            }
            throw Error{MARAY_E_INTERNAL, "operand outside its section"};
        };
        for (size_t j = 0; j < items.size(); j++) {
            const SItem &it = items[j];
            if (it.out >= 0) {
                uint32_t a = opref(it.node, j);
                release(it.node, j);
                sec.ops.push_back(MARAY_INS(MARAY_OP_OUT, (uint32_t)it.out, MARAY_DST_NONE, a, 0));
                continue;
            }
            if (it.target >= 0) {
                // the region's result slot is reserved here so that both paths define it
                const uint32_t a = opref(it.node, j);
                release(it.node, j);
                uint32_t dst = MARAY_DST_NONE;
                if (slot[it.target] >= 0) dst = (uint32_t)slot[it.target];      // reserved by an enclosing region with the same end
                else if (needs_slot[it.target]) { dst = alloc(); slot[it.target] = (int32_t)dst; }
                const uint32_t count = (uint32_t)(item_pos[it.target] - (int32_t)j);
                sec.ops.push_back(MARAY_INS(it.nz ? MARAY_OP_SKIPNZ : MARAY_OP_SKIPZ, count, dst, a, 0));
                sec.n_skips++;
                continue;
            }
            const DNode &d = g.n[it.node];
            uint32_t a = d.a >= 0 ? opref(d.a, j) : 0;
            uint32_t b = d.b >= 0 ? opref(d.b, j) : 0;
            release(d.a, j);
            if (d.b != d.a) release(d.b, j);
            uint32_t dst = MARAY_DST_NONE;
            if (slot[it.node] >= 0) dst = (uint32_t)slot[it.node];            // reserved by this region's SKIP op
            else if (needs_slot[it.node]) { dst = alloc(); slot[it.node] = (int32_t)dst; }
            if (d.aux > 0x1FFFu) throw Error{MARAY_E_LIMIT, "App id above 8191"};
            uint32_t aux = d.aux;
            if ((d.op == MARAY_OP_SIN || d.op == MARAY_OP_STEPSIN) && sin_bounded[it.node]) aux |= MARAY_AUX_SIN_BOUNDED;
            sec.ops.push_back(MARAY_INS(d.op, aux, dst, a, b));
        }
        sec.n_slots = next_slot;
    }
};

}   // namespace

void lower_scene(const Scene &scene_in, const maray_lower_opts &opts, Tape &t)
{
    // MARAY_TRACE_LOWER=1: where the time of a lowering goes (stderr)
    const bool trace = getenv("MARAY_TRACE_LOWER") && getenv("MARAY_TRACE_LOWER")[0] == '1';
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "maray lower: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const Scene s = scene_fixed(scene_in);
    lap("fix_color");

    Dag g;
    g.commute = opts.plain_cse == 0;
    g.fuse = opts.no_fuse == 0;
    SymEval ev(s, g);
    int32_t roots[3];
    for (int c = 0; c < 3; c++) roots[c] = ev.eval(s.color[c], -1);   // outer Context::new() is empty (src/render.rs:53)
    lap("symbolic evaluation");

    // Boolean typing, intervals, monotonicity on the scene's own DAG; then (optionally) the row bounds,
    // which append y-only nodes to the DAG, and the typing / intervals once more over the grown DAG.
    auto bool_typing = [&]() {
        std::vector<uint8_t> isb(g.n.size(), 0);
        for (size_t i = 0; i < g.n.size(); i++) {                  // values that are provably +0.0 or 1.0
            const DNode &d = g.n[i];
            auto is_one = [&](int32_t c) { return c >= 0 && g.n[c].op == D_CONST && bits_of(g.n[c].cval) == 0x3ff0000000000000ull; };
            auto is_negbool = [&](int32_t c) { return c >= 0 && g.n[c].op == MARAY_OP_NEG && isb[g.n[c].a]; };
            switch (d.op) {
            case D_CONST: isb[i] = bits_of(d.cval) == 0 || bits_of(d.cval) == 0x3ff0000000000000ull; break;
            case MARAY_OP_STEP: case MARAY_OP_STEPSIN: isb[i] = 1; break;
            case MARAY_OP_MUL: case MARAY_OP_MIN: case MARAY_OP_MAX: isb[i] = isb[d.a] && isb[d.b]; break;
            case MARAY_OP_ADD: isb[i] = (is_one(d.a) && is_negbool(d.b)) || (is_one(d.b) && is_negbool(d.a)); break;   // 1 + -(b) = NOT b
            default: break;
            }
        }
        return isb;
    };
    const bool row_guards = opts.no_skips == 0 && opts.hoist_rows != 0 && opts.no_row_guards == 0;
    uint32_t rebalanced = 0;
    if (row_guards && opts.no_rebalance == 0) {
        const std::vector<uint8_t> isb = bool_typing();
        Rebalancer rb(g, isb, roots);
        for (int c = 0; c < 3; c++) roots[c] = rb.map(roots[c]);
        rebalanced = rb.rebuilt;
    }
    lap("rebalance");
    const size_t N0 = g.n.size();
    if (row_guards) g.reserve(4 * N0 + 1024);      // the row bounds grow the DAG about threefold
    const uint32_t folded_scene = g.folded;     // constant ops folded in the scene itself (the row bounds fold more)
    std::vector<int32_t> rowub;
    std::vector<Ival> iv_known;                 // intervals of the DAG's first nodes, for the passes that follow its growth
    if (row_guards) {
        const std::vector<uint8_t> isb0 = bool_typing();
        const std::vector<Ival> iv0 = intervals(g);
        const std::vector<Mono> mono0 = monotonicity(g, iv0);
        lap("  typing, intervals, monotone");
        RowBounds rb(g, isb0, mono0, iv0, DEP_X, D_X, D_XMIN, D_XMAX);
        rowub.assign(N0, -1);
        // sign bit provably clear and no NaN (so that "+0.0" statements compose), and finite on top
        std::vector<uint8_t> clear(N0, 0), finite(N0, 0);
        for (size_t i = 0; i < N0; i++) {
            const DNode &d = g.n[i];
            const bool ca = d.a >= 0 && clear[d.a], cb = d.b >= 0 && clear[d.b];
            bool c = false;
            switch (d.op) {
            case D_CONST: c = d.cval == d.cval && !std::signbit(d.cval); break;
            case D_X: case D_Y: case D_XMAX: case D_XMIN: case D_YMAX: case D_YMIN: c = true; break;
            case MARAY_OP_STEP: case MARAY_OP_STEPSIN: case MARAY_OP_APP: case MARAY_OP_TEXDIM: c = true; break;
            case MARAY_OP_ABS: c = !iv0[d.a].nan; break;                        // |-0| = +0
            case MARAY_OP_SQRT: case MARAY_OP_MOV: c = ca; break;                // sqrt(+0) = +0
            case MARAY_OP_ADD: case MARAY_OP_MUL: case MARAY_OP_MIN: case MARAY_OP_MAX: c = ca && cb; break;
            default: break;
            }
            clear[i] = isb0[i] || (c && !iv0[i].nan);      // a boolean is +0.0 or 1.0 whatever it is made of (1 + -(b) passes through -1)
            finite[i] = clear[i] && iv0[i].hi < INFINITY;
        }
        std::vector<uint32_t> shapes(N0, 1);          // operands of the max tree below a node
        for (size_t i = 0; i < N0; i++) {
            const uint8_t op = g.n[i].op;
            if (!(g.n[i].dep & DEP_X) || g.n[i].a < 0 || g.n[i].b < 0) continue;
            if (op == MARAY_OP_MAX) shapes[i] = shapes[g.n[i].a] + shapes[g.n[i].b];
            // conjunctions (a shape), shapes times a colour, and max over a few shapes (two levels of a balanced tree:
            // groups of 3-4 and of 9-16); a max accumulating many shapes has a bound that is almost always true
            const bool conj = op == MARAY_OP_MUL || op == MARAY_OP_MIN;
            const bool group = op == MARAY_OP_MAX && ((shapes[i] >= 3 && shapes[i] <= 4) || (shapes[i] >= 9 && shapes[i] <= 16));
            if (!conj && !group) continue;
            const int32_t ub = isb0[i] ? rb.bounds((int32_t)i).ub : rb.zub((int32_t)i, clear, finite);
            if (g.n[ub].op < D_CONST) rowub[i] = ub;          // a real y-only op (not folded to a constant)
        }
        // Second pass, over y: a guard that is built from booleans monotone in y as well is bounded over the rows
        // [YMIN, YMAX] the same way, and then holds for a rectangle of pixels -- an evaluator may compute it once for
        // several rows.  A guard that is not keeps reading Y: exact for its row, valid for that row only.
        lap("  bounds over x");
        if (trace) rb.trace("x");
        if (opts.no_y_spans == 0) {
            const std::vector<uint8_t> isb1 = bool_typing();
            const std::vector<Ival> iv1 = intervals(g, &iv0);
            const std::vector<Mono> mono_y = monotonicity(g, iv1, DEP_Y, D_Y);
            RowBounds rby(g, isb1, mono_y, iv1, DEP_Y, D_Y, D_YMIN, D_YMAX);
            // A lift that had to give up on some sub-expression (`lossy`: a y-only Step(Sin) of a pattern, say, bounded by
            // "anything") is weaker than the guard it comes from, which is exact for its row; one that gave up on all of it is
            // the constant 1.  The evaluators bound guards per rectangle of pixels only when NO guard of the program reads Y
            // (any_guard_reads_y), per row x span otherwise -- 32 times the guard work and none of the specialised kernel's
            // rectangle machinery.  So when most guards that read Y have a lift worth having, every one of them takes its
            // lift, exact or not, and the few that have none are dropped (a guard is an optimisation: without it the region
            // is simply evaluated); otherwise only exact lifts are taken, as the guards stay per row anyway.
            std::vector<std::pair<size_t, RowBounds::B>> lifts;
            size_t n_real = 0, n_none = 0;
            for (size_t i = 0; i < N0; i++) {
                if (rowub[i] < 0 || !(g.n[rowub[i]].dep & DEP_Y)) continue;
                const RowBounds::B u = rby.bounds(rowub[i]);
                lifts.push_back({i, u});
                (g.n[u.ub].op < D_CONST ? n_real : n_none)++;
            }
            const bool all_free = n_real > 0 && n_real >= n_none;
            if (trace && getenv("MARAY_TRACE_GUARDS")) {
                std::function<std::string(int32_t, int)> show = [&](int32_t i, int depth) -> std::string {
                    const DNode &d = g.n[i];
                    static const char *leaf[] = {"", "X", "Y", "XMAX", "XMIN", "YMAX", "YMIN"};
                    static const char *opn[] = {"nop", "mov", "neg", "abs", "recip", "sqrt", "step", "sin", "exp", "ln", "add", "mul", "max", "min", "app", "texdim", "out", "stepsin"};
                    if (d.op == D_CONST) { char b[32]; snprintf(b, sizeof b, "%g", d.cval); return b; }
                    if (d.op > D_CONST) return leaf[d.op - D_CONST];
                    if (depth <= 0) return "...";
                    std::string r = std::string("(") + opn[d.op] + " " + show(d.a, depth - 1);
                    if (d.b >= 0) r += " " + show(d.b, depth - 1);
                    return r + ")";
                };
                for (auto &l : lifts) fprintf(stderr, "maray lower: node %zu guard %s\n   lift %s lossy %d\n", l.first, show(rowub[l.first], 12).c_str(), show(l.second.ub, 12).c_str(), (int)l.second.lossy);
            }
            if (trace) fprintf(stderr, "maray lower: guards over y: %zu lifted, %zu without a lift -> %s\n", n_real, n_none, all_free ? "rectangles" : "rows");
            for (auto &l : lifts) {
                if (g.n[l.second.ub].op < D_CONST) { if (!l.second.lossy || all_free) rowub[l.first] = l.second.ub; }
                else if (all_free) rowub[l.first] = -1;
            }
            if (trace) rby.trace("y");
            iv_known = iv1;
        } else iv_known = iv0;
        rowub.resize(g.n.size(), -1);
    }
    lap("row bounds");

    auto mark_reach = [&]() {                   // reachability from the channel roots
        std::vector<uint8_t> r(g.n.size(), 0);
        std::vector<int32_t> st(roots, roots + 3);
        while (!st.empty()) {
            int32_t i = st.back(); st.pop_back();
            if (r[i]) continue;
            r[i] = 1;
            if (g.n[i].a >= 0) st.push_back(g.n[i].a);
            if (g.n[i].b >= 0) st.push_back(g.n[i].b);
        }
        return r;
    };
    auto is_op = [&](int32_t i) { return g.n[i].op < D_CONST; };

    // census of the scene's own DAG (before row regions get private copies of shared values)
    maray_tape_info &info = t.info;
    memset(&info, 0, sizeof info);
    info.folded_ops = folded_scene;
    uint32_t max_app = 0;
    std::vector<uint8_t> reach = mark_reach();
    for (size_t i = 0; i < g.n.size(); i++) {
        if (!reach[i]) continue;
        info.dag_nodes++;
        if (!is_op((int32_t)i)) continue;
        const uint32_t cnt = g.n[i].op == MARAY_OP_STEPSIN ? 2u : 1u;   // a fused op stands for two Expr ops
        info.alg_ops += cnt;
        switch (g.n[i].dep) {
        case 0: info.alg_ops_uniform += cnt; break;
        case DEP_X: info.alg_ops_x += cnt; break;
        case DEP_Y: info.alg_ops_y += cnt; break;
        default: info.alg_ops_xy += cnt;
        }
        if (g.n[i].op == MARAY_OP_APP || g.n[i].op == MARAY_OP_TEXDIM) max_app = std::max(max_app, g.n[i].aux + 1);
    }
    info.n_app = max_app;
    info.rebalanced_chains = rebalanced;

    if (!rowub.empty() && opts.no_private_regions == 0) {
        Privatizer pv(g, rowub);
        for (int c = 0; c < 3; c++) pv.select(roots[c]);
        for (int c = 0; c < 3; c++) roots[c] = pv.map(roots[c]);
        rowub.resize(g.n.size(), -1);
        info.private_regions = pv.inst;
        reach = mark_reach();
    }
    lap("census, private regions");
    const size_t N = g.n.size();

    Lowerer L(g);
    L.regions = opts.no_skips == 0;
    L.isbool = bool_typing();
    L.rowub = rowub;
    {
        const std::vector<Ival> iv = intervals(g, &iv_known);
        L.sin_bounded.assign(N, 0);
        for (size_t i = 0; i < N; i++) {
            if (!reach[i] || (g.n[i].op != MARAY_OP_SIN && g.n[i].op != MARAY_OP_STEPSIN)) continue;
            const Ival &a = iv[g.n[i].a];
            if (!a.nan && std::max(std::fabs(a.lo), std::fabs(a.hi)) < 105414350.0) { L.sin_bounded[i] = 1; info.sin_bounded++; }
            info.sin_ops++;
        }
    }
    const bool hoist = opts.hoist_rows != 0;
    // ROW section: reachable ops that do not depend on X.
    Section row, pix;
    std::vector<uint8_t> is_row(N, 0);
    if (hoist) for (size_t i = 0; i < N; i++) if (reach[i] && is_op((int32_t)i) && !(g.n[i].dep & DEP_X)) is_row[i] = 1;
    // y values = ROW nodes consumed by PIXEL ops or by a channel output
    {
        std::vector<uint8_t> frontier(N, 0);
        for (size_t i = 0; i < N; i++) {
            if (!reach[i] || !is_op((int32_t)i) || is_row[i]) continue;
            for (int32_t c : {g.n[i].a, g.n[i].b}) if (c >= 0 && is_row[c]) frontier[c] = 1;
        }
        for (int c = 0; c < 3; c++) if (is_row[roots[c]]) frontier[roots[c]] = 1;
        // Number the y values in the order the PIXEL schedule first reads them, so that consecutive
        // reads hit consecutive table entries (the specialised kernel fetches them in 64-byte scalar
        // loads; scattered indices would keep whole batches parked in SGPRs).
        std::vector<uint8_t> is_pix0(N, 0);
        for (size_t i = 0; i < N; i++) if (reach[i] && is_op((int32_t)i) && !is_row[i]) is_pix0[i] = 1;
        Section dry;
        for (int c = 0; c < 3; c++) dry.outs.push_back({roots[c], (uint32_t)c});
        L.in_section = is_pix0;
        L.prepare(dry);
        for (int c = 0; c < 3; c++) if (is_pix0[roots[c]]) L.visit(roots[c], dry);
        // Row bounds the schedule decided to use: they (and what they are computed from) join the ROW
        // section, and each becomes a y value.
        for (int32_t rg : L.used_rowguards) {
            std::vector<int32_t> st{rg};
            while (!st.empty()) {
                const int32_t v = st.back(); st.pop_back();
                if (v < 0 || (reach[v] && (is_row[v] || !is_op(v)))) continue;
                reach[v] = 1;
                if (is_op(v)) is_row[v] = 1;
                st.push_back(g.n[v].a); st.push_back(g.n[v].b);
            }
            frontier[rg] = 1;
        }
        uint32_t k = 0;
        auto number = [&](int32_t c) {
            if (c >= 0 && frontier[c] && L.yval_of[c] < 0) { L.yval_of[c] = (int32_t)k; row.outs.push_back({c, k}); k++; }
        };
        // arithmetic operands first (the specialised kernel stages exactly this prefix in LDS) ...
        for (const SItem &it : dry.sched) if (it.target < 0) { number(g.n[it.node].a); number(g.n[it.node].b); }
        for (int c = 0; c < 3; c++) number(roots[c]);
        // ... then the y values that are only ever SKIP guards (row bounds): read as scalars
        for (const SItem &it : dry.sched) if (it.target >= 0) number(it.node);
        for (size_t i = 0; i < N; i++) number((int32_t)i);
        info.n_yvals = k;
        if (k > MARAY_MAX_INDEX + 1) throw Error{MARAY_E_LIMIT, "more than 16384 row values"};
        pix.sched = std::move(dry.sched);           // = the PIXEL section's schedule (the nodes of the section are the same: the guards' cones joined the ROW section)
    }
    lap("typing, dry schedule");
    std::vector<uint8_t> is_pix(N, 0);
    for (size_t i = 0; i < N; i++) if (reach[i] && is_op((int32_t)i) && !is_row[i]) is_pix[i] = 1;
    for (int c = 0; c < 3; c++) pix.outs.push_back({roots[c], (uint32_t)c});
    // The two sections are scheduled side by side (each is a third of a large scene's lowering; a first render call is
    // as long as its lowering): the ROW section on a copy of the scheduler, on a thread of its own.  What they share is
    // the constant pool, in which the ROW section's constants come first: the PIXEL section numbers its new constants
    // from the pool as it stood before either, and is renumbered once the ROW section's are known -- the tape is the one
    // the sections give one after the other (MARAY_LOWER_SERIAL=1 does that; a CPU test compares the two).
    const char *serial = getenv("MARAY_LOWER_SERIAL");
    if ((serial && serial[0] == '1') || row.outs.size() + (size_t)std::count(is_row.begin(), is_row.end(), 1) < 2000) {
        L.in_section = is_row;
        L.build(row, false);
        lap("ROW section");
        L.in_section = is_pix;
        pix.sched.clear();                      // (serial mode schedules the section again: the reference the reuse below is tested against)
        L.build(pix, true);
        lap("PIXEL section");
    } else {
        const size_t base = L.consts.size();
        Lowerer LR = L;
        // (a thread with the stack the scheduler's recursion needs: this function itself runs on one of 1 GiB, api.cpp)
        struct RowJob {
            Lowerer &LR; Section &row; const std::vector<uint8_t> &is_row; std::exception_ptr error;
            static void *main(void *p) {
                RowJob *j = (RowJob *)p;
                try { j->LR.in_section = j->is_row; j->LR.build(j->row, false); } catch (...) { j->error = std::current_exception(); }
                return nullptr;
            }
        } job{LR, row, is_row, nullptr};
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setstacksize(&at, (size_t)1 << 30);
        pthread_t th;
        const bool threaded = pthread_create(&th, &at, RowJob::main, &job) == 0;
        pthread_attr_destroy(&at);
        struct Join { pthread_t &t; bool on; ~Join() { if (on) pthread_join(t, nullptr); } } join{th, threaded};
        if (!threaded) RowJob::main(&job);
        L.in_section = is_pix;
        L.build(pix, true, /*scheduled=*/true);
        if (threaded) { pthread_join(th, nullptr); join.on = false; }
        if (job.error) std::rethrow_exception(job.error);
        // pool = [before | ROW's new | PIXEL's new that the ROW section did not bring]
        std::vector<uint32_t> remap(L.consts.size());
        for (size_t i = 0; i < L.consts.size(); i++) remap[i] = i < base ? (uint32_t)i : LR.const_ref(L.consts[i]);
        auto fix = [&](uint32_t r) { return MARAY_REF_KIND(r) == MARAY_K_CONST ? MARAY_REF(MARAY_K_CONST, remap[MARAY_REF_INDEX(r)]) : r; };
        for (uint64_t &ins : pix.ops) {
            const uint32_t op = MARAY_INS_OP(ins);
            if (op == MARAY_OP_NOP || op == MARAY_OP_TEXDIM) continue;
            const bool binary = op >= MARAY_OP_ADD && op <= MARAY_OP_APP;
            ins = MARAY_INS(op, MARAY_INS_AUX(ins), MARAY_INS_DST(ins), fix(MARAY_INS_A(ins)), binary ? fix(MARAY_INS_B(ins)) : MARAY_INS_B(ins));
        }
        L.consts = std::move(LR.consts);
        lap("ROW and PIXEL sections");
    }

    t.consts = std::move(L.consts);
    if (t.consts.empty()) t.consts.push_back(0.0);
    t.row_ops = std::move(row.ops);
    t.pix_ops = std::move(pix.ops);
    info.n_consts = (uint32_t)t.consts.size();
    info.n_row_ops = (uint32_t)t.row_ops.size();
    info.n_row_slots = row.n_slots;
    info.n_pix_ops = (uint32_t)t.pix_ops.size();
    info.n_pix_slots = pix.n_slots;
    info.acc_operands = pix.acc_operands;
    info.skip_ops = pix.n_skips;
    for (size_t i = 0; i < N; i++) if (is_pix[i] && L.isbool[i]) info.bool_ops++;
    for (uint64_t ins : t.pix_ops) info.op_histogram[MARAY_INS_OP(ins)]++;
}

}   // namespace maray
