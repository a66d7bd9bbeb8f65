// compress.cpp — `Expr::compress` of the reference, restated on the arena data model (product code, host side).
//
// SURVEY.md §8(f) N4, second half: the authoring-time pass that names repeated sub-expressions
// (`examples/chess.rs:43`: `shape.simplify(mem).compress(mem)` before `save`).
//
//   expr_compress      <- Expr::compress                    src/lib.rs:610-614
//   flatten            <- compressor::flatten               src/compressor.rs:167-211   (Let / Var -> shared Arc sub-trees)
//   compress loop      <- compressor::compress              src/compressor.rs:214-236
//   count / Terms      <- Compressor::count_expr, from_expr src/compressor.rs:22-82
//   last_max_benefit   <- Compressor::last_max_benefit      src/compressor.rs:85-107
//   is_simple          <- is_simple_expr                    src/compressor.rs:111-121
//   is_compressed      <- is_compressed, IsCompressedCache  src/compressor.rs:124-155
//   benefit            <- compression_benefit               src/compressor.rs:158-164
//   display_len        <- `format!("{}", expr).chars().count()`, impl Display for Expr   src/lib.rs:196-366
//   rewrite            <- Expr::rewrite                     src/lib.rs:560-598
//   Mem                <- MemoryManager::{get, get_map, set_map, get_count}             src/memory_manager.rs
//
// The pass is greedy and its choices are driven by the LENGTH OF THE PRINTED FORM of every candidate term, so the
// restatement keeps the reference's printing rules (only their lengths are needed), the order in which terms are
// met and counted, the "last of the equally good" choice, and the side effects of its caches (`from_expr` clears the
// rewrite map; the counts of an `Arc`'s contents are computed once per distinct contents): its result is data, a file
// written with it is what the reference would write.  Structural equality (`==` on `Expr`, used by every lookup of
// the reference) is class identity here: every node gets the id of its hash-consed structure once.
// It changes no value: the compressed expression evaluates bit for bit like the original (tests/test_compress.py).
// Not on the render path: nothing here runs per pixel.
#include <cstdint>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "expr.hpp"
#include "maray_hip.h"

namespace maray {

namespace {

struct Comp {
    Scene &s;
    explicit Comp(Scene &s_) : s(s_) {}

    // ---- structural classes: cls(x) == cls(y)  <=>  x == y (derived PartialEq of Expr) -------------------------------
    std::vector<int32_t> cls_of;                         // per node, -1 = not yet
    std::unordered_map<std::string, int32_t> intern;
    std::vector<int32_t> rep;                            // class -> a node of it
    static void put(std::string &k, uint64_t v) { k.append((const char *)&v, 8); }

    int32_t cls(int32_t e) {
        if ((size_t)e < cls_of.size() && cls_of[e] >= 0) return cls_of[e];
        const Node n = s.nodes[e];
        std::string k(1, (char)n.tag);
        switch (n.tag) {
        case T_X: case T_Y: case T_TAU: case T_E: break;
        case T_VAR: case T_NAT: put(k, n.u); break;
        case T_APP: put(k, n.app); put(k, (uint64_t)cls(n.a)); put(k, (uint64_t)cls(n.b)); break;
        case T_LET: {
            const Ctx c = s.ctxs[n.ctx];
            put(k, c.ids.size());
            for (size_t i = 0; i < c.ids.size(); i++) { put(k, c.ids[i]); put(k, (uint64_t)cls(c.defs[i])); }
            put(k, (uint64_t)cls(n.a));
            break;
        }
        case T_DECOR: {
            put(k, (uint64_t)cls(n.a));
            const std::vector<Token> t = s.toklists[n.toks];
            put(k, t.size());
            for (const Token &tk : t) {
                put(k, tk.kind);
                if (tk.kind == 0) put(k, (uint64_t)cls(tk.expr));
                if (tk.kind == 1) { put(k, tk.str.size()); k += tk.str; }
            }
            break;
        }
        default:
            put(k, (uint64_t)cls(n.a));
            if (is_binary(n.tag)) put(k, (uint64_t)cls(n.b));
        }
        auto it = intern.find(k);
        int32_t c;
        if (it != intern.end()) c = it->second;
        else { c = (int32_t)rep.size(); rep.push_back(e); intern.emplace(std::move(k), c); }
        if (cls_of.size() < s.nodes.size()) cls_of.resize(s.nodes.size(), -1);
        cls_of[e] = c;
        return c;
    }
    bool equal(int32_t x, int32_t y) { return cls(x) == cls(y); }

    int32_t mk(uint8_t tag, int32_t a = -1, int32_t b = -1, uint64_t u = 0, uint32_t app = 0) {
        Node n; n.tag = tag; n.a = a; n.b = b; n.u = u; n.app = app;
        return s.add(n);
    }
    uint8_t tag(int32_t e) const { return s.nodes[e].tag; }

    // ---- Display, as a length in chars (src/lib.rs:196-366) ----------------------------------------------------------
    std::unordered_map<int32_t, uint64_t> len_memo;
    static uint64_t digits(uint64_t v) { uint64_t d = 1; while (v >= 10) { v /= 10; d++; } return d; }
    static uint64_t utf8_chars(const std::string &t) { uint64_t n = 0; for (unsigned char ch : t) n += (ch & 0xC0) != 0x80; return n; }
    bool needs_parens(int32_t e) const {                // src/lib.rs:390-401
        switch (tag(e)) {
        case T_X: case T_Y: case T_TAU: case T_E: case T_VAR: case T_NAT: case T_ABS: case T_SIN: case T_STEP: case T_SQRT:
        case T_EXP: case T_LN: case T_MIN: case T_MAX: return false;
        default: return true;
        }
    }
    bool is_recip(int32_t e) const { return tag(e) == T_RECIP; }
    bool is_div(int32_t e) const { return tag(e) == T_MUL && tag(s.nodes[e].b) == T_RECIP; }
    bool is_sub(int32_t e) const { return tag(e) == T_ADD && tag(s.nodes[e].b) == T_NEG; }
    bool is_square(int32_t e) { return tag(e) == T_MUL && equal(s.nodes[e].a, s.nodes[e].b); }
    uint64_t wrapped(int32_t e, bool parens) { return display_len(e) + (parens ? 2 : 0); }

    uint64_t display_len(int32_t e) {
        const int32_t c = cls(e);
        auto it = len_memo.find(c);
        if (it != len_memo.end()) return it->second;
        const Node n = s.nodes[e];
        uint64_t r = 0;
        switch (n.tag) {
        case T_ARC: r = display_len(n.a); break;
        case T_X: case T_Y: case T_TAU: case T_E: r = 1; break;                       // "x" "y" "τ" "𝐞": one char each
        case T_VAR: r = 1 + digits(n.u); break;                                       // "${}"
        case T_NAT: r = digits(n.u); break;
        case T_NEG: r = needs_parens(n.a) ? 3 + display_len(n.a) : 1 + display_len(n.a); break;      // "-({})" / "-{}"
        case T_ABS: r = 5 + display_len(n.a); break;                                  // "abs({})"
        case T_RECIP: r = needs_parens(n.a) ? 4 + display_len(n.a) : 2 + display_len(n.a); break;    // "1/({})" / "1/{}"
        case T_SQRT: r = 6 + display_len(n.a); break;
        case T_STEP: r = 6 + display_len(n.a); break;
        case T_SIN: r = 5 + display_len(n.a); break;
        case T_EXP: r = 4 + display_len(n.a); break;                                  // "𝐞^({})"
        case T_LN: r = 4 + display_len(n.a); break;
        case T_ADD:
            if (tag(n.b) == T_NEG) {                                                  // :230-252  a - b
                const int32_t b = s.nodes[n.b].a;
                r = wrapped(n.a, needs_parens(n.a) && !is_recip(n.a) && !is_div(n.a) && !is_sub(n.a) && !is_square(n.a) && tag(n.a) != T_MUL);
                r += 1;
                r += wrapped(b, needs_parens(b) && !is_recip(b) && !is_div(b) && !is_square(b));
            } else {                                                                  // :253-276  a + b
                r = wrapped(n.a, needs_parens(n.a) && !is_recip(n.a) && !is_div(n.a) && !is_sub(n.a) && !is_square(n.a));
                r += 1;
                r += wrapped(n.b, needs_parens(n.b) && !is_recip(n.b) && !is_div(n.b) && !is_square(n.b));
            }
            break;
        case T_MUL:
            if (tag(n.b) == T_RECIP) {                                                // :279-291  a / b
                const int32_t b = s.nodes[n.b].a;
                r = wrapped(n.a, needs_parens(n.a)) + 1 + wrapped(b, needs_parens(b));
            } else if (is_square(e)) r = wrapped(n.a, needs_parens(n.a)) + 2;         // :293-299  a^2
            else r = wrapped(n.a, needs_parens(n.a)) + 1 + wrapped(n.b, needs_parens(n.b));          // :300-312
            break;
        case T_MAX: case T_MIN: r = 6 + display_len(n.a) + display_len(n.b); break;   // "max({},{})"
        case T_LET: {                                                                 // :318-324
            const Ctx c = s.ctxs[n.ctx];
            r = display_len(n.a) + 7;                                                 // "{}\nwhere\n"
            for (size_t i = 0; i < c.ids.size(); i++) r += 3 + digits(c.ids[i]) + 3 + display_len(c.defs[i]) + 1;   // "  ${} = {}\n"
            break;
        }
        case T_DECOR: {                                                               // :325-363
            r = display_len(n.a) + (needs_parens(n.a) ? 5 : 3);                       // "({}) : " / "{} : "
            const std::vector<Token> toks = s.toklists[n.toks];
            int64_t tabs = 0;
            bool last_start = false;
            for (const Token &t : toks) {
                switch (t.kind) {
                case 0: r += display_len(t.expr); break;
                case 1: r += utf8_chars(t.str); break;
                case 2: case 3: case 4: case 5: case 6: case 7: case 8: case 9: r += 1; break;      // brackets, comma, space
                case 10: if (last_start) tabs++; r += 1; break;                       // NewLine
                case 11: r += 2 * (uint64_t)(tabs > 0 ? tabs : 0); break;              // Tabs
                case 12: tabs--; r += 2 * (uint64_t)(tabs > 0 ? tabs : 0); break;      // TabsPrev
                default: break;
                }
                last_start = t.kind == 2 || t.kind == 4 || t.kind == 6;               // Token::is_start_bracket
            }
            break;
        }
        case T_APP: r = 4 + digits(n.app) + 1 + display_len(n.a) + 1 + display_len(n.b) + 1; break;   // "app({},{},{})"
        default: throw Error{MARAY_E_INTERNAL, "display_len: bad tag"};
        }
        len_memo.emplace(c, r);
        return r;
    }

    // ---- compression_benefit (src/compressor.rs:158-164) -------------------------------------------------------------
    uint64_t benefit(int32_t e, uint64_t count, uint64_t var_len) {
        const uint64_t len = display_len(e);
        const uint64_t cost = var_len + 3 + len + 3;
        if (var_len > len) return 0;
        const uint64_t main = (len - var_len) * count;
        if (cost > main) return 0;
        return main - cost;
    }

    bool is_simple(int32_t e, int level) const {         // :111-121
        switch (tag(e)) {
        case T_X: case T_Y: case T_TAU: case T_E: case T_NAT: case T_VAR: return true;
        case T_RECIP: return level >= 1 && is_simple(s.nodes[e].a, level - 1);
        case T_MUL: return level >= 2 && is_simple(s.nodes[e].a, level - 1) && is_simple(s.nodes[e].b, level - 1);
        default: return false;
        }
    }

    // ---- MemoryManager (src/memory_manager.rs): keyed by value = by class --------------------------------------------
    struct Terms {                                       // Compressor { terms: Vec<(Expr, usize)> }: insertion order matters
        std::vector<std::pair<int32_t, uint64_t>> v;     // (a node of the term's class, count)
        std::unordered_map<int32_t, size_t> at;          // class -> index in v
        void add(Comp &C, int32_t e, uint64_t n) {
            const int32_t c = C.cls(e);
            auto it = at.find(c);
            if (it != at.end()) v[it->second].second += n;
            else { at.emplace(c, v.size()); v.emplace_back(e, n); }
        }
        void clear() { v.clear(); at.clear(); }
    };
    std::unordered_map<int32_t, int32_t> map;            // class of an Arc's contents -> node it was rewritten to
    std::unordered_map<int32_t, Terms> count;            // class of an Arc's contents -> its counted terms

    bool get_map(int32_t inner, int32_t &out) {          // follows the chain (:49-55)
        bool found = false;
        int32_t cur = inner;
        for (;;) {
            auto it = map.find(cls(cur));
            if (it == map.end()) break;
            cur = it->second; found = true;
        }
        out = cur;
        return found;
    }

    void count_expr(Terms &t, int32_t e) {               // Compressor::count_expr (:33-82)
        if (is_simple(e, 2)) return;
        const Node n = s.nodes[e];
        if (n.tag == T_ARC) {
            const int32_t ic = cls(n.a);
            if (!count.count(ic)) {                      // MemoryManager::get_count -> Compressor::from_expr (clears the map)
                map.clear();
                Terms inner;
                count_expr(inner, n.a);
                count.emplace(ic, std::move(inner));
            }
            const std::vector<std::pair<int32_t, uint64_t>> terms = count[ic].v;
            for (const auto &p : terms) t.add(*this, p.first, p.second);
            return;
        }
        t.add(*this, e, 1);
        switch (n.tag) {
        case T_X: case T_Y: case T_TAU: case T_E: case T_VAR: case T_NAT: case T_LET: break;
        case T_DECOR: count_expr(t, n.a); break;
        case T_APP: count_expr(t, n.a); count_expr(t, n.b); break;
        default:
            count_expr(t, n.a);
            if (is_binary(n.tag)) count_expr(t, n.b);
        }
    }

    std::unordered_map<int32_t, bool> compressed_memo;   // IsCompressedCache
    bool is_compressed(int32_t e) {                      // :124-155
        const int32_t c = cls(e);
        auto it = compressed_memo.find(c);
        if (it != compressed_memo.end()) return it->second;
        map.clear();                                     // Compressor::from_expr
        Terms t;
        count_expr(t, e);
        bool r = true;
        for (const auto &p : t.v)
            if (p.second > 1 && benefit(p.first, p.second, 3) != 0) { r = false; break; }
        compressed_memo.emplace(c, r);
        return r;
    }

    // ---- Expr::rewrite (src/lib.rs:560-598) --------------------------------------------------------------------------
    std::unordered_map<int32_t, uint64_t> var_of;        // class of a variable's formula -> its id (first definition wins)
    std::unordered_map<int32_t, int32_t> rw_memo;        // per call: class -> rewritten node (the function is pure given the map)
    int32_t rewrite(int32_t e) {
        const int32_t c = cls(e);
        auto vit = var_of.find(c);
        if (vit != var_of.end()) return mk(T_VAR, -1, -1, vit->second);
        const Node n = s.nodes[e];
        if (n.tag == T_ARC) {                            // the map decides, not the memo (it may change within a call)
            int32_t to;
            if (get_map(n.a, to)) return mk(T_ARC, to);
            const int32_t a = rewrite(n.a);
            if (equal(a, n.a)) return e;
            map[cls(n.a)] = a;                           // mem.get(a) interns by value: nothing to do here
            return mk(T_ARC, a);
        }
        auto mit = rw_memo.find(c);
        if (mit != rw_memo.end()) return mit->second;
        int32_t r;
        switch (n.tag) {
        case T_X: case T_Y: case T_TAU: case T_E: case T_NAT: case T_VAR: case T_LET: r = e; break;
        case T_DECOR: { Node d = n; d.a = rewrite(n.a); r = s.add(d); break; }
        case T_APP: { const int32_t a = rewrite(n.a), b = rewrite(n.b); r = mk(T_APP, a, b, 0, n.app); break; }
        default:
            if (is_binary(n.tag)) { const int32_t a = rewrite(n.a), b = rewrite(n.b); r = mk(n.tag, a, b); }
            else r = mk(n.tag, rewrite(n.a));
        }
        rw_memo.emplace(c, r);
        return r;
    }

    // ---- flatten (src/compressor.rs:167-211): Let / Var -> Arc of the definition, Decor dropped ----------------------
    std::unordered_map<uint64_t, int32_t> var_memo;      // (ctx, id) -> flattened definition (the reference recomputes it)
    int32_t flatten(int32_t e, int32_t ctx) {
        const Node n = s.nodes[e];
        switch (n.tag) {
        case T_X: case T_Y: case T_TAU: case T_E: case T_NAT: return e;
        case T_ARC: {
            int32_t to;
            if (get_map(n.a, to)) return mk(T_ARC, to);
            const int32_t a = flatten(n.a, ctx);
            if (equal(a, n.a)) return e;
            map[cls(n.a)] = a;
            return mk(T_ARC, a);
        }
        case T_DECOR: return flatten(n.a, ctx);
        case T_APP: { const int32_t a = flatten(n.a, ctx), b = flatten(n.b, ctx); return mk(T_APP, a, b, 0, n.app); }
        case T_LET: return flatten(n.a, n.ctx);
        case T_VAR: {
            if (ctx >= 0) {
                const uint64_t key = ((uint64_t)(uint32_t)ctx << 40) ^ n.u;
                auto it = var_memo.find(key);
                if (it != var_memo.end()) return mk(T_ARC, it->second);
                const Ctx c = s.ctxs[ctx];
                for (size_t i = 0; i < c.ids.size(); i++)
                    if (c.ids[i] == n.u) {
                        const int32_t a = flatten(c.defs[i], ctx);
                        var_memo.emplace(key, a);
                        return mk(T_ARC, a);
                    }
            }
            throw Error{MARAY_E_ARG, "compress: could not find variable $" + std::to_string(n.u) + " (the reference panics)"};
        }
        default:
            if (is_binary(n.tag)) { const int32_t a = flatten(n.a, ctx), b = flatten(n.b, ctx); return mk(n.tag, a, b); }
            return mk(n.tag, flatten(n.a, ctx));
        }
    }

    // ---- compressor::compress (:214-236) -----------------------------------------------------------------------------
    int32_t compress(int32_t root, uint32_t *n_vars_out) {
        int32_t res = root;
        Ctx ctx;
        map.clear();
        Terms terms;
        uint64_t var_len = 2;
        for (;;) {
            terms.clear();
            count_expr(terms, res);
            // last_max_benefit(2, var_len): the LAST of the terms with the greatest benefit (`>=`)
            bool have = false;
            size_t ind = 0;
            uint64_t best = 0;
            for (size_t i = 0; i < terms.v.size(); i++) {
                if (terms.v[i].second < 2) continue;
                if (!is_compressed(terms.v[i].first)) continue;
                const uint64_t b = benefit(terms.v[i].first, terms.v[i].second, var_len);
                if (b == 0) continue;
                if (!have || b >= best) { have = true; ind = i; best = b; }
            }
            if (!have) break;
            const uint64_t id = ctx.ids.size();
            const int32_t formula = terms.v[ind].first;
            var_len = 1 + digits(id);                    // "a{}".chars().count()
            ctx.ids.push_back(id);
            ctx.defs.push_back(formula);
            var_of.emplace(cls(formula), id);            // rewrite: `for var in &ctx.vars { if var.1 == self ...` -- first match
            rw_memo.clear();
            res = rewrite(res);
        }
        if (n_vars_out) *n_vars_out = (uint32_t)ctx.ids.size();
        if (ctx.ids.empty()) return res;
        s.ctxs.push_back(ctx);
        Node l; l.tag = T_LET; l.a = res; l.ctx = (int32_t)s.ctxs.size() - 1;
        return s.add(l);
    }
};

}   // namespace

// Expr::compress (src/lib.rs:610-614) on each channel, each with a MemoryManager of its own.
void scene_compress(Scene &s, uint32_t n_vars[3])
{
    for (int c = 0; c < 3; c++) {
        Comp C(s);
        C.map.clear();
        const int32_t flat = C.flatten(s.color[c], -1);
        uint32_t nv = 0;
        s.color[c] = C.compress(flat, &nv);
        if (n_vars) n_vars[c] = nv;
    }
    s.fixed = false;
}

// format!("{}", expr).chars().count() of a channel: what drives compress, exposed so that it can be pinned on its own.
uint64_t scene_display_len(Scene &s, int c)
{
    Comp C(s);
    return C.display_len(s.color[c]);
}

}   // namespace maray
