// scene.cpp — scene file reader/writer, fix_color and rescale (product code).
//
//   scene_decode     <- maray::open             src/lib.rs:1227-1235
//   scene_encode     <- maray::save             src/lib.rs:1216-1224
//   scene_fix_color  <- var_fixer::fix_color    src/var_fixer.rs:25-82
//   scene_rescale    <- Expr::scale semantics   src/lib.rs:804-806 (applied through Let as well)
#include "expr.hpp"

#include <cstring>
#include <unordered_map>

#include "maray_hip.h"

namespace maray {

// ---------------------------------------------------------------- reader ----
namespace {

struct Reader {
    const uint8_t *p;
    size_t len, pos = 0;
    bool legacy;
    Scene &s;
    int depth = 0;

    Reader(const uint8_t *p_, size_t len_, bool legacy_, Scene &s_) : p(p_), len(len_), legacy(legacy_), s(s_) {}

    [[noreturn]] void fail(const char *what) { throw Error{MARAY_E_DECODE, std::string("bincode: ") + what + " at offset " + std::to_string(pos)}; }
    uint32_t u32() {
        if (pos + 4 > len) fail("unexpected end of input");
        uint32_t v; memcpy(&v, p + pos, 4); pos += 4; return v;
    }
    uint64_t u64() {
        if (pos + 8 > len) fail("unexpected end of input");
        uint64_t v; memcpy(&v, p + pos, 8); pos += 8; return v;
    }

    int32_t expr() {
        if (++depth > 100000) fail("expression nested too deeply");
        uint32_t tag = u32();
        if (legacy) tag += 1;   // legacy numbering has no Arc variant
        if (tag >= T_COUNT) fail("invalid Expr variant");
        Node n;
        n.tag = (uint8_t)tag;
        switch (tag) {
        case T_X: case T_Y: case T_TAU: case T_E: break;
        case T_VAR: case T_NAT: n.u = u64(); break;
        case T_LET: {
            uint64_t k = u64();
            if (k > (len - pos) / 12) fail("Let variable count exceeds input");
            Ctx c;
            c.ids.reserve(k); c.defs.reserve(k);
            for (uint64_t i = 0; i < k; i++) {
                uint64_t id = u64();
                int32_t d = expr();
                c.ids.push_back(id); c.defs.push_back(d);
            }
            n.a = expr();
            s.ctxs.push_back(std::move(c));
            n.ctx = (int32_t)s.ctxs.size() - 1;
            break;
        }
        case T_DECOR: {
            n.a = expr();
            uint64_t k = u64();
            if (k > (len - pos) / 4) fail("token count exceeds input");
            std::vector<Token> toks;
            toks.reserve(k);
            for (uint64_t i = 0; i < k; i++) {
                Token t;
                t.kind = u32();
                if (t.kind == 0) t.expr = expr();
                else if (t.kind == 1) {
                    uint64_t sl = u64();
                    if (sl > len - pos) fail("string exceeds input");
                    t.str.assign((const char *)p + pos, (size_t)sl);
                    pos += (size_t)sl;
                } else if (t.kind > 12) fail("invalid Token variant");
                toks.push_back(std::move(t));
            }
            s.toklists.push_back(std::move(toks));
            n.toks = (int32_t)s.toklists.size() - 1;
            break;
        }
        case T_APP:
            n.app = u32(); n.a = expr(); n.b = expr(); break;
        default:
            n.a = expr();
            if (is_binary((uint8_t)tag)) n.b = expr();
        }
        depth--;
        return s.add(n);
    }

    void file() {
        s.w = u32(); s.h = u32();
        for (int c = 0; c < 3; c++) s.color[c] = expr();
        if (pos != len) fail("trailing bytes");
        s.legacy = legacy;
    }
};

}   // namespace

void scene_decode(const uint8_t *buf, size_t len, Scene &out)
{
    // Accept the numbering that consumes the buffer exactly: current first, then legacy.
    Error first{0, ""};
    for (int legacy = 0; legacy < 2; legacy++) {
        Scene s;
        try {
            Reader r(buf, len, legacy != 0, s);
            r.file();
            out = std::move(s);
            return;
        } catch (const Error &e) {
            if (!legacy) first = e;
            else throw Error{MARAY_E_DECODE, "current numbering: " + first.msg + "; legacy numbering: " + e.msg};
        }
    }
}

// ---------------------------------------------------------------- writer ----
namespace {

struct Writer {
    const Scene &s;
    std::vector<uint8_t> &out;
    void u32(uint32_t v) { uint8_t b[4]; memcpy(b, &v, 4); out.insert(out.end(), b, b + 4); }
    void u64(uint64_t v) { uint8_t b[8]; memcpy(b, &v, 8); out.insert(out.end(), b, b + 8); }
    void expr(int32_t i) {
        const Node &n = s.nodes[i];
        u32(n.tag);
        switch (n.tag) {
        case T_X: case T_Y: case T_TAU: case T_E: break;
        case T_VAR: case T_NAT: u64(n.u); break;
        case T_LET: {
            const Ctx &c = s.ctxs[n.ctx];
            u64(c.ids.size());
            for (size_t k = 0; k < c.ids.size(); k++) { u64(c.ids[k]); expr(c.defs[k]); }
            expr(n.a);
            break;
        }
        case T_DECOR: {
            expr(n.a);
            const auto &toks = s.toklists[n.toks];
            u64(toks.size());
            for (const Token &t : toks) {
                u32(t.kind);
                if (t.kind == 0) expr(t.expr);
                else if (t.kind == 1) { u64(t.str.size()); out.insert(out.end(), t.str.begin(), t.str.end()); }
            }
            break;
        }
        case T_APP: u32(n.app); expr(n.a); expr(n.b); break;
        default:
            expr(n.a);
            if (n.b >= 0) expr(n.b);
        }
    }
};

uint64_t count_rec(const Scene &s, int32_t i)
{
    const Node &n = s.nodes[i];
    uint64_t c = 1;
    if (n.tag == T_LET) for (int32_t d : s.ctxs[n.ctx].defs) c += count_rec(s, d);
    if (n.tag == T_DECOR) for (const Token &t : s.toklists[n.toks]) if (t.kind == 0) c += count_rec(s, t.expr);
    if (n.a >= 0) c += count_rec(s, n.a);
    if (n.b >= 0) c += count_rec(s, n.b);
    return c;
}

}   // namespace

void scene_encode(const Scene &s, std::vector<uint8_t> &out)
{
    Writer w{s, out};
    w.u32(s.w); w.u32(s.h);
    for (int c = 0; c < 3; c++) w.expr(s.color[c]);
}

uint64_t scene_node_count(const Scene &s, int c) { return count_rec(s, s.color[c]); }

// ------------------------------------------------------------- fix_color ----
// VarFixer (src/var_fixer.rs:8-70).  `ids: HashMap<Expr,u64>` is keyed by
// structural equality; here every fixed expression is hash-consed into the
// output scene, so structural equality is index equality.
namespace {

struct Interner {
    Scene &out;
    std::unordered_map<std::string, int32_t> map;             // Let / Decor nodes: the key spells out their context / tokens
    // plain nodes: open addressing over node ids + 1 (a slot's key is read from the node it names; four bytes a slot: the
    // table of a large scene stays in the cache) -- a tenth of a large scene's lowering was this table as a node-based map
    std::vector<int32_t> plain;
    size_t n_plain = 0;

    static void put(std::string &k, uint64_t v) { k.append((const char *)&v, 8); }
    static size_t hash_plain(const Node &n) {
        uint64_t h = 0x9e3779b97f4a7c15ull;
        for (uint64_t w : {(uint64_t)n.tag, n.u, (uint64_t)n.app, (uint64_t)(int64_t)n.a, (uint64_t)(int64_t)n.b}) { h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 32; }
        return (size_t)h;
    }
    static bool same_plain(const Node &x, const Node &y) { return x.tag == y.tag && x.u == y.u && x.app == y.app && x.a == y.a && x.b == y.b; }

    void reserve_plain(size_t nodes) {
        size_t want = 1024;
        while (want < 2 * nodes) want *= 2;
        if (want > plain.size()) rehash(want);
    }
    void rehash(size_t size) {
        std::vector<int32_t> old(size, 0);
        old.swap(plain);
        for (int32_t e : old) {
            if (!e) continue;
            size_t at = hash_plain(out.nodes[e - 1]) & (size - 1);
            while (plain[at]) at = (at + 1) & (size - 1);
            plain[at] = e;
        }
    }

    int32_t intern(const Node &n, const Ctx *ctx, const std::vector<Token> *toks) {
        if (!ctx && !toks) {
            if (2 * (n_plain + 1) > plain.size()) rehash(plain.empty() ? 1024 : 2 * plain.size());
            const size_t mask = plain.size() - 1;
            size_t at = hash_plain(n) & mask;
            for (; plain[at]; at = (at + 1) & mask)
                if (same_plain(out.nodes[plain[at] - 1], n)) return plain[at] - 1;
            const int32_t id = out.add(n);
            plain[at] = id + 1;
            n_plain++;
            return id;
        }
        std::string k;
        k.reserve(48);
        put(k, n.tag); put(k, n.u); put(k, n.app); put(k, (uint64_t)(int64_t)n.a); put(k, (uint64_t)(int64_t)n.b);
        if (ctx) {
            put(k, ctx->ids.size());
            for (size_t i = 0; i < ctx->ids.size(); i++) { put(k, ctx->ids[i]); put(k, (uint64_t)ctx->defs[i]); }
        }
        if (toks) {
            put(k, toks->size());
            for (const Token &t : *toks) { put(k, t.kind); put(k, (uint64_t)(int64_t)t.expr); put(k, t.str.size()); k += t.str; }
        }
        auto it = map.find(k);
        if (it != map.end()) return it->second;
        Node m = n;
        if (ctx) { out.ctxs.push_back(*ctx); m.ctx = (int32_t)out.ctxs.size() - 1; }
        if (toks) { out.toklists.push_back(*toks); m.toks = (int32_t)out.toklists.size() - 1; }
        int32_t id = out.add(m);
        map.emplace(std::move(k), id);
        return id;
    }
};

// ctx: &mut Vec<(u64,u64)> = (old, new); the reference scans it front to back (:32-34: the first entry with the old id
// wins), so a map that keeps the first entry per old id answers the same
struct Renames {
    std::unordered_map<uint64_t, uint64_t> first;
    void emplace_back(uint64_t old_id, uint64_t new_id) { first.emplace(old_id, new_id); }       // (emplace keeps an existing entry)
};

struct Fixer {
    const Scene &in;
    Interner I;
    std::unordered_map<int32_t, uint64_t> ids;   // VarFixer::ids (:10)
    uint64_t var_count = 0;                      // VarFixer::var_count (:12)

    Fixer(const Scene &in_, Scene &out) : in(in_), I{out, {}, {}} { I.reserve_plain(in_.nodes.size()); }

    // verbatim structural copy (Decor tokens are carried over unfixed, :67)
    int32_t copy(int32_t e) {
        const Node &n = in.nodes[e];
        Node m = n;
        m.ctx = m.toks = -1;
        if (n.a >= 0) m.a = copy(n.a);
        if (n.b >= 0) m.b = copy(n.b);
        if (n.tag == T_LET) {
            Ctx c;
            const Ctx &ic = in.ctxs[n.ctx];
            for (size_t i = 0; i < ic.ids.size(); i++) { c.ids.push_back(ic.ids[i]); c.defs.push_back(copy(ic.defs[i])); }
            return I.intern(m, &c, nullptr);
        }
        if (n.tag == T_DECOR) {
            std::vector<Token> toks = copy_tokens(n.toks);
            return I.intern(m, nullptr, &toks);
        }
        return I.intern(m, nullptr, nullptr);
    }
    std::vector<Token> copy_tokens(int32_t tl) {
        std::vector<Token> toks = in.toklists[tl];
        for (Token &t : toks) if (t.kind == 0) t.expr = copy(t.expr);
        return toks;
    }

    // VarFixer::fix (:25-70)
    int32_t fix(int32_t e, const Renames &ctx) {
        const Node &n = in.nodes[e];
        Node m;
        m.tag = n.tag; m.u = n.u; m.app = n.app;
        switch (n.tag) {
        case T_ARC: return fix(n.a, ctx);                                        // :29
        case T_X: case T_Y: case T_TAU: case T_E: case T_NAT: break;             // :30
        case T_VAR:                                                              // :31-36
            { auto it = ctx.first.find(n.u); if (it != ctx.first.end()) m.u = it->second; }
            break;
        case T_LET: {                                                            // :49-66
            const Ctx &ic = in.ctxs[n.ctx];
            Ctx c;
            Renames nc;
            for (size_t i = 0; i < ic.ids.size(); i++) {
                int32_t d = fix(ic.defs[i], ctx);          // definitions are fixed under the OUTER ctx (:52)
                uint64_t id;
                auto it = ids.find(d);
                if (it != ids.end()) id = it->second;      // :53-55
                else { id = var_count++; ids.emplace(d, id); }   // :56-61
                nc.emplace_back(ic.ids[i], id);
                c.ids.push_back(id); c.defs.push_back(d);
            }
            m.a = fix(n.a, nc);                            // body under the NEW ctx only (:65)
            return I.intern(m, &c, nullptr);
        }
        case T_DECOR: {                                                          // :67
            m.a = fix(n.a, ctx);
            std::vector<Token> toks = copy_tokens(n.toks);
            return I.intern(m, nullptr, &toks);
        }
        default:                                                                 // :37-48, :68
            m.a = fix(n.a, ctx);
            if (n.b >= 0) m.b = fix(n.b, ctx);
        }
        return I.intern(m, nullptr, nullptr);
    }
};

}   // namespace

Scene scene_fixed(const Scene &s)
{
    if (s.fixed) return s;
    Scene out;
    out.w = s.w; out.h = s.h; out.legacy = s.legacy;
    Fixer f(s, out);                      // one VarFixer for all three channels (:76)
    Renames empty;
    for (int c = 0; c < 3; c++) out.color[c] = f.fix(s.color[c], empty);
    out.fixed = true;
    return out;
}

void scene_fix_color(Scene &s)
{
    if (!s.fixed) s = scene_fixed(s);
}

// --------------------------------------------------------------- rescale ----
void scene_rescale(Scene &s, uint32_t sx, uint32_t sy)
{
    if (sx == 0 || sy == 0) throw Error{MARAY_E_ARG, "scale factors must be non-zero"};
    auto scaled = [&](uint8_t leaf, uint32_t k) {   // div(x(), nat(k)) = Mul(X, Recip(Nat k)), src/lib.rs:950-952
        Node l; l.tag = leaf;
        Node nat; nat.tag = T_NAT; nat.u = k;
        Node rc; rc.tag = T_RECIP; rc.a = s.add(nat);
        Node mu; mu.tag = T_MUL; mu.a = s.add(l); mu.b = s.add(rc);
        return s.add(mu);
    };
    size_t n0 = s.nodes.size();
    int32_t nx = scaled(T_X, sx), ny = scaled(T_Y, sy);
    auto remap = [&](int32_t &i) {
        if (i < 0 || (size_t)i >= n0) return;
        if (s.nodes[i].tag == T_X) i = nx;
        else if (s.nodes[i].tag == T_Y) i = ny;
    };
    for (size_t i = 0; i < n0; i++) { remap(s.nodes[i].a); remap(s.nodes[i].b); }
    for (Ctx &c : s.ctxs) for (int32_t &d : c.defs) remap(d);
    for (auto &tl : s.toklists) for (Token &t : tl) if (t.kind == 0) remap(t.expr);
    for (int c = 0; c < 3; c++) remap(s.color[c]);
    uint64_t w = (uint64_t)s.w * sx, h = (uint64_t)s.h * sy;
    if (w > 0xFFFFFFFFull || h > 0xFFFFFFFFull) throw Error{MARAY_E_ARG, "rescaled size overflows u32"};
    s.w = (uint32_t)w; s.h = (uint32_t)h;
}

}   // namespace maray
