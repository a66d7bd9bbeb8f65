// expr.hpp — host-side Expr data model of libmaray_hip (product code).
//
// Mirrors the reference's input type of the render path: `Expr`
// (src/lib.rs:101-149), `Context` (src/lib.rs:51-55), `Token`
// (src/token.rs:10-38) and the file tuple `([u32;2], [Expr;3])`
// (src/lib.rs:1216-1235).  Nodes live in index-addressed arenas.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace maray {

enum Tag : uint8_t {   // current numbering, src/lib.rs:101-149
    T_ARC = 0, T_X, T_Y, T_TAU, T_E, T_VAR, T_NAT, T_NEG, T_ABS, T_RECIP, T_SQRT,
    T_STEP, T_SIN, T_EXP, T_LN, T_ADD, T_MUL, T_MAX, T_MIN, T_LET, T_DECOR, T_APP,
    T_COUNT
};

inline bool is_leaf(uint8_t t) { return t == T_X || t == T_Y || t == T_TAU || t == T_E; }
inline bool is_unary(uint8_t t) { return t == T_ARC || (t >= T_NEG && t <= T_LN); }
inline bool is_binary(uint8_t t) { return t >= T_ADD && t <= T_MIN; }

struct Node {
    uint8_t tag = 0;
    uint32_t app = 0;     // App function id
    uint64_t u = 0;       // Var id / Nat value
    int32_t a = -1;       // first child (unary, Arc, Decor inner, Let body, App lhs)
    int32_t b = -1;       // second child
    int32_t ctx = -1;     // Let: index into Scene::ctxs
    int32_t toks = -1;    // Decor: index into Scene::toklists
};

struct Ctx {              // Context { vars: Vec<(u64, Expr)> }
    std::vector<uint64_t> ids;
    std::vector<int32_t> defs;
};

struct Token {
    uint32_t kind = 0;    // 0 TokenExpr, 1 Str, 2..12 unit variants
    int32_t expr = -1;
    std::string str;
};

struct Scene {
    std::vector<Node> nodes;
    std::vector<Ctx> ctxs;
    std::vector<std::vector<Token>> toklists;
    int32_t color[3] = {-1, -1, -1};
    uint32_t w = 0, h = 0;
    bool legacy = false;
    bool fixed = false;   // fix_color applied

    int32_t add(const Node &n) { nodes.push_back(n); return (int32_t)nodes.size() - 1; }
};

// Error carrier used across the host code; converted to a code + message at the C boundary.
struct Error {
    int code;
    std::string msg;
};

// scene.cpp
void scene_decode(const uint8_t *buf, size_t len, Scene &out);            // throws Error
void scene_encode(const Scene &s, std::vector<uint8_t> &out);            // current numbering
uint64_t scene_node_count(const Scene &s, int c);
void scene_fix_color(Scene &s);                                          // var_fixer::fix_color
Scene scene_fixed(const Scene &s);                                       // the same into a new scene (the lowering's input stays as it is: no copy of an 88,000-node scene first)
void scene_rescale(Scene &s, uint32_t sx, uint32_t sy);
// simplify.cpp
void scene_simplify(Scene &s, uint32_t flags = 0);                       // Expr::simplify on each channel (flags: MARAY_SIMPLIFY_*)
// compress.cpp
void scene_compress(Scene &s, uint32_t n_vars[3]);                       // Expr::compress on each channel (n_vars: variables introduced; may be null)
uint64_t scene_display_len(Scene &s, int c);                             // format!("{}", color[c]).chars().count()

}   // namespace maray
