// jit_emit.hpp — the emitter of the specialised kernels: tape ops -> statements of HIP source (product code; included by
// jit_source.cpp alone).  Values are typed as they are emitted (f64, lane mask, negated mask), SKIP regions become scalar
// branches, guarded OR / max trees become reductions over the set bits of the guard words (RedPlan).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "jit_parts.hpp"
#include "maray_hip.h"

namespace maray {

namespace {

bool jit_row_guards_enabled()
{
    const char *e_ = getenv("MARAY_JIT_ROW_GUARDS");      // "0": compile the row-level SKIP ops away (ablation)
    return !(e_ && e_[0] == '0');
}

// f64 values a VALU instruction encodes as an inline constant (gfx9: 0, +-0.5, +-1, +-2, +-4, 1/(2 pi))
bool inline_f64(uint64_t bits)
{
    const uint64_t mag = bits & 0x7fffffffffffffffull;
    return mag == 0 || mag == 0x3fe0000000000000ull || mag == 0x3ff0000000000000ull || mag == 0x4000000000000000ull ||
           mag == 0x4010000000000000ull || bits == 0x3fc45f306dc9c882ull;
}

std::string lit(double v)
{
    if (v != v) return "__builtin_nan(\"\")";
    if (std::isinf(v)) return v > 0 ? "__builtin_inf()" : "(-__builtin_inf())";
    char buf[64];
    snprintf(buf, sizeof buf, v < 0 || std::signbit(v) ? "(%a)" : "%a", v);
    return buf;
}

// Values whose every possible result is exactly +0.0 or 1.0 ("booleans": Step, Step(Sin), and
// Mul / Min / Max / `1 + Neg(.)` of booleans) are carried as 64-bit wave lane masks (`mr_mask`, one
// bit per lane, the result of a ballot): wave-uniform integers that live in SGPR pairs and are
// combined by s_and_b64 / s_or_b64 / s_not_b64 instead of v_mul_f64 / v_min_f64 / v_max_f64 +
// v_cndmask, and tested by s_cmp_*_u64.  Exact, because on {+0.0, 1.0}: a*b = min(a,b) = a AND b,
// max(a,b) = a OR b, 1 + (-a) = NOT a (1 + -1 = +0, 1 + -0 = 1) — all results are again +0.0 or
// 1.0, never -0.0.  About half of chess.maray's ops are of this kind.  The f64 value is
// materialised (mr_pos(m), one v_cndmask_b32 on the mask) only where a non-boolean op
// consumes it.  Explicit masks rather than C++ `bool`s: LLVM keeps an i1 that crosses a basic
// block (every region result does) as a 0/1 VGPR and re-derives the mask with v_cmp, three VALU
// ops per region on the path that skips it.  Bits of lanes that are not executing are garbage
// (NOT sets them); they can only make a region run that could have been skipped.
// Where each guard (a y value that only gates SKIP ops) lives in a tile's guard words.  A guard whose value is the OR of
// other guards -- the bound of a group of shapes is max(bound of shape, bound of shape, ...), and hash-consing makes
// those operands the very values that are OUT as the shapes' own guards -- gets no bit: its test in the PIXEL section
// is "any of its members' bits", one s_and on the word.  Computed in the ROW kernel such a guard costs its members'
// cones all over again (chess: 40 group guards = 54 % of the ROW kernel's work, and the dearest jobs: 2,300 ops in a
// chain where a shape's guard has 100-200).

struct GuardPlan {
    uint32_t n_pos = 0;                          // bits in use
    std::vector<int32_t> pos;                    // per guard: its bit, or -1: derived
    std::vector<std::vector<uint32_t>> members;  // per derived guard: the bits it is the OR of
};

// ---- guarded OR-reductions of the PIXEL section -------------------------------------------------------------------
// A scene that paints shape over shape is `max(shape, max(shape, ...))` of booleans: after the lowering a balanced OR
// tree whose leaves are conjunctions, each inside a SKIPZ region that a rectangle guard (one bit of the guard words)
// switches off, with group guards and "all lanes already covered" SKIPNZ regions around the sub-trees.  Walked as
// written, a pass of 64 pixels tests that whole skeleton -- chess: ~85 scalar instructions of bit tests, branches and
// mask moves for the 13 top-level groups alone, whatever is set -- to enter the 2-4 shapes whose bit is set; the busy
// tiles are bound by exactly that unit.  So the tree is recognised and evaluated from the other end: the OR of the
// leaves WHOSE BIT IS SET, found with s_ff1 on the masked guard words and reached through a branch table -- cost
// proportional to the set bits, not to the tree.  Legal because OR on {+0.0, 1.0} (masks) is associative and commutative,
// a leaf whose bit is clear is +0.0 over the whole rectangle (that is what its guard says), and an evaluator may ignore
// any SKIP op: the group guards and the SKIPNZ regions of the tree are not consulted at all (the loop leaves as soon as
// every lane is covered, which is what the SKIPNZ regions were for).  The same for a max tree of f64 values -- every shape
// with a colour of its own, channel = max_i(shape_i * c_i): NaN-ignoring max with -0 < +0 is associative and commutative
// bit for bit, a leaf whose bit is clear is +0.0, so the value is the max of the leaves whose bit is set, of the free
// leaves, and of +0.0 if any bit is clear (the accumulator starts as +0.0 then, else as NaN, max's identity).
struct RedPlan {
    enum Role : uint8_t { NONE = 0, LEAF_SKIP, LEAF_END, INNER, ROOT, IGNORED_SKIP };
    struct Red {
        uint32_t root = 0;
        std::vector<uint32_t> leaf_skip, leaf_end, leaf_bit;        // guarded leaves: their SKIPZ op, their last op, their guard bit
        // per leaf: the y values that are boolean FACTORS of it (the leaf is an AND tree and they are among its operands: a
        // shape's horizontal edge).  0 on this row => the leaf is 0 on this row, whatever its rectangle's guard bit says: one
        // scalar test ahead of the leaf's body
        std::vector<std::vector<uint32_t>> leaf_yfactors;
        bool boolean = true;                                         // an OR of lane masks; else a max of f64 values
    };
    std::vector<uint8_t> role;          // per op
    std::vector<int32_t> red;           // per op with a role: its reduction
    std::vector<int32_t> leaf;          // LEAF_SKIP / LEAF_END: index into Red::leaf_*
    std::vector<Red> reds;
    bool empty() const { return reds.empty(); }
};

// is_bool: per op, the emitter's own typing (a dry run).  guard g of a SKIPZ on y value guard_first + g has bit gp.pos[g]
// (or is derived: no bit).  A tree qualifies when it has at least `min_leaves` guarded leaves.
RedPlan plan_reductions(const uint64_t *ops, uint32_t n, uint32_t n_slots, const std::vector<uint8_t> &is_bool, uint32_t guard_first,
                        const GuardPlan &gp, const std::vector<uint8_t> &ybool, uint32_t min_leaves = 4)
{
    RedPlan rp;
    rp.role.assign(n, RedPlan::NONE); rp.red.assign(n, -1); rp.leaf.assign(n, -1);
    // producers of every op's operands (ops write slots and ACC), use counts, and the regions that end at an op
    std::vector<int32_t> pa(n, -1), pb(n, -1), slot(n_slots, -1);
    std::vector<uint32_t> uses(n, 0);                       // reads by computing ops and OUTs
    std::vector<std::vector<uint32_t>> skips_ending(n), skips_on(n);       // per op: the SKIP ops that end at it / that it guards
    int32_t acc = -1;
    auto prod = [&](uint32_t r) -> int32_t {
        if (MARAY_REF_KIND(r) == MARAY_K_SLOT) return slot[MARAY_REF_INDEX(r)];
        if (MARAY_REF_KIND(r) == MARAY_K_SPEC && MARAY_REF_INDEX(r) == MARAY_SPEC_ACC) return acc;
        return -1;
    };
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t ins = ops[i];
        const uint32_t op = MARAY_INS_OP(ins), dst = MARAY_INS_DST(ins);
        if (op == MARAY_OP_NOP) continue;
        if (op != MARAY_OP_TEXDIM) pa[i] = prod(MARAY_INS_A(ins));
        if (op == MARAY_OP_SKIPZ || op == MARAY_OP_SKIPNZ) {
            skips_ending[i + MARAY_INS_AUX(ins)].push_back(i);
            if (pa[i] >= 0) skips_on[pa[i]].push_back(i);
            continue;
        }
        if (pa[i] >= 0) uses[pa[i]]++;
        if (op == MARAY_OP_OUT) continue;
        if (op >= MARAY_OP_ADD && op <= MARAY_OP_APP) { pb[i] = prod(MARAY_INS_B(ins)); if (pb[i] >= 0) uses[pb[i]]++; }
        acc = (int32_t)i;
        if (dst != MARAY_DST_NONE) slot[dst] = (int32_t)i;
    }
    // the rectangle-guarded region that ends at op e (outermost SKIPZ on a guard with a bit of its own), or -1
    auto guarded_region = [&](uint32_t e, uint32_t *bit) -> int32_t {
        for (uint32_t s : skips_ending[e]) {         // ascending: outermost first
            const uint32_t g = MARAY_INS_A(ops[s]);
            if (MARAY_INS_OP(ops[s]) != MARAY_OP_SKIPZ || MARAY_REF_KIND(g) != MARAY_K_YVAL || MARAY_REF_INDEX(g) < guard_first) continue;
            const uint32_t k = MARAY_REF_INDEX(g) - guard_first;
            if (k >= gp.pos.size() || gp.pos[k] < 0) continue;
            *bit = (uint32_t)gp.pos[k];
            return (int32_t)s;
        }
        return -1;
    };
    std::vector<uint8_t> inside_leaf(n, 0);
    for (int32_t R = (int32_t)n - 1; R >= 0; R--) {         // outermost trees first
        if (MARAY_INS_OP(ops[R]) != MARAY_OP_MAX || rp.role[R] != RedPlan::NONE || inside_leaf[R]) continue;
        std::vector<uint32_t> inner, st{(uint32_t)R};
        RedPlan::Red red;
        red.root = (uint32_t)R;
        red.boolean = is_bool[R] != 0;
        bool ok = true;
        while (!st.empty() && ok) {
            const uint32_t v = st.back(); st.pop_back();
            inner.push_back(v);
            for (int32_t p : {pa[v], pb[v]}) {
                if (p < 0) continue;                                               // a literal or a y value: a free leaf
                if (MARAY_INS_OP(ops[p]) == MARAY_OP_MAX && (is_bool[p] != 0) == red.boolean && uses[p] == 1 && rp.role[p] == RedPlan::NONE) { st.push_back((uint32_t)p); continue; }
                uint32_t bit = 0;
                const int32_t s = ((is_bool[p] || !red.boolean) && uses[p] == 1) ? guarded_region((uint32_t)p, &bit) : -1;
                if (s < 0) continue;                                               // a free leaf: evaluated where it stands
                for (uint32_t j = (uint32_t)s; j <= (uint32_t)p && ok; j++) ok = rp.role[j] == RedPlan::NONE && !inside_leaf[j];
                red.leaf_skip.push_back((uint32_t)s); red.leaf_end.push_back((uint32_t)p); red.leaf_bit.push_back(bit);
            }
        }
        // a part of the tree may guard a SKIP op only if that op goes with the tree (it ends at an OR of the tree: the
        // "every lane is covered already" regions); any other reader of its value needs the value
        {
            std::vector<uint32_t> parts = inner;
            parts.insert(parts.end(), red.leaf_end.begin(), red.leaf_end.end());
            for (uint32_t v : parts) {
                if (v == (uint32_t)R) continue;
                for (uint32_t sk : skips_on[v]) {
                    const uint32_t end = sk + MARAY_INS_AUX(ops[sk]);
                    ok = ok && std::find(inner.begin(), inner.end(), end) != inner.end();
                }
            }
        }
        if (!ok || red.leaf_end.size() < min_leaves) continue;
        // two leaves on one bit (a shape two sub-trees share) cannot be told apart by the dispatch: leave such a tree alone
        { std::vector<uint32_t> b = red.leaf_bit; std::sort(b.begin(), b.end()); if (std::adjacent_find(b.begin(), b.end()) != b.end()) continue; }
        const int32_t id = (int32_t)rp.reds.size();
        for (uint32_t v : inner) {
            rp.role[v] = v == (uint32_t)R ? RedPlan::ROOT : RedPlan::INNER; rp.red[v] = id;
            for (uint32_t s : skips_ending[v]) { rp.role[s] = RedPlan::IGNORED_SKIP; rp.red[s] = id; }
        }
        red.leaf_yfactors.resize(red.leaf_end.size());
        for (size_t k = 0; k < red.leaf_end.size(); k++) {
            const uint32_t s = red.leaf_skip[k], e = red.leaf_end[k];
            if (red.boolean) {          // the AND tree under the leaf's last op, through ANDs with one reader
                std::vector<uint32_t> andst{e};
                while (!andst.empty()) {
                    const uint32_t v = andst.back(); andst.pop_back();
                    const uint32_t op = MARAY_INS_OP(ops[v]);
                    if (!(op == MARAY_OP_MUL || op == MARAY_OP_MIN) || !is_bool[v]) continue;
                    const uint32_t refs[2] = {MARAY_INS_A(ops[v]), MARAY_INS_B(ops[v])};
                    const int32_t prods[2] = {pa[v], pb[v]};
                    for (int q = 0; q < 2; q++) {
                        if (MARAY_REF_KIND(refs[q]) == MARAY_K_YVAL && MARAY_REF_INDEX(refs[q]) < ybool.size() && ybool[MARAY_REF_INDEX(refs[q])])
                            red.leaf_yfactors[k].push_back(MARAY_REF_INDEX(refs[q]));
                        else if (prods[q] >= (int32_t)s && uses[prods[q]] == 1) andst.push_back((uint32_t)prods[q]);       // (a factor may guard a wave-level region as well)
                    }
                }
            }
            for (uint32_t j = s; j <= e; j++) inside_leaf[j] = 1;
            // regions that end at the leaf's last op and start before its guard's SKIPZ would enclose it: ignored as well
            for (uint32_t q : skips_ending[e]) if (q < s) { rp.role[q] = RedPlan::IGNORED_SKIP; rp.red[q] = id; }
            rp.role[s] = RedPlan::LEAF_SKIP; rp.role[e] = RedPlan::LEAF_END;
            rp.red[s] = rp.red[e] = id; rp.leaf[s] = rp.leaf[e] = (int32_t)k;
        }
        rp.reds.push_back(std::move(red));
    }
    return rp;
}

// f64::max / f64::min on constants, as lower.cpp folds them and v_max_f64 / v_min_f64 compute them (NaN-ignoring, -0 < +0)
inline double fold_max(double a, double b) { if (a != a) return b; if (b != b) return a; if (a == b) return std::signbit(a) ? b : a; return a > b ? a : b; }
inline double fold_min(double a, double b) { if (a != a) return b; if (b != b) return a; if (a == b) return std::signbit(a) ? a : b; return a < b ? a : b; }

struct Emitter {
    enum Kind { DBL, BOOL, NEGBOOL, REDPART };   // NEGBOOL: value is -(b) in {-0.0, -1.0}, b = the named mask; REDPART: part of a guarded OR-reduction (RedPlan): no value of its own
    struct Val {
        Kind kind = DBL;
        std::string d;   // name / literal of the double, empty until materialised
        std::string b;   // name / literal of the lane mask (BOOL, NEGBOOL)
        uint32_t d_scope = 0;   // the region (C++ block) the materialised double was declared in; 0 = the kernel's own block
        bool wide = false;      // two rows per lane (Emitter::pair): the value differs between the rows (a pair, mr_p / mr_pm); else one value serves both
        // value = x + k (cmp_kind 1) or -(x + k) (2) with k a finite constant: Step of it is the compare x >= -k (x <= -k), without
        // the addition (jit_emit: case MARAY_OP_STEP)
        int cmp_kind = 0; std::string cmp_x; double cmp_k = 0.0; bool cmp_wide = false;
        bool cst = false;       // a known constant (a literal of the tape, MR_NONE / MR_ALL as numbers, or arithmetic on such): cval
        double cval = 0.0;
    };
    const maray_program &P;
    std::string out;
    std::vector<Val> vals;   // one per op of the current section
    std::string yv_name = "yv";
    bool ignore_row_guards = false;
    uint32_t guard_first = 0, guard_words = 0;   // y values >= guard_first are SKIP guards; the kernel packs them as bits, 64 per word
    std::vector<uint8_t> is_bool_op;             // out: per op of the last section(), was its value carried as a bool
    std::vector<uint8_t> bool_hint;              // in: the same from a dry run without row guards (types their regions)
    // Constants that no VALU instruction can encode inline are read from a table in constant memory, laid out in the
    // order the code reads them (one entry per use, shared inside a basic block): the compiler then fetches a block's
    // constants with a few s_load_dwordx4/x8/x16 instead of two s_mov_b32 per use, and the scalar unit -- which also
    // does all the boolean algebra and every region's branch -- is what bounds this kernel.
    bool assume_guards_zero = false;             // PIXEL: emit the variant for a tile none of whose guard bits is set
    bool out_guard_bits = false;                 // ROW-section OUT of a guard (index >= guard_first): OR its bit into `gacc`
    bool ktab = false;
    int sin_k = -1;                    // >= 0: the bounded Step(Sin) reads its reduction constants from mr_kc[sin_k .. sin_k + 5]
    uint32_t min_region = 0;                    // PIXEL: wave-level SKIP ops over fewer ops than this are ignored
    uint32_t min_region_row = 0;                // ROW: the same (a wavefront's lanes are 64 rows, or the 64 rectangles of a band of rows)
    // y values that are booleans (exactly +0.0 or 1.0 on every row: a Step of y-only arguments and what AND / OR / NOT make
    // of such): the PIXEL section reads them as lane masks (all lanes or none) from a scalar compare, so that min / max /
    // mul with them stay mask algebra.  Left as numbers they turn every shape they clip -- and then the whole OR tree of
    // shapes above -- into f64 code: three v_max, a v_cndmask and a v_cmp where one s_or_b64 does.
    std::vector<uint8_t> ybool;                 // PIXEL in: per y value; ROW out (dry run): what each OUT wrote
    std::vector<uint8_t> out_is_bool;
    const GuardPlan *plan = nullptr;            // guard -> bit(s) (identity when null)
    uint32_t gw_inline_max = 12;                // > 12 guard words: lane i of mr_gt<j> holds word 64 j + i of the tile at hand
    std::string gw_lane_base;                   // narrow rectangles: lane (this expression) + i of mr_gt0 holds word i of the pass's rectangle
    int stage_first = -1;                       // ROW: >= 0: OUT k goes to LDS, ys[(k - stage_first) * 68 + lane] (jit_source_rows)
    std::string td = "double", tm = "mr_mask";  // types of a value / a boolean in the generated text ("mr_d" / "mr_m": four pixels per lane)
    std::vector<double> ktab_vals;
    std::unordered_map<uint64_t, uint32_t> ktab_block;
    // Two rows per lane (device_math.h, mr_p / mr_pm): the NARROW passes of a wavefront that owns the same 64 pixels of two
    // neighbouring rows.  y values and Y are pairs (yv / yw: row r, yv1 / yw1: row r + 1); an op is a pair when an operand is.
    bool pair = false;
    std::vector<uint8_t> is_wide_op;             // out: per op of the last section(), was its value a pair
    std::vector<uint8_t> wide_hint;              // in: the same from a dry run (types the regions' variables, which are declared ahead of their last op)
    bool fuse_cmp = true;                       // Step(x + k) as one compare (MARAY_JIT_FUSE_CMP=0: ablation)
    bool texel_once = false;                    // PIXEL: App ops of one image on the same coordinates share one mr_texel (descriptors mr_t<image> in scope)
    std::map<std::string, std::pair<std::string, uint32_t>> texels;      // (image, x, y, width) -> its variable and the block it was declared in
    const RedPlan *rplan = nullptr;             // PIXEL: the guarded OR-reductions of the section being emitted (null: walk the tree as written)
    uint32_t red_serial = 0;                    // (names of the labels: a section may be emitted more than once into one kernel)
    explicit Emitter(const maray_program &p) : P(p) {}

    // word wi of the guard words of the rectangle at hand: an SGPR pair by name (<= 12 words); beyond, lane wi % 64 of a
    // per-lane value (one v_readlane pair)
    // many guard words (one per lane): is word wi of the rectangle at hand non-zero at all?  One ballot per pass answers for
    // every word (mr_gnzp / mr_gnz<j>, jit_source); "" when the words sit in SGPRs and the test is the loop's own
    std::string guard_word_nonzero(uint32_t wi) const {
        if (guard_words <= gw_inline_max) return "";
        if (!gw_lane_base.empty()) return "((mr_gnzp >> " + std::to_string(wi) + "u) & 1ull) != 0ull";
        return "((mr_gnz" + std::to_string(wi / 64) + " >> " + std::to_string(wi % 64) + "u) & 1ull) != 0ull";
    }
    std::string guard_word(uint32_t wi) const {
        if (guard_words <= gw_inline_max) return "gq" + std::to_string(wi);
        if (!gw_lane_base.empty()) return "mr_lane64(mr_gt0, " + gw_lane_base + " + " + std::to_string(wi) + "u)";
        return "mr_lane64(mr_gt" + std::to_string(wi / 64) + ", " + std::to_string(wi % 64) + "u)";
    }

    void section(const uint64_t *ops, uint32_t n, uint32_t n_slots, bool pixel, const char *prefix)
    {
        vals.assign(n, Val());
        texels.clear();
        is_bool_op.assign(n, 0);
        is_wide_op.assign(n, 0);
        const std::string tdw = pair ? "mr_p" : td, tmw = pair ? "mr_pm" : tm;      // types of a pair
        std::vector<int> slot(n_slots, -1);
        int acc = -1;
        char name[48];
        Val tmp_const[2];

        // operand -> Val* (constants get a temporary)
        auto ref = [&](uint32_t r, int which) -> Val * {
            const uint32_t kind = MARAY_REF_KIND(r), idx = MARAY_REF_INDEX(r);
            Val &t = tmp_const[which];
            t = Val();
            switch (kind) {
            case MARAY_K_SLOT: return &vals[slot[idx]];
            case MARAY_K_CONST: {
                const double c = P.consts[idx];
                t.d = lit(c);
                t.cst = true; t.cval = c;
                uint64_t bits; memcpy(&bits, &c, 8);
                if (ktab && !inline_f64(bits)) {
                    auto it = ktab_block.find(bits);
                    if (it == ktab_block.end()) { it = ktab_block.emplace(bits, (uint32_t)ktab_vals.size()).first; ktab_vals.push_back(c); }
                    t.d = "mr_kc[" + std::to_string(it->second) + "]";
                }
                if (bits == 0x3ff0000000000000ull) { t.kind = BOOL; t.b = "MR_ALL"; }
                else if (bits == 0) { t.kind = BOOL; t.b = "MR_NONE"; }
                return &t;
            }
            case MARAY_K_YVAL:
                t.d = yv_name + "[" + std::to_string(idx) + "]";
                if (pair) { t.d = "mr_p(" + t.d + ", " + yv_name + "1[" + std::to_string(idx) + "])"; t.wide = true; }
                if (idx < ybool.size() && ybool[idx]) {
                    t.kind = BOOL; t.b = "mr_ym(yw, " + std::to_string(idx) + "u)";
                    if (pair) t.b = "mr_pm(" + t.b + ", mr_ym(yw1, " + std::to_string(idx) + "u))";
                }
                return &t;
            default:
                if (idx == MARAY_SPEC_ACC) return &vals[acc];
                static const char *const spec_name[] = {"X", "Y", "", "XMAX", "XMIN", "YMAX", "YMIN"};
                t.d = spec_name[idx];
                t.wide = pair && idx == MARAY_SPEC_Y;
                return &t;
            }
        };
        // the double form of a value, materialising it once if needed
        // leaf: the block of a reduction's leaf (no variable, no else).  mask: the lanes on which what the region computes can
        // matter -- a wave-level region of booleans computes q of n = p AND q (p OR q): where p is 0 (1), n does not depend on q
        struct Open { uint32_t end; bool as_bool; bool nz; uint32_t id; bool leaf; std::string mask; bool mask_wide; };
        std::vector<Open> open;     // SKIPZ / SKIPNZ regions being emitted, innermost last
        uint32_t next_scope = 1;
        auto dbl = [&](Val *v, const char *hint, uint32_t i, int which) -> std::string {
            // a mask defined outside a region may have been materialised inside one: that variable is out of scope now
            bool in_scope = v->d_scope == 0;
            for (const Open &o : open) in_scope |= o.id == v->d_scope;
            if (!v->d.empty() && in_scope) return v->d;
            if (v->b == "MR_NONE") return v->d = v->kind == BOOL ? "0.0" : "(-0.0)";
            if (v->b == "MR_ALL") return v->d = v->kind == BOOL ? "1.0" : "(-1.0)";
            snprintf(name, sizeof name, "%s%u_%c", hint, i, which ? 'b' : 'a');
            out += "    const " + (v->wide ? tdw : td) + " ";
            out += name;
            out += v->kind == BOOL ? " = mr_pos(" + v->b + ");\n" : " = mr_neg01(" + v->b + ");\n";
            v->d = name;
            v->d_scope = open.empty() ? 0 : open.back().id;
            return v->d;
        };

        std::vector<uint8_t> forced(n, 0);      // op ends a region known at compile time to be skipped: its value is 0 (1) or 1 (2)
        // guarded OR-reductions (RedPlan): a leaf's text is collected aside and placed behind the branch table at the root
        const RedPlan *rp = (pixel && rplan && !rplan->empty() && !assume_guards_zero && !ignore_row_guards) ? rplan : nullptr;
        std::vector<std::vector<std::string>> red_leaf_text(rp ? rp->reds.size() : 0);
        std::vector<std::vector<std::string>> red_free(rp ? rp->reds.size() : 0);
        if (rp) for (size_t k = 0; k < rp->reds.size(); k++) red_leaf_text[k].resize(rp->reds[k].leaf_end.size());
        std::string out_saved;                  // the section's text while a leaf's is being collected in `out`
        const uint32_t serial = red_serial++;
        for (uint32_t i = 0; i < n; i++) {
            const uint64_t ins = ops[i];
            const uint32_t op = MARAY_INS_OP(ins), aux = MARAY_INS_AUX(ins), dst = MARAY_INS_DST(ins);
            if (op == MARAY_OP_NOP) continue;
            const uint8_t role = rp ? rp->role[i] : (uint8_t)RedPlan::NONE;
            if (role == RedPlan::IGNORED_SKIP) continue;            // legal: an evaluator may ignore any SKIP op
            if (role == RedPlan::LEAF_SKIP) {
                out_saved.swap(out);                                // (out_saved was empty: leaves do not nest)
                open.push_back(Open{i + aux, true, false, next_scope++, true, std::string(), false});
                ktab_block.clear();
                continue;
            }
            Val *va = (op != MARAY_OP_TEXDIM && !forced[i]) ? ref(MARAY_INS_A(ins), 0) : nullptr;
            if (op == MARAY_OP_SKIPZ || op == MARAY_OP_SKIPNZ) {
                // if (some lane still needs it) { region } else result = 0 / 1;  -- a scalar branch on the ballot
                const bool nz = op == MARAY_OP_SKIPNZ;
                const uint32_t end = i + aux;
                snprintf(name, sizeof name, "%s%u", prefix, end);
                const uint32_t gref = MARAY_INS_A(ins);
                const bool row_guard = pixel && MARAY_REF_KIND(gref) == MARAY_K_YVAL;
                if (row_guard && ignore_row_guards) continue;      // legal: an evaluator may ignore any SKIP op
                const uint32_t min_region = pixel ? this->min_region : min_region_row;
                if (!row_guard && min_region) {                            // a wave-level region too cheap to pay for its test and branch
                    // what the region's ops cost the vector unit, roughly in instructions: a gather, a libm body or a division
                    // is not "an op" (a 20-op region around a texture lookup is worth its branch)
                    uint32_t cost = 0;
                    for (uint32_t j = i + 1; j <= end && cost < min_region; j++)
                        switch (MARAY_INS_OP(ops[j])) {
                        case MARAY_OP_NOP: case MARAY_OP_SKIPZ: case MARAY_OP_SKIPNZ: break;
                        case MARAY_OP_RECIP: case MARAY_OP_SQRT: cost += 12; break;
                        case MARAY_OP_SIN: case MARAY_OP_EXP: case MARAY_OP_LN: case MARAY_OP_STEPSIN: case MARAY_OP_APP: cost += 30; break;
                        default: cost += 1;
                        }
                    if (cost < min_region) continue;
                }
                // the region's variable is a lane mask when its last op yields one: the guard tells for a wave-level
                // region (a boolean guards a boolean AND / OR), the dry run for a row-level one (its guard is a y value)
                const bool as_bool = row_guard ? (end < bool_hint.size() && bool_hint[end]) : va->kind == BOOL;
                std::string cond;
                if (row_guard && !nz && guard_words && MARAY_REF_INDEX(gref) >= guard_first && assume_guards_zero) {
                    forced[end] = 1;            // nothing of the region is emitted; op `end` becomes the constant
                    i = end - 1;
                    continue;
                }
                if (!row_guard && va->kind == BOOL && (va->b == "MR_NONE" || va->b == "MR_ALL")) {
                    // the guard is a literal (it followed from regions skipped above): decide here
                    if ((va->b == "MR_NONE") != nz) { forced[end] = nz ? 2 : 1; i = end - 1; }     // taken: the region is never emitted
                    continue;                                                                    // not taken: an evaluator may ignore a SKIP op
                }
                if (row_guard && !nz && guard_words && MARAY_REF_INDEX(gref) >= guard_first) {
                    // a row bound: one bit of a guard word that sits in an SGPR since the kernel's prologue
                    const uint32_t g = MARAY_REF_INDEX(gref) - guard_first;
                    auto word = [&](uint32_t wi) { return guard_word(wi); };
                    // tests on 32-bit halves of the words: s_and_b32 sets SCC and the branch follows (a 64-bit test is
                    // s_and + s_cmp_u64 + the branch, on the unit that bounds the busy tiles)
                    std::vector<uint64_t> m(guard_words, 0);
                    if (!plan || plan->pos[g] >= 0) {
                        const uint32_t k = plan ? (uint32_t)plan->pos[g] : g;
                        m[k / 64] = 1ull << (k % 64);
                    } else for (uint32_t k : plan->members[g]) m[k / 64] |= 1ull << (k % 64);       // derived: any of its members' bits
                    std::string any;
                    int terms = 0;
                    for (uint32_t wi = 0; wi < guard_words; wi++)
                        for (int half = 0; half < 2; half++) {
                            const uint32_t bits = (uint32_t)(m[wi] >> (32 * half));
                            if (!bits) continue;
                            char hex[24];
                            snprintf(hex, sizeof hex, "0x%xu", bits);
                            any += std::string(terms++ ? " | " : "") + "((unsigned)" + (half ? "(" + word(wi) + " >> 32)" : word(wi)) + " & " + hex + ")";
                        }
                    cond = "(" + (any.empty() ? std::string("0u") : any) + ") != 0u";
                } else if (row_guard) {
                    // a y value is uniform over the block: test its bits on the scalar unit, no ballot, no VALU (two rows per
                    // lane: the region is entered when either row needs it)
                    const std::string k = std::to_string(MARAY_REF_INDEX(gref));
                    auto test = [&](const std::string &yw_) {
                        return nz ? "!(" + yw_ + "[2 * " + k + " + 1] == 0x3ff00000u && " + yw_ + "[2 * " + k + "] == 0u)"
                                  : "((" + yw_ + "[2 * " + k + " + 1] << 1) | " + yw_ + "[2 * " + k + "]) != 0u";
                    };
                    cond = pair ? "(" + test("yw") + ") || (" + test("yw1") + ")" : test("yw");
                } else if (as_bool) cond = nz ? "mr_any(~" + va->b + ")" : "mr_any(" + va->b + ")";      // a scalar compare
                else cond = (nz ? "mr_any(mr_ne1(" : "mr_any(mr_ne0(") + dbl(va, "m", i, 0) + "))";
                // several regions may end at one op (a row-level guard around a wave-level one): one variable
                bool typed_bool = as_bool;
                bool declared = false;
                for (const Open &o : open) if (o.end == end && !o.leaf) { declared = true; typed_bool = o.as_bool; }
                const bool wide_var = pair && (end >= wide_hint.size() || wide_hint[end]);      // (no hint: the typing run, whose text is not used)
                if (!declared) out += typed_bool ? "    " + (wide_var ? tmw : tm) + " b" + std::string(name) + ";\n" : "    " + (wide_var ? tdw : td) + " " + std::string(name) + ";\n";
                // A region behind a rectangle guard is entered rarely (chess: 4 of the 15 a pass tests): unlikely, so that the block
                // placement keeps the skip path as the fall-through and moves the bodies out of line (taken jumps stall on
                // instruction fetch).  A wave-level region of the PIXEL section sits inside a shape whose guard let the wavefront
                // in, and is entered nine times in ten (18 of 20 per pass): likely (board crop 82.3 -> 81.6 us).
                out += "    if (__builtin_expect(" + cond + (pixel && !row_guard ? ", 1)) {\n" : ", 0)) {\n");
                open.push_back(Open{end, typed_bool, nz, next_scope++, false,
                                    (!row_guard && as_bool && va->kind == BOOL) ? (nz ? "~" + va->b : va->b) : std::string(), va->wide});
                ktab_block.clear();
                continue;
            }
            if (op == MARAY_OP_OUT) {
                if (!pixel) { if (out_is_bool.size() <= aux) out_is_bool.resize(aux + 1, 0); out_is_bool[aux] = va->kind == BOOL; }
                const std::string a = dbl(va, "m", i, 0);
                if (!pixel && out_guard_bits && aux >= guard_first) {
                    const uint32_t k = plan ? (uint32_t)plan->pos[aux - guard_first] : aux - guard_first;     // (a derived guard has no job: its OUT is in no cone)
                    out += "    gacc |= (" + a + " != 0.0) ? (1ull << " + std::to_string(k % 8) + ") : 0ull;\n";
                }
                else
                    out += pixel ? "    o" + std::to_string(aux) + " = " + a + ";\n"
                           : stage_first >= 0 ? "    ys[" + std::to_string((aux - (uint32_t)stage_first) * 68) + "u + mr_lane] = " + a + ";\n"
                                              : "    yout[" + std::to_string(aux) + "] = " + a + ";\n";
                continue;
            }
            Val *vb = (op >= MARAY_OP_ADD && op <= MARAY_OP_APP && !forced[i]) ? ref(MARAY_INS_B(ins), 1) : nullptr;
            snprintf(name, sizeof name, "%s%u", prefix, i);
            const std::string self = name;
            Val r;
            std::string e;      // double expression
            std::string be;     // bool expression
            const bool both_bool = va && vb && va->kind == BOOL && vb->kind == BOOL;
            r.wide = pair && ((va && va->wide) || (vb && vb->wide));
            auto m_and = [](const std::string &a, const std::string &b) -> std::string {
                if (a == "MR_NONE" || b == "MR_NONE") return "MR_NONE";
                if (a == "MR_ALL") return b;
                if (b == "MR_ALL") return a;
                return "(" + a + " & " + b + ")";
            };
            auto m_or = [](const std::string &a, const std::string &b) -> std::string {
                if (a == "MR_ALL" || b == "MR_ALL") return "MR_ALL";
                if (a == "MR_NONE") return b;
                if (b == "MR_NONE") return a;
                return "(" + a + " | " + b + ")";
            };
            auto m_not = [](const std::string &a) -> std::string {
                return a == "MR_NONE" ? "MR_ALL" : (a == "MR_ALL" ? "MR_NONE" : "~" + a);
            };
            // The argument of a Sin that may send its tile to the interpreter (huge, inf, NaN, or too close to a multiple of
            // pi/2 for the fast sign): on the lanes where the enclosing regions' result does not depend on what they compute,
            // 0.0 instead -- a texture coordinate runs wild OUTSIDE its shape, where the shape's mask discards the pattern
            // anyway, and without this nearly every tile of a scene of textured shapes was re-rendered by the interpreter
            // (1,000 triangles: 2.2 ms of the frame's 2.4).
            auto quiet_arg = [&](const std::string &x) -> std::string {
                std::string m;
                for (const Open &o : open) if (!o.mask.empty()) { m += (m.empty() ? "" : " & ") + o.mask; if (pair && o.mask_wide) r.wide = true; }      // (a pair of masks makes a pair of arguments)
                return m.empty() ? x : "mr_sel0(" + m + ", " + x + ")";
            };
            // Arithmetic on known constants is done here (the lowering folded what it could see; what is left appears when a
            // variant makes guarded regions literals: a tile without a guard bit paints `0.0 * 255`, and as long as that was a
            // multiply the sky paid a constant's load, its wait and twelve conversions per lane for a colour known beforehand).
            // + * max min neg on doubles are IEEE-exact on the host (lower.cpp folds with the same functions).
            bool folded = false;
            double fold_val = 0.0;
            auto known = [](const Val *v, double *c) -> bool {
                if (!v) return false;
                if (v->cst) { *c = v->cval; return true; }
                if ((v->kind == BOOL || v->kind == NEGBOOL) && (v->b == "MR_NONE" || v->b == "MR_ALL")) {
                    *c = v->b == "MR_ALL" ? 1.0 : 0.0;
                    if (v->kind == NEGBOOL) *c = -*c;
                    return true;
                }
                return false;
            };
            {
                double ca = 0.0, cb = 0.0;
                const bool ka = known(va, &ca), kb = known(vb, &cb);
                if (!forced[i] && role != RedPlan::INNER && role != RedPlan::ROOT) {
                    if (ka && kb && (op == MARAY_OP_ADD || op == MARAY_OP_MUL || op == MARAY_OP_MIN || op == MARAY_OP_MAX)) {
                        folded = true;
                        fold_val = op == MARAY_OP_ADD ? ca + cb : op == MARAY_OP_MUL ? ca * cb : op == MARAY_OP_MAX ? fold_max(ca, cb) : fold_min(ca, cb);
                    } else if (ka && op == MARAY_OP_NEG) { folded = true; fold_val = -ca; }
                }
            }
            if (folded) {
                uint64_t fb; memcpy(&fb, &fold_val, 8);
                if (fb == 0) be = "MR_NONE";                                   // +0.0 and 1.0 stay what constants of the tape are: literal masks
                else if (fb == 0x3ff0000000000000ull) be = "MR_ALL";
                else e = lit(fold_val);
            }
            else if (forced[i]) be = forced[i] == 2 ? "MR_ALL" : "MR_NONE";     // exactly +0.0 / 1.0: a boolean whatever the op
            else if (role == RedPlan::INNER || role == RedPlan::ROOT) ;     // an OR of a reduction: below
            else
            switch (op) {
            case MARAY_OP_MOV: { const bool w_ = r.wide; r = *va; r.wide = w_ || va->wide; break; }
            case MARAY_OP_NEG:
                if (va->kind == BOOL) { r.kind = NEGBOOL; r.b = va->b; }
                else {
                    e = "mr_neg(" + dbl(va, "m", i, 0) + ")";
                    if (va->cmp_kind) { r.cmp_kind = 3 - va->cmp_kind; r.cmp_x = va->cmp_x; r.cmp_k = va->cmp_k; }
                }
                break;
            case MARAY_OP_STEP:
                // Step(x + k) with k a finite constant is the compare x >= -k, bit for bit: fl(x + k) >= 0 iff x + k >= 0 in the reals
                // (rounding is monotone and a sum is never rounded to zero: sums in the subnormal range are exact; an exact zero sum
                // is +0, and Step(+0) = 1 = [x >= -k]); +-inf and NaN agree on both sides.  Step(-(x + k)) likewise is x <= -k
                // (a zero sum gives -0, Step(-0) = 1).  The addition drops out of the chain in front of the compare -- a shape's
                // edge tests are `0 <= u < 1`: Step(u) and Step(u - 1) -- and, where nothing else reads it, altogether.
                if (fuse_cmp && va->kind == DBL && va->cmp_kind && pixel) be = std::string(va->cmp_kind == 1 ? "mr_gek(" : "mr_lek(") + va->cmp_x + ", " + lit(-va->cmp_k) + ")";
                else be = "mr_ge0(" + dbl(va, "m", i, 0) + ")";
                break;
            case MARAY_OP_STEPSIN:
                if ((aux & MARAY_AUX_SIN_BOUNDED) && sin_k >= 0 && td == "double") be = "mr_stepsin_bounded_mk(" + dbl(va, "m", i, 0) + ", mr_kc + " + std::to_string(sin_k) + ")";
                else if (aux & MARAY_AUX_SIN_BOUNDED) be = "mr_stepsin_bounded_m(" + dbl(va, "m", i, 0) + ")";
                else if (pixel && sin_k >= 0 && td == "double") e = "mr_stepsin_fast_k(" + quiet_arg(dbl(va, "m", i, 0)) + ", &mr_defer, mr_kc + " + std::to_string(sin_k) + ")";
                else e = pixel ? "mr_stepsin_fast(" + quiet_arg(dbl(va, "m", i, 0)) + ", &mr_defer)" : "mr_stepsin(" + dbl(va, "m", i, 0) + ")";
                break;
            case MARAY_OP_ADD:
                // 1.0 + (-(b)) = NOT b
                if (va->kind == BOOL && va->b == "MR_ALL" && vb->kind == NEGBOOL) be = m_not(vb->b);
                else if (vb->kind == BOOL && vb->b == "MR_ALL" && va->kind == NEGBOOL) be = m_not(va->b);
                else {
                    e = dbl(va, "m", i, 0) + " + " + dbl(vb, "m", i, 1);
                    for (int q = 0; q < 2; q++) {
                        const Val *c = q ? vb : va, *x = q ? va : vb;
                        if (c->cst && std::isfinite(c->cval) && !x->cst && x->kind == DBL && !x->d.empty()) { r.cmp_kind = 1; r.cmp_x = x->d; r.cmp_k = c->cval; break; }
                    }
                }
                break;
            case MARAY_OP_MUL:
                if (both_bool) be = m_and(va->b, vb->b);
                else e = dbl(va, "m", i, 0) + " * " + dbl(vb, "m", i, 1);
                break;
            case MARAY_OP_MIN:
                if (both_bool) be = m_and(va->b, vb->b);
                else e = "mr_min(" + dbl(va, "m", i, 0) + ", " + dbl(vb, "m", i, 1) + ")";
                break;
            case MARAY_OP_MAX:
                if (both_bool) be = m_or(va->b, vb->b);
                else e = "mr_max(" + dbl(va, "m", i, 0) + ", " + dbl(vb, "m", i, 1) + ")";
                break;
            case MARAY_OP_ABS: e = "mr_abs(" + dbl(va, "m", i, 0) + ")"; break;
            case MARAY_OP_RECIP: e = "mr_recip(" + dbl(va, "m", i, 0) + ")"; break;
            case MARAY_OP_SQRT: e = "mr_sqrt(" + dbl(va, "m", i, 0) + ")"; break;
            case MARAY_OP_SIN:
                e = (aux & MARAY_AUX_SIN_BOUNDED) ? "mr_sin_bounded(" + dbl(va, "m", i, 0) + ")" : "mr_sin(" + (pixel ? quiet_arg(dbl(va, "m", i, 0)) : dbl(va, "m", i, 0)) + ")";
                break;
            case MARAY_OP_EXP: e = "mr_exp(" + dbl(va, "m", i, 0) + ")"; break;
            case MARAY_OP_LN: e = "mr_ln(" + dbl(va, "m", i, 0) + ")"; break;
            case MARAY_OP_APP: {
                const std::string ax = dbl(va, "m", i, 0), ay = dbl(vb, "m", i, 1);
                if (!(pixel && texel_once)) { e = "mr_app(tex, " + std::to_string(aux) + "u, " + ax + ", " + ay + ")"; break; }
                // one texel, three channels: the coordinate work and the address are shared by the App ops of one image on the
                // same two operands (device_math.h, mr_texel); the texel's variable is reused while its block is open
                const std::string key = std::to_string(aux / 5u) + "|" + ax + "|" + ay + "|" + td + (r.wide ? "2" : "");
                auto it = texels.find(key);
                bool in_scope = false;
                if (it != texels.end()) { in_scope = it->second.second == 0; for (const Open &o : open) in_scope |= o.id == it->second.second; }
                if (!in_scope) {
                    const std::string tn = "mr_tl" + std::to_string(i);          // (not mr_tx<i>: mr_tx2 and mr_tx4 are types -- seed 8157 of a compile sweep had its lookup at op 4)
                    // (four pixels per lane: both coordinates may be numbers that are not typed mr_d -- a constant and a y value, say: the texel is
                    // then the same for the four pixels, and without the conversions the call would resolve to the scalar form)
                    const bool four = td == "mr_d";
                    out += "    const " + std::string(four ? "mr_tx4 " : r.wide ? "mr_tx2 " : "mr_tx ") + tn + " = mr_texel(mr_t" + std::to_string(aux / 5u) + ", tex, " +
                           (four ? "mr_d(" + ax + "), mr_d(" + ay + ")" : ax + ", " + ay) + ");\n";
                    it = texels.insert_or_assign(key, std::make_pair(tn, open.empty() ? 0u : open.back().id)).first;
                }
                e = "mr_texch(" + it->second.first + ", " + std::to_string(aux % 5u) + "u)";
                break;
            }
            case MARAY_OP_TEXDIM: e = "mr_texdim(tex, " + std::to_string(aux) + "u)"; break;
            default: throw Error{MARAY_E_ARG, "invalid opcode"};
            }
            if (role == RedPlan::INNER || role == RedPlan::ROOT) {
                // an OR of the tree: its operands are parts of the tree (nothing to do) or free leaves (OR-ed in at the root)
                const int32_t id = rp->red[i];
                const bool rbool = rp->reds[id].boolean;
                for (Val *v : {va, vb}) {
                    if (!v || v->kind == REDPART) continue;
                    const std::string m = !rbool ? dbl(v, "m", i, v == vb) : v->kind == BOOL ? v->b : "mr_ne0(" + dbl(v, "m", i, v == vb) + ")";
                    if (m != "MR_NONE") red_free[id].push_back(m);
                }
                if (role == RedPlan::INNER) {
                    r.kind = REDPART; r.wide = pair;
                    vals[i] = r;
                    is_bool_op[i] = rbool;
                    acc = (int)i;
                    if (dst != MARAY_DST_NONE) slot[dst] = (int)i;
                    continue;
                }
                // the root: the OR of the free leaves and of the guarded leaves whose bit is set in the rectangle's guard words.
                // Per word: the reduction's bits of it, lowest first (s_ff1), each reached through a table of branches that
                // follows an s_setpc (s_getpc returns the address of the instruction after itself: the table starts 12 bytes on)
                const RedPlan::Red &red = rp->reds[id];
                const std::string rid = std::to_string(serial) + "_" + std::to_string(id);
                const std::string racc = "mr_racc" + rid;
                r.wide = pair;                  // the accumulator is a pair whatever the leaves are: a shape may show on one row only
                if (rbool) {
                    out += "    " + std::string(pair ? "mr_pm " : "mr_mask ") + racc + " = MR_NONE";
                    for (const std::string &m : red_free[id]) out += " | " + m;
                    out += ";\n";
                } else {
                    // +0.0 stands for the leaves whose bit is clear; with every bit set there is none: NaN, the identity of max
                    std::string all;
                    for (uint32_t wi = 0; wi < guard_words; wi++) {
                        uint64_t mask = 0;
                        for (uint32_t b : red.leaf_bit) if (b / 64 == wi) mask |= 1ull << (b % 64);
                        if (!mask) continue;
                        char hex[32];
                        snprintf(hex, sizeof hex, "0x%llxull", (unsigned long long)mask);
                        all += std::string(all.empty() ? "" : " && ") + "(" + guard_word(wi) + " & " + hex + ") == " + hex;
                    }
                    out += "    " + std::string(pair ? "mr_p " : "double ") + racc + " = (" + all + ") ? __builtin_nan(\"\") : 0.0;\n";
                    for (const std::string &m : red_free[id]) out += "    " + racc + " = mr_max(" + racc + ", " + m + ");\n";
                }
                for (uint32_t wi = 0; wi < guard_words; wi++) {
                    std::vector<int32_t> leaf_of_bit(64, -1);
                    uint64_t mask = 0;
                    int top = -1;
                    for (size_t k = 0; k < red.leaf_bit.size(); k++)
                        if (red.leaf_bit[k] / 64 == wi) { leaf_of_bit[red.leaf_bit[k] % 64] = (int32_t)k; mask |= 1ull << (red.leaf_bit[k] % 64); top = std::max(top, (int)(red.leaf_bit[k] % 64)); }
                    if (!mask) continue;
                    char hex[32];
                    snprintf(hex, sizeof hex, "0x%llxull", (unsigned long long)mask);
                    const std::string w = std::to_string(wi), next = "mr_rn" + rid + "_" + w;
                    const std::string nz = guard_word_nonzero(wi);
                    if (!nz.empty()) out += "    if (" + nz + ")\n";
                    out += "    for (mr_mask mr_rm = " + guard_word(wi) + " & " + hex + "; mr_rm != 0ull" + (rbool ? " && !mr_covered(" + racc + ")" : std::string()) + "; ) {\n"
                           "        const unsigned mr_rk = (unsigned)__builtin_ctzll(mr_rm);\n"
                           "        mr_rm &= mr_rm - 1ull;\n"
                           "        asm goto(\"s_getpc_b64 s[20:21]\\n\\ts_add_u32 s20, s20, %0\\n\\ts_addc_u32 s21, s21, 0\\n\\ts_setpc_b64 s[20:21]\"";
                    std::string labels;
                    int n_labels = 0;
                    std::vector<int> label_no(red.leaf_bit.size(), -1);
                    for (int b = 0; b <= top; b++) {
                        int ln;
                        if (leaf_of_bit[b] < 0) ln = 0;
                        else { if (label_no[leaf_of_bit[b]] < 0) { label_no[leaf_of_bit[b]] = ++n_labels; labels += ", mr_rl" + rid + "_" + std::to_string(leaf_of_bit[b]); } ln = label_no[leaf_of_bit[b]]; }
                        out += "\n                 \"\\n\\ts_branch %l" + std::to_string(1 + ln) + "\"";
                    }
                    out += "\n                 : : \"s\"(mr_rk * 4u + 12u) : \"s20\", \"s21\", \"scc\" : " + next + labels + ");\n"
                           "        goto " + next + ";              // (not reached: the asm always jumps; `unreachable` here crashes the back end)\n";
                    for (int b = 0; b <= top; b++) {
                        if (leaf_of_bit[b] < 0) continue;
                        std::string yf;              // the leaf's y factors: all must hold on this row
                        for (uint32_t yk : red.leaf_yfactors[leaf_of_bit[b]]) {
                            const std::string k_ = std::to_string(yk);
                            yf += (yf.empty() ? "" : " & ") + (pair ? "mr_pm(mr_ym(yw, " + k_ + "u), mr_ym(yw1, " + k_ + "u))" : "mr_ym(yw, " + k_ + "u)");
                        }
                        out += "    mr_rl" + rid + "_" + std::to_string(leaf_of_bit[b]) + ": {\n" +
                               (yf.empty() ? std::string() : "    if (!mr_any(" + yf + ")) goto " + next + ";      // not on this row (frame 29.5 -> 29.0 us)\n") +
                               red_leaf_text[id][leaf_of_bit[b]] + "    } goto " + next + ";\n";
                    }
                    out += "    " + next + ": ;\n    }\n";
                }
                ktab_block.clear();
                if (rbool) be = racc; else e = racc;
            }
            const bool closes = !open.empty() && open.back().end == i && !open.back().leaf;
            if (closes) {
                ktab_block.clear();
                // the AND / OR that ends a region: assign the variable declared before the `if`
                const Open o = open.back();
                open.pop_back();
                if (pair && i < wide_hint.size()) r.wide = wide_hint[i] != 0;      // (= the width its variable was declared with; the typing run has no hint and finds it)
                if (o.as_bool && !be.empty()) {
                    out += "    b" + self + " = " + be + ";\n    } else b" + self + (o.nz ? " = MR_ALL;\n" : " = MR_NONE;\n");
                    r.kind = BOOL; r.b = "b" + self;
                } else if (o.as_bool) {
                    // guard was boolean but the result is not typed so: keep the double form
                    out += "    b" + self + " = mr_ne0(" + e + ");\n    } else b" + self + (o.nz ? " = MR_ALL;\n" : " = MR_NONE;\n");
                    r.kind = BOOL; r.b = "b" + self;
                } else {
                    const std::string ee = !e.empty() ? e : (be == "MR_NONE" ? "0.0" : be == "MR_ALL" ? "1.0" : "mr_pos(" + be + ")");
                    out += "    " + self + " = " + ee + ";\n    } else " + self + (o.nz ? " = 1.0;\n" : " = 0.0;\n");
                    r.kind = DBL; r.d = self;
                }
                while (!open.empty() && open.back().end == i && !open.back().leaf) {     // enclosing regions that end here too
                    const Open o2 = open.back();
                    open.pop_back();
                    out += o.as_bool ? "    } else b" + self + (o2.nz ? " = MR_ALL;\n" : " = MR_NONE;\n")
                                     : "    } else " + self + (o2.nz ? " = 1.0;\n" : " = 0.0;\n");
                }
            } else if (be == "MR_NONE" || be == "MR_ALL") {
                r.kind = BOOL; r.b = be; r.wide = false;      // a literal: later ops fold it
            } else if (!be.empty() && be[0] != '(' && be[0] != '~' && be.compare(0, 3, "mr_") != 0) {
                r.kind = BOOL; r.b = be;             // folded to one of its operands: an alias, no new variable
            } else if (!be.empty()) {
                out += "    const " + (r.wide ? tmw : tm) + " b" + self + " = " + be + ";\n";
                r.kind = BOOL; r.b = "b" + self;
            } else if (!e.empty() && folded) {
                r.kind = DBL; r.d = e; r.cst = true; r.cval = fold_val; r.wide = false;      // a literal: no statement
            } else if (!e.empty()) {
                out += "    const " + (r.wide ? tdw : td) + " " + self + " = " + e + ";\n";
                r.kind = DBL; r.d = self;
            }
            if (role == RedPlan::LEAF_END) {
                // the leaf's block ends: its mask joins the reduction's accumulator; the text goes to its place behind the table
                if (open.empty() || !open.back().leaf || open.back().end != i) throw Error{MARAY_E_INTERNAL, "reduction leaf out of step"};
                open.pop_back();
                const int32_t id = rp->red[i];
                const std::string ra = "mr_racc" + std::to_string(serial) + "_" + std::to_string(id);
                if (rp->reds[id].boolean) out += "    " + ra + " |= " + (r.kind == BOOL ? r.b : "mr_ne0(" + dbl(&r, "m", i, 0) + ")") + ";\n";
                else out += "    " + ra + " = mr_max(" + ra + ", " + dbl(&r, "m", i, 0) + ");\n";
                red_leaf_text[id][rp->leaf[i]].swap(out);
                out.swap(out_saved);
                out_saved.clear();
                ktab_block.clear();
                r = Val();
                r.kind = REDPART; r.wide = pair;
                vals[i] = r;
                is_wide_op[i] = pair;
                is_bool_op[i] = 1;
                acc = (int)i;
                if (dst != MARAY_DST_NONE) slot[dst] = (int)i;
                continue;
            }
            vals[i] = r;
            is_bool_op[i] = r.kind == BOOL;
            is_wide_op[i] = r.wide;
            acc = (int)i;
            if (dst != MARAY_DST_NONE) slot[dst] = (int)i;
        }
    }
};

}   // namespace

}   // namespace maray
