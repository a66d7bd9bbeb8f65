// jit_backend.cpp — tape -> straight-line HIP -> gfx950 code object via hiprtc
// (product code).  The GPU analogue of the reference's wasmer JIT:
//
//   jit_source   <- wasm::gen_expr / gen_vars / Wasm::from_expr   src/wasm.rs:77-158
//                   (Expr -> WAT text -> compiled module; one `(local $id f64)`
//                   per Let variable)
//   JitBackend   <- wasm_par_gen_to_image                         src/render.rs:102-192
//
// Differences by design: the input is the lowered tape (one DAG shared by R, G
// and B, constants folded, Y-only work hoisted into a per-row kernel), every op
// is an inline f64 instruction (the reference's JIT calls host imports for
// abs/recip/step/sin/exp/ln, src/wasm.rs:40-47), max/min follow the interpreter
// (f64::max/min), not wasm's NaN-propagating f64.max/min (src/wasm.rs:58-59).
//
// Generated code: every tape op becomes `const double vN = op(...)`; value
// slots and ACC disappear (the compiler allocates registers), constants become
// exact hex-float literals, y values are scalar loads from the row table.
// Compiled with -ffp-contract=off so no a*b+c is fused behind the tape's back.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <sys/stat.h>
#include <fcntl.h>
#include <csignal>
#include <unistd.h>
#include <sstream>
#include <spawn.h>
#include <dlfcn.h>
#include <sys/wait.h>
#include <cerrno>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "backend.hpp"
#include "host_pipe.hpp"
#include "maray_hip.h"

extern "C" const char maray_build_id[];          // _obj/build_id.cpp (Makefile): a hash of this library's sources
extern "C" const char maray_embedded_device_math_h[];
extern "C" const char maray_embedded_libm_h[];
extern "C" const char maray_embedded_libm_tables_h[];

// of jit_compile; part of the cache key.  -structurizecfg-skip-uniform-regions: the generated kernels branch on wave-uniform
// conditions throughout (lane masks tested on the scalar unit), and the branch table of a guarded OR-reduction (an asm
// goto inside a loop) only survives when the structurizer leaves uniform regions alone: without the option the back end
// rewrites the table's edges into tests of flags nobody sets
static const char JIT_OPTIONS[] = "--offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -mllvm -structurizecfg-skip-uniform-regions";

namespace maray {

namespace {

// Can the pixel kernel defer tiles to the interpreter?  Only a Sin / Step(Sin) whose argument is not proven bounded can.
bool may_defer_tiles(const maray_program &P)
{
    for (uint32_t i = 0; i < P.n_pix_ops; i++) {
        const uint32_t op = MARAY_INS_OP(P.pix_ops[i]);
        if ((op == MARAY_OP_SIN || op == MARAY_OP_STEPSIN) && !(MARAY_INS_AUX(P.pix_ops[i]) & MARAY_AUX_SIN_BOUNDED)) return true;
    }
    return false;
}

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
struct GuardGeom { uint32_t gw, gh; };          // the rectangle a guard is bounded over: gw pixels x gh rows (jit_guard_geom)

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
        is_bool_op.assign(n, 0);
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
                if (idx < ybool.size() && ybool[idx]) { t.kind = BOOL; t.b = "mr_ym(yw, " + std::to_string(idx) + "u)"; }
                return &t;
            default:
                if (idx == MARAY_SPEC_ACC) return &vals[acc];
                static const char *const spec_name[] = {"X", "Y", "", "XMAX", "XMIN", "YMAX", "YMIN"};
                t.d = spec_name[idx];
                return &t;
            }
        };
        // the double form of a value, materialising it once if needed
        // leaf: the block of a reduction's leaf (no variable, no else).  mask: the lanes on which what the region computes can
        // matter -- a wave-level region of booleans computes q of n = p AND q (p OR q): where p is 0 (1), n does not depend on q
        struct Open { uint32_t end; bool as_bool; bool nz; uint32_t id; bool leaf; std::string mask; };
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
            out += "    const " + td + " ";
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
                open.push_back(Open{i + aux, true, false, next_scope++, true, std::string()});
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
                    // a y value is uniform over the block: test its bits on the scalar unit, no ballot, no VALU
                    const std::string k = std::to_string(MARAY_REF_INDEX(gref));
                    cond = nz ? "!(yw[2 * " + k + " + 1] == 0x3ff00000u && yw[2 * " + k + "] == 0u)"
                              : "((yw[2 * " + k + " + 1] << 1) | yw[2 * " + k + "]) != 0u";
                } else if (as_bool) cond = nz ? "mr_any(~" + va->b + ")" : "mr_any(" + va->b + ")";      // a scalar compare
                else cond = (nz ? "mr_any(mr_ne1(" : "mr_any(mr_ne0(") + dbl(va, "m", i, 0) + "))";
                // several regions may end at one op (a row-level guard around a wave-level one): one variable
                bool typed_bool = as_bool;
                bool declared = false;
                for (const Open &o : open) if (o.end == end && !o.leaf) { declared = true; typed_bool = o.as_bool; }
                if (!declared) out += typed_bool ? "    " + tm + " b" + std::string(name) + ";\n" : "    " + td + " " + std::string(name) + ";\n";
                // A region behind a rectangle guard is entered rarely (chess: 4 of the 15 a pass tests): unlikely, so that the block
                // placement keeps the skip path as the fall-through and moves the bodies out of line (taken jumps stall on
                // instruction fetch).  A wave-level region of the PIXEL section sits inside a shape whose guard let the wavefront
                // in, and is entered nine times in ten (18 of 20 per pass): likely (board crop 82.3 -> 81.6 us).
                out += "    if (__builtin_expect(" + cond + (pixel && !row_guard ? ", 1)) {\n" : ", 0)) {\n");
                open.push_back(Open{end, typed_bool, nz, next_scope++, false,
                                    (!row_guard && as_bool && va->kind == BOOL) ? (nz ? "~" + va->b : va->b) : std::string()});
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
                for (const Open &o : open) if (!o.mask.empty()) m += (m.empty() ? "" : " & ") + o.mask;
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
            case MARAY_OP_MOV: r = *va; break;
            case MARAY_OP_NEG:
                if (va->kind == BOOL) { r.kind = NEGBOOL; r.b = va->b; }
                else e = "mr_neg(" + dbl(va, "m", i, 0) + ")";
                break;
            case MARAY_OP_STEP: be = "mr_ge0(" + dbl(va, "m", i, 0) + ")"; break;
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
                else e = dbl(va, "m", i, 0) + " + " + dbl(vb, "m", i, 1);
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
            case MARAY_OP_APP: e = "mr_app(tex, " + std::to_string(aux) + "u, " + dbl(va, "m", i, 0) + ", " + dbl(vb, "m", i, 1) + ")"; break;
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
                    r.kind = REDPART;
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
                if (rbool) {
                    out += "    mr_mask " + racc + " = MR_NONE";
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
                    out += "    double " + racc + " = (" + all + ") ? __builtin_nan(\"\") : 0.0;\n";
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
                    out += "    for (mr_mask mr_rm = " + guard_word(wi) + " & " + hex + "; mr_rm != 0ull" + (rbool ? " && " + racc + " != MR_ALL" : std::string()) + "; ) {\n"
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
                        for (uint32_t yk : red.leaf_yfactors[leaf_of_bit[b]]) yf += (yf.empty() ? "" : " & ") + std::string("mr_ym(yw, ") + std::to_string(yk) + "u)";
                        out += "    mr_rl" + rid + "_" + std::to_string(leaf_of_bit[b]) + ": {\n" +
                               (yf.empty() ? std::string() : "    if ((" + yf + ") == MR_NONE) goto " + next + ";      // not on this row (frame 29.5 -> 29.0 us)\n") +
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
                r.kind = BOOL; r.b = be;             // a literal: later ops fold it
            } else if (!be.empty() && be[0] != '(' && be[0] != '~' && be.compare(0, 3, "mr_") != 0) {
                r.kind = BOOL; r.b = be;             // folded to one of its operands: an alias, no new variable
            } else if (!be.empty()) {
                out += "    const " + tm + " b" + self + " = " + be + ";\n";
                r.kind = BOOL; r.b = "b" + self;
            } else if (!e.empty() && folded) {
                r.kind = DBL; r.d = e; r.cst = true; r.cval = fold_val;      // a literal: no statement
            } else if (!e.empty()) {
                out += "    const " + td + " " + self + " = " + e + ";\n";
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
                r.kind = REDPART;
                vals[i] = r;
                is_bool_op[i] = 1;
                acc = (int)i;
                if (dst != MARAY_DST_NONE) slot[dst] = (int)i;
                continue;
            }
            vals[i] = r;
            is_bool_op[i] = r.kind == BOOL;
            acc = (int)i;
            if (dst != MARAY_DST_NONE) slot[dst] = (int)i;
        }
    }
};

}   // namespace

// Which y values are booleans: a dry run of the emitter over the ROW section (its typing is the one the PIXEL section
// will rely on).
std::vector<uint8_t> jit_bool_yvals(const maray_program &P)
{
    std::vector<uint8_t> r(P.n_yvals, 0);
    if (!P.n_row_ops) return r;
    Emitter D(P);
    D.section(P.row_ops, P.n_row_ops, P.n_row_slots, false, "r");
    const uint32_t n_ynum = numeric_yvals(P);
    for (uint32_t k = 0; k < P.n_yvals && k < D.out_is_bool.size() && k < n_ynum; k++) r[k] = D.out_is_bool[k];
    return r;
}

// The guard plan of a program (GuardPlan).  A guard is derived when its source is a MAX tree, boolean-typed all the way
// (on {+0.0, 1.0} max is OR, so "value != 0" distributes over it exactly), whose leaves are sources of other guards.
// Bits are handed out in the order the members are met, so that a group's bits are neighbours (one word, one s_and).
GuardPlan jit_guard_plan(const maray_program &P)
{
    GuardPlan gp;
    const uint32_t n_ynum = numeric_yvals(P), n_guards = P.n_yvals - n_ynum;
    gp.pos.assign(n_guards, -1);
    gp.members.assign(n_guards, {});
    const bool derive = true;
    const RowTapeDeps d = row_tape_deps(P);
    std::vector<int32_t> src(n_guards, -1);                    // op that produces a guard's value
    std::unordered_map<int32_t, uint32_t> guard_of;            // op -> (first) guard it is the source of
    for (uint32_t o : d.outs) {
        const uint32_t aux = MARAY_INS_AUX(P.row_ops[o]);
        if (aux < n_ynum) continue;
        src[aux - n_ynum] = d.deps[o][0];
        if (d.deps[o][0] >= 0) guard_of.emplace(d.deps[o][0], aux - n_ynum);
    }
    std::vector<uint8_t> isb;
    if (derive && P.n_row_ops) {
        Emitter D(P);
        D.section(P.row_ops, P.n_row_ops, P.n_row_slots, false, "r");
        isb = D.is_bool_op;
    }
    // leaves of guard g's MAX tree; false if it is not one
    std::vector<std::vector<uint32_t>> kids(n_guards);
    std::vector<uint8_t> derived(n_guards, 0);
    for (uint32_t g = 0; g < n_guards && derive; g++) {
        const int32_t s0 = src[g];
        if (s0 < 0 || MARAY_INS_OP(P.row_ops[s0]) != MARAY_OP_MAX || !isb[s0]) continue;
        std::vector<int32_t> st = {d.deps[s0][0], d.deps[s0][1]};
        std::vector<uint32_t> leaves;
        bool ok = true;
        while (ok && !st.empty()) {
            const int32_t o = st.back(); st.pop_back();
            if (o < 0) { ok = false; break; }
            auto it = guard_of.find(o);
            if (it != guard_of.end() && it->second != g) { leaves.push_back(it->second); continue; }
            if (MARAY_INS_OP(P.row_ops[o]) == MARAY_OP_MAX && isb[o]) { st.push_back(d.deps[o][1]); st.push_back(d.deps[o][0]); continue; }
            ok = false;
        }
        if (ok && !leaves.empty() && leaves.size() <= 64) { derived[g] = 1; kids[g] = leaves; }
    }
    // bits: members of a derived guard first, in tree order (recursively: a member may be derived itself), then the rest
    std::function<void(uint32_t, std::vector<uint32_t> &)> place = [&](uint32_t g, std::vector<uint32_t> &into) {
        if (derived[g]) {
            if (gp.members[g].empty()) for (uint32_t k : kids[g]) place(k, gp.members[g]);
            into.insert(into.end(), gp.members[g].begin(), gp.members[g].end());
            return;
        }
        if (gp.pos[g] < 0) gp.pos[g] = (int32_t)gp.n_pos++;
        into.push_back((uint32_t)gp.pos[g]);
    };
    std::vector<uint32_t> sink;
    // dearest groups first: their members end up contiguous
    std::vector<uint32_t> order(n_guards);
    for (uint32_t g = 0; g < n_guards; g++) order[g] = g;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return kids[a].size() > kids[b].size(); });
    for (uint32_t g : order) { sink.clear(); place(g, sink); }
    return gp;
}

// The ROW section split into chunks that different wavefronts evaluate side by side.  One
// work-item per row is all the parallelism a straight-line ROW kernel has (4096 rows = 64 waves,
// each walking thousands of dependent f64 ops: ~45 us for chess, an eighth of the frame).  The y
// values are independent outputs, so the tape is cut by outputs: chunk k keeps the ops its outputs
// depend on (ops two chunks share are computed in both) and the rest become NOPs.
// Chunk k writes the y values [first[k], first[k] + count[k]): outputs are taken in index order, so that a chunk's
// values are neighbours in a row of the table and leave the kernel as full cache lines (jit_source_rows).
struct RowChunks {
    std::vector<std::vector<uint64_t>> tapes;
    std::vector<uint32_t> first, count;
};
static const uint32_t ROW_CHUNK_MAX_OUTS = 16;      // x 68 x 8 B of LDS per wavefront

RowChunks split_row_tape(const maray_program &P, const RowTapeDeps &d, uint32_t out_limit)
{
    const uint32_t n = P.n_row_ops;
    std::vector<uint32_t> outs;
    for (uint32_t j : d.outs) if (MARAY_INS_AUX(P.row_ops[j]) < out_limit) outs.push_back(j);
    std::sort(outs.begin(), outs.end(), [&](uint32_t x, uint32_t y) { return MARAY_INS_AUX(P.row_ops[x]) < MARAY_INS_AUX(P.row_ops[y]); });
    size_t total = 0;
    (void)row_tape_cone(P, d, outs, &total);
    const size_t per_chunk = 64;
    const uint32_t n_chunks = (uint32_t)std::min<size_t>(64, std::max<size_t>(std::max<size_t>(1, total / per_chunk), (outs.size() + ROW_CHUNK_MAX_OUTS - 1) / ROW_CHUNK_MAX_OUTS));
    const size_t budget = (total + n_chunks - 1) / n_chunks;
    RowChunks rc;
    size_t next = 0;
    while (next < outs.size()) {
        std::vector<uint32_t> mine;
        size_t cost = 0;
        while (next < outs.size() && mine.size() < ROW_CHUNK_MAX_OUTS && cost < budget) {
            // a y value with no OUT of its own between two others would break the chunk's index range: a new chunk starts there
            if (!mine.empty() && MARAY_INS_AUX(P.row_ops[outs[next]]) != MARAY_INS_AUX(P.row_ops[mine.back()]) + 1) break;
            mine.push_back(outs[next++]);
            (void)row_tape_cone(P, d, mine, &cost);
        }
        rc.first.push_back(MARAY_INS_AUX(P.row_ops[mine.front()]));
        rc.count.push_back((uint32_t)mine.size());
        rc.tapes.push_back(row_tape_cone(P, d, mine, &cost));
    }
    if (rc.tapes.empty()) { rc.tapes.emplace_back(n, 0); rc.first.push_back(0); rc.count.push_back(0); }
    return rc;
}

// How the specialised kernels use the row guards of a program: as bits, 64 per word, one set per
// 256-pixel tile of a row (guard_words = 0: not at all -- none, far too many, or switched off).  Up to 12 words a tile's
// words sit in SGPRs; beyond, a guard test reads its word from LDS.
uint32_t jit_guard_words(const maray_program &P)
{
    if (!jit_row_guards_enabled() || P.n_yvals == numeric_yvals(P)) return 0;
    const uint32_t nw = (jit_guard_plan(P).n_pos + 63) / 64;
    return nw <= 1024 ? std::max(nw, 1u) : 0;       // 1024 words x 8 tiles = 64 KB of LDS
}

static const uint32_t GW_INLINE_MAX = 12;       // up to this many guard words a rectangle's words are named SGPR pairs; beyond, they stay one per lane (v_readlane per test)

// The rectangle a guard is bounded over: `gh` rows x `gw` pixels.  gh = 1 (and gw = 256) when some guard's cone reads Y;
// else every guard bounds its boolean over the rows [YMIN, YMAX] too (include/maray_tape.h) and the rectangle is the
// back-end's choice.  The number of rectangles is what the ROW kernel pays for, their shape is what the PIXEL kernel
// gains from: a shape's edge is met by ~(extent / side + 1) rectangles each way, a wavefront enters regions per 64
// pixels of ONE row, and shapes are tall against 8 rows -- so a rectangle that is narrower and taller by the same factor
// costs the ROW kernel nothing and spares the PIXEL kernel region entries.  Default 64 x 32 (chess @4096^2, frame / board
// crop in us, 256 x 8: 49.3 / 104; 256 x 16: 48.9 / 104; 128 x 16: 45.1 / 92; 64 x 8: 49.0 / 83 -- four times the guard work;
// 64 x 16: 45.5 / 84; 64 x 32: 43.8 / 84; 64 x 64: 45.1 / 86; 64 x 128: 48.1 / 88).  MARAY_JIT_GUARD_W = 64 / 128 / 256,
// MARAY_JIT_GUARD_H = 8 ... 128: measurement knobs.  A strip's words are held one per lane, so a tile's rectangles together
// have to fit a wavefront's 64 lanes: a program with many guard words gets wider rectangles.
GuardGeom jit_guard_geom(const maray_program &P)
{
    GuardGeom g{256u, 1u};
    const uint32_t nw = jit_guard_words(P);
    if (!nw || any_guard_reads_y(P)) return g;
    g.gh = 32u;
    const char *env_h = getenv("MARAY_JIT_GUARD_H");
    if (env_h) { const int v = atoi(env_h); if (v == 8 || v == 16 || v == 32 || v == 64 || v == 128) g.gh = (uint32_t)v; }
    uint32_t want = 64u;
    if (const char *e_ = getenv("MARAY_JIT_GUARD_W")) { const int v = atoi(e_); if (v == 64 || v == 128 || v == 256) want = (uint32_t)v; }
    while (want < 256u && nw * (256u / want) > 64u) want *= 2u;
    if (want == 256u && !env_h) g.gh = 8u;      // wide rectangles gain nothing from height (chess, 256 x 8 / 256 x 32: 48.6 / 50.2 us per frame)
    g.gw = want;
    return g;
}

static const unsigned ROW_BLOCK = 256;          // threads per block of the ROW kernel (64 ... 1024 move a chess frame by less than a microsecond)

// Source of the ROW kernel, maray_jit_rows: one wavefront per block, blockIdx.y picks the job.
//  y < n_chunks: chunk y of the ROW section, one work-item per row; writes the y values the pixel
//    kernel reads as operands (and, for a program that may defer tiles to the interpreter, the
//    guards too, bounded over the whole row as the interpreter expects).
//  y >= n_chunks: guards 8 (y - n_chunks) .. +7, one work-item per rectangle (group of `yrows` rows,
//    run of jit_guard_geom().gw pixels), evaluated with XMIN / XMAX = the run's ends and YMIN / YMAX = the group's
//    (a bound over a rectangle skips far more than one over the row); writes its byte of
//    the rectangle's guard words (64 guards per word).  yrows = 1 when some guard reads Y.  Small
//    jobs on purpose: each is one long dependent chain, and only more wavefronts hide that.
// One launch for both: the few y-value wavefronts run in the shadow of the guard ones.
// Plain device_math.h: the rare huge-argument tail of sin is a real (out-of-line) call here.
std::string jit_source_rows(const maray_program &P, uint32_t *n_chunks_out, uint32_t *n_gjobs_out)
{
    validate_program(P);
    Emitter E(P);
    const RowTapeDeps deps = row_tape_deps(P);
    const uint32_t n_ynum = numeric_yvals(P), n_gwords = jit_guard_words(P);
    const GuardPlan plan = jit_guard_plan(P);
    const GuardGeom geom = jit_guard_geom(P);
    // wave-level SKIP ops of the ROW section: a wavefront's lanes are 64 rows, or the 64 rectangles of a band of rows, and
    // agree on the sky only; a job is one wavefront's chain, and every short region it has to test and branch around
    // lengthens it (chess, step minus pixel kernel in us, regions kept from 0 / 12 / 24 / 60 / 200 instructions / none:
    // 7.2 / 6.9 / 6.3 / 6.8 / 8.3 / 8.3)
    E.min_region_row = 24;
    const uint32_t n_gjobs = n_gwords ? (plan.n_pos + 7) / 8 : 0;                // 8 bits = one byte of a word per job
    if (n_gjobs_out) *n_gjobs_out = n_gjobs;
    // the interpreter (which drains deferred tiles from the same y-value table) does read the guard values
    const uint32_t out_limit = may_defer_tiles(P) ? 0xFFFFFFFFu : n_ynum;
    const RowChunks rc = split_row_tape(P, deps, out_limit);
    const std::vector<std::vector<uint64_t>> &chunks = rc.tapes;
    if (n_chunks_out) *n_chunks_out = (uint32_t)chunks.size();
    std::string &s = E.out;
    s += "// generated by libmaray_hip (jit_backend.cpp): ROW section, " + std::to_string(P.n_row_ops) + " ops; y values in " +
         std::to_string(chunks.size()) + " chunks, " + std::to_string(P.n_yvals - n_ynum) + " guards in " + std::to_string(n_gwords) + " words,\n"
         "// each bounded over rectangles of " + std::to_string(geom.gw) + " pixels x " + std::to_string(geom.gh) + " rows (the height is a launch parameter, and part of the code key through this line:\n"
         "// a cached code object carries its geometry)\n";
    s += "#include \"device_math.h\"\n\n";
    // (constants stay literals here: from a table in constant memory like the PIXEL kernel's, the code is a tenth shorter
    // and the kernel 0.8 us slower -- the loads' waits sit in the one chain a job is -- and spills to scratch)
    const unsigned row_block = ROW_BLOCK;
    s += "extern \"C\" __global__ void __launch_bounds__(" + std::to_string(row_block) + ") maray_jit_rows(double *__restrict__ yvals, unsigned long long *__restrict__ gbits,\n"
         "                                                                 const MarayTex *__restrict__ tex,\n"
         "                                                                 unsigned y0, unsigned rows, unsigned n_yvals, unsigned w, unsigned n_tx,\n"
         "                                                                 unsigned blk_rows, unsigned blk_stride, unsigned yrows)\n{\n"
         "    const unsigned item = blockIdx.x * blockDim.x + threadIdx.x;         // (the host keeps the items of a launch below 2^32)\n"
         "    (void)tex; (void)gbits; (void)n_tx;\n"
         "    // guard jobs first in the grid (they are the long ones: the y-value jobs fill in behind them): job = the switch index\n"
         "    const unsigned mr_job = blockIdx.y < " + std::to_string(n_gjobs) + "u ? " + std::to_string(chunks.size()) + "u + blockIdx.y : blockIdx.y - " + std::to_string(n_gjobs) + "u;\n"
         "    if (mr_job < " + std::to_string(chunks.size()) + "u) {\n"
         +
         "    // y values: a work-item per row.  A lane's values go to LDS ([value][row], values 68 apart) and leave as rows of the\n"
         "    // table, a chunk's values side by side: full cache lines.  Stored from the registers, a\n"
         "    // wavefront's store touches 64 lines for 8 bytes each -- 1.2 M partial writes per frame, which is what the kernel\n"
         "    // then waits for (11.8 us; the arithmetic needs 2).\n"
         "    __shared__ double mr_ys[" + std::to_string(row_block / 64) + " * " + std::to_string(ROW_CHUNK_MAX_OUTS * 68) + "];\n"
         "    const unsigned mr_lane = threadIdx.x & 63u;\n"
         "    double *ys = mr_ys + (threadIdx.x >> 6) * " + std::to_string(ROW_CHUNK_MAX_OUTS * 68) + "u;\n"
         "    const unsigned row0 = item - mr_lane;                                 // first row of this wavefront\n"
         "    if (row0 >= rows) return;                                            // whole wavefronts only: every lane helps to store\n"
         "    const unsigned r = item;\n"
         "    const double Y = (double)(blk_stride == 0u ? y0 + r : y0 + (r / blk_rows) * blk_stride + r % blk_rows), XMIN = 0.0, XMAX = (double)(w - 1u);\n"
         "    const double YMIN = Y, YMAX = Y;\n"
         "    (void)Y; (void)XMIN; (void)XMAX; (void)YMIN; (void)YMAX; (void)yrows;\n"
         "    unsigned mr_k0 = 0u, mr_kn = 0u;\n"
         "    switch (mr_job) {\n";
    for (size_t k = 0; k < chunks.size(); k++) {
        s += "    case " + std::to_string(k) + ": {\n";
        E.stage_first = (int)rc.first[k];
        E.section(chunks[k].data(), P.n_row_ops, P.n_row_slots, false, "r");
        E.stage_first = -1;
        s += "    mr_k0 = " + std::to_string(rc.first[k]) + "u; mr_kn = " + std::to_string(rc.count[k]) + "u;\n    } break;\n";
    }
    s += "    }\n"
         "    __builtin_amdgcn_wave_barrier();                                       // same wavefront: LDS keeps its order\n"
         "    // lane 16 q + j stores value j of the rows 4 i + q, i = 0 .. 15: an instruction writes four rows of the chunk\n"
         "    const unsigned mr_j = mr_lane & 15u, mr_q = mr_lane >> 4;\n"
         "    double *mr_dst = yvals + (size_t)(row0 + mr_q) * n_yvals + mr_k0 + mr_j;\n"
         "    const size_t mr_step = (size_t)4u * n_yvals;\n"
         "    if (mr_j < mr_kn) {\n"
         "        _Pragma(\"unroll\") for (unsigned i = 0; i < 16u; i++)\n"
         "            if (row0 + 4u * i + mr_q < rows) mr_dst[i * mr_step] = ys[mr_j * 68u + 4u * i + mr_q];\n"
         "    }\n"
         "    return;\n    }\n";
    if (n_gwords) {
        s += "    // guards: (row group, tile), the tiles of a group adjacent\n"
             "    const unsigned n_groups = (rows + yrows - 1u) / yrows;\n"
             "    if (item >= n_groups * n_tx) return;\n"
             "    const unsigned grp = item / n_tx, tile = item - grp * n_tx;\n"
             "    const unsigned r = grp * yrows, r_last = r + yrows - 1u < rows - 1u ? r + yrows - 1u : rows - 1u;      // launch rows of the group\n"
             "    // (n_tx counts rectangles here; the last 256-pixel tile of a ragged row may own rectangles past the edge: they bound the last pixel)\n"
             "    const unsigned xlo_ = tile * " + std::to_string(geom.gw) + "u, xlo = xlo_ < w - 1u ? xlo_ : w - 1u, xhi = xlo_ + " + std::to_string(geom.gw - 1) + "u < w - 1u ? xlo_ + " + std::to_string(geom.gw - 1) + "u : w - 1u;\n"
             "    // a group never straddles two row blocks (the host picks yrows | blk_rows), so its image rows are consecutive\n"
             "    const double Y = (double)(blk_stride == 0u ? y0 + r : y0 + (r / blk_rows) * blk_stride + r % blk_rows), XMIN = (double)xlo, XMAX = (double)xhi;\n"
             "    const double YMIN = Y, YMAX = Y + (double)(r_last - r);\n"
             "    unsigned long long gacc = 0ull;\n"
             "    double *yout = nullptr;\n"
             "    (void)Y; (void)XMIN; (void)XMAX; (void)YMIN; (void)YMAX; (void)yout;\n"
             "    switch (mr_job - " + std::to_string(chunks.size()) + "u) {\n";
        E.out_guard_bits = true;
        E.guard_first = n_ynum;
        E.plan = &plan;
        for (uint32_t j = 0; j < n_gjobs; j++) {
            std::vector<uint32_t> outs;          // the guards whose bits are 8 j .. 8 j + 7 (derived guards have none: no job computes them)
            for (uint32_t o : deps.outs) {
                const uint32_t aux = MARAY_INS_AUX(P.row_ops[o]);
                if (aux < n_ynum) continue;
                const int32_t k = plan.pos[aux - n_ynum];
                if (k >= (int32_t)(8 * j) && k < (int32_t)(8 * (j + 1))) outs.push_back(o);
            }
            const std::vector<uint64_t> tape = row_tape_cone(P, deps, outs, nullptr);
            s += "    case " + std::to_string(j) + ": {\n";
            E.section(tape.data(), P.n_row_ops, P.n_row_slots, false, "r");
            s += "    } break;\n";
        }
        s += "    }\n"
             "    ((unsigned char *)gbits)[(size_t)item * " + std::to_string(8 * n_gwords) + "u + (mr_job - " + std::to_string(chunks.size()) + "u)] = (unsigned char)gacc;\n";
    }
    s += "}\n";
    // Launch order of the PIXEL kernel: groups of rows by what they cost, dearest first, so that the tail of the launch is
    // made of cheap blocks.  Cost of a group = set bits in the guard words of its rectangles (shapes that may show there).
    // The bits are a function of the program and of the launch's geometry only: the order is computed once per geometry
    // (one block; rank by counting) and reused.  Rows of a group stay neighbours (they share guard words and cache lines).
    if (n_gwords)
        s += "extern \"C\" __global__ void __launch_bounds__(256) maray_jit_order(const unsigned long long *__restrict__ gbits, unsigned *__restrict__ order,\n"
             "                                                                  unsigned rows, unsigned n_tx, unsigned yrows)\n{\n"
             "    extern __shared__ unsigned mr_cost[];\n"
             "    const unsigned n_groups = (rows + yrows - 1u) / yrows, n_full = rows / yrows, per = n_tx * " + std::to_string(n_gwords) + "u;\n"
             "    for (unsigned g = threadIdx.x; g < n_groups; g += 256u) {\n"
             "        unsigned c = 0;\n"
             "        for (unsigned i = 0; i < per; i++) c += (unsigned)__builtin_popcountll(gbits[(size_t)g * per + i]);\n"
             "        mr_cost[g] = c;\n"
             "    }\n"
             "    __syncthreads();\n"
             "    for (unsigned g = threadIdx.x; g < n_full; g += 256u) {\n"
             "        const unsigned c = mr_cost[g];\n"
             "        unsigned rank = 0;\n"
             "        for (unsigned h = 0; h < n_full; h++) rank += (mr_cost[h] > c || (mr_cost[h] == c && h < g)) ? 1u : 0u;\n"
             "        for (unsigned i = 0; i < yrows; i++) order[rank * yrows + i] = g * yrows + i;\n"
             "    }\n"
             "    for (unsigned r = n_full * yrows + threadIdx.x; r < rows; r += 256u) order[r] = r;      // a partial last group stays last\n"
             "}\n";
    return s;
}

// The general section four pixels per lane: only a short program without guards whose ops are single instructions (no libm
// bodies, no gathers).
bool jit_wide_general(const maray_program &P, uint32_t n_gwords)        // n_gwords = jit_guard_words(P) (a walk over the ROW tape: the caller has it)
{
    bool heavy = false;
    for (uint32_t i = 0; i < P.n_pix_ops; i++) {
        const uint32_t op = MARAY_INS_OP(P.pix_ops[i]);
        heavy |= op == MARAY_OP_SIN || op == MARAY_OP_EXP || op == MARAY_OP_LN || op == MARAY_OP_STEPSIN || op == MARAY_OP_APP;
    }
    return n_gwords == 0 && !heavy && P.n_pix_slots <= 6 && P.n_pix_ops <= 256;
}

// Source of the PIXEL kernel, maray_jit_pixels.  A wavefront owns a strip of `tiles` consecutive 256-pixel tiles of one
// row (blockIdx.y); a block is four wavefronts = four neighbouring strips that share nothing but the instruction cache:
// no staging, no barrier.  The strip's guard words arrive with one vector load (lane i = word i); per tile one scalar
// test of a ballot picks the variant:
//
//  * WIDE, four pixels per lane (device_math.h, MR_VEC4: every value four f64, every boolean four lane masks).  The
//    variant of a tile none of whose guard bits is set (every guarded region is the literal 0: for chess the background,
//    one multiply), and the whole section of a small program without guards (config 2: six ops).  The scalar unit's
//    share of a tile and the store's address arithmetic are paid once per 256 pixels, and a lane's four RGB8 pixels are
//    12 contiguous bytes: one global_store_dwordx3, no cross-lane packing.  This is the path that is bound by the store
//    (3 B per pixel) and little else.
//  * NARROW, one pixel per lane, four passes of 64 pixels (a loop: the section's code exists once).  The variant of a
//    tile where shapes may show.  Regions are entered per 64 pixels, where a wave-level SKIP op still finds all lanes
//    agreeing; values are single f64.  The passes leave their packed pixels in LDS (same-wave traffic: no barrier) and
//    the tile is stored like a wide one.
//
// When f64 planes are wanted too, element e of a wide lane l is pixel x0 + 64 e + l and every 64-pixel run is stored on
// its own (24 B per lane, the coalesced pattern of the f64 planes).  A Sin whose argument is huge (|x| >= 105414350), inf
// or NaN does not call the slow reduction here (a call site per Sin op would force every live value through scratch): the
// tile is flagged instead and re-evaluated by the tape interpreter kernel afterwards, so the final raster is identical.
// Layouts that were measured and lost (a wavefront per 64 pixels with guard words staged in LDS, a busy tile on the
// block's four wavefronts side by side, persistent wavefronts, two pixels per lane, guard words by scalar loads, a
// sky loop of its own ...) are history: DESIGN.md section 7.1, profiles/r2_ablations.jsonl.
std::string jit_source(const maray_program &P, int min_waves)
{
    validate_program(P);
    // 6 waves per SIMD, i.e. up to 102 SGPRs (at 8 the compiler gets 76 and spills ~400 of them to VGPR lanes, in the skeleton
    // of bit tests and branches every pass walks; chess needs 38 VGPRs either way and runs 7 waves per SIMD)
    const int min_waves_arg = min_waves;
    Emitter E(P);
    // Wave-level SKIP ops over fewer than 12 instructions' worth of ops are ignored: a busy tile is bound by the scalar unit
    // (branches, bit tests, mask algebra: 0.59 SALU instructions per cycle and CU against 35 % VALU issue), and a short
    // region's test and branch cost that unit more than its ops cost the vector one (chess board, us per 16.7 Mpx, with
    // guards per 256 x 8 pixels: none ignored 111, 24: 104; with guards per 64 x 32: 8 / 12 / 16 ... 32 / 64 / 200:
    // 83.3 / 82.4 / 84.7 / 85.4 / 128).
    E.min_region = 12;
    if (const char *e_ = getenv("MARAY_JIT_MIN_REGION")) E.min_region = (uint32_t)atoi(e_);
    E.ybool = jit_bool_yvals(P);
    E.ktab = true;
    for (uint32_t i = 0; i < P.n_pix_ops && E.sin_k < 0; i++)
        if (MARAY_INS_OP(P.pix_ops[i]) == MARAY_OP_STEPSIN) {
            // the first cache line of the table: what every leaf with a texture reads
            static const double sin_k[8] = {0x1.45f306dc9c883p-1, 0x1.8p52, 0x1.921fb58000000p+0, -0x1.dde973c000000p-27, -0x1.cb3b398000000p-55, -0x1.d747f23e32ed7p-83, 0x1p-70, 0.0};
            E.sin_k = 0;
            E.ktab_vals.assign(sin_k, sin_k + 8);
        }
    std::string &s = E.out;
    const uint32_t n_ynum = numeric_yvals(P);
    const uint32_t n_gwords = jit_guard_words(P);
    E.ignore_row_guards = n_gwords == 0;
    const GuardPlan plan = jit_guard_plan(P);
    const GuardGeom geom = jit_guard_geom(P);
    const uint32_t sub = 256u / geom.gw;                   // guard rectangles per 256-pixel tile (> 1: their words are taken per pass)
    const std::string tw = std::to_string(sub * n_gwords);  // guard words per tile
    if (n_gwords) { E.guard_first = n_ynum; E.guard_words = n_gwords; E.plan = &plan; E.gw_inline_max = GW_INLINE_MAX; }
    RedPlan reductions;
    if (!E.ignore_row_guards) {         // dry run: which ops yield lane masks
        Emitter D(P);
        D.ignore_row_guards = true;
        D.min_region = E.min_region;
        D.ybool = E.ybool;
        D.section(P.pix_ops, P.n_pix_ops, P.n_pix_slots, true, "v");
        E.bool_hint = D.is_bool_op;
        // OR trees of guarded shapes: evaluated from their set guard bits (RedPlan).  MARAY_JIT_REDUCE=0: walked as written (ablation)
        const char *e_ = getenv("MARAY_JIT_REDUCE");
        if (!(e_ && e_[0] == '0')) reductions = plan_reductions(P.pix_ops, P.n_pix_ops, P.n_pix_slots, D.is_bool_op, n_ynum, plan, E.ybool);
    }
    // Occupancy asked of the compiler.  Walking a tree of bit tests needs the SGPRs of 6 waves per SIMD (up to 102; at 8 the
    // compiler gets 80 and spilled ~400 of them to VGPR lanes, in the skeleton every pass walked); with the tree evaluated as
    // a reduction chess fits 78 and runs 8 (frame 29.9 -> 29.5 us, sky 12.3 -> 11.5, board 67.1 -> 65.1)
    if (min_waves_arg == 0) min_waves = reductions.empty() ? 6 : 8;
    const bool defer = may_defer_tiles(P);
    const std::string nw = std::to_string(n_gwords);
    // a strip's guard words: one vector load per wavefront (lane i holds word i), then v_readlane per tile or pass -- one
    // memory latency per strip instead of one per tile
    const bool gw_vgpr = n_gwords && n_gwords <= GW_INLINE_MAX;
    const bool wide_general = jit_wide_general(P, n_gwords);
    const std::string esub = "(e >> " + std::to_string(sub == 4 ? 0 : 1) + "u)";       // rectangle of pass e inside its tile
    s += "// generated by libmaray_hip (jit_backend.cpp) from a v" + std::to_string(P.version) + " tape: PIXEL section, " +
         std::to_string(P.n_pix_ops) + " ops; general variant " + (wide_general ? "four pixels per lane" : "one pixel per lane, four passes per tile") + "\n"
         "#define MR_VEC4 1\n"
         "__shared__ unsigned mr_slow[4];           // per wavefront: some Sin of the tile at hand needs the slow path\n"
         "__shared__ unsigned mr_tp[4 * 256];       // per wavefront: the packed pixels of a tile's four passes\n"
         "__device__ inline double mr_defer_sin(double) { ((volatile unsigned *)mr_slow)[threadIdx.x >> 6] = 1u; return 0.0; }\n"
         "#define MR_SIN_HUGE(x) mr_defer_sin(x)   // plain Sin ops: flag the tile from the (rare) branch\n"
         "#include \"device_math.h\"\n"
         "typedef const __attribute__((address_space(4))) double *mr_kptr;\n"
         "struct __attribute__((aligned(4))) mr_u3 { unsigned a, b, c; };\n"
         "struct __attribute__((aligned(16))) mr_u4 { unsigned a, b, c, d; };\n"
         "__device__ inline mr_mask mr_lane64(unsigned long long v, unsigned lane)      // lane `lane` (wave-uniform) of a per-lane 64-bit value -> SGPR pair\n"
         "{\n"
         "    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, (int)lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), (int)lane);\n"
         "    return ((mr_mask)hi << 32) | lo;\n"
         "}\n/*MR_KTAB*/\n";
    s += "extern \"C\" __global__ void __launch_bounds__(256, " + std::to_string(min_waves) +
         ") maray_jit_pixels(unsigned char *__restrict__ rgb8, double *__restrict__ rgb64,\n"
         "                                                                    const double *__restrict__ yvals, const MarayTex *__restrict__ tex,\n"
         "                                                                    unsigned *__restrict__ tile_list, unsigned tile_base,\n"
         "                                                                    const unsigned long long *__restrict__ gbits, unsigned n_tx,\n"
         "                                                                    unsigned w, unsigned y0, unsigned n_yvals, unsigned tiles,\n"
         "                                                                    unsigned blk_rows, unsigned blk_stride, unsigned row_base, unsigned yrows,\n"
         "                                                                    const unsigned *__restrict__ row_order)\n{\n"
         "    const unsigned mr_wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), mr_lane = threadIdx.x & 63u;\n"
         "    // (workgroups go to the 8 XCDs round robin by their linear id: with 2, 4 or 8 blocks per row a column of the image\n"
         "    // always meets the same XCDs; rotating a row's strips by the row was measured and is not worth it, DESIGN.md 7.1)\n"
         "    const unsigned tile0 = (blockIdx.x * 4u + mr_wv) * tiles;               // this wavefront's strip of the row\n"
         "    if (tile0 >= n_tx) return;\n"
         "    const unsigned r = row_order ? row_order[blockIdx.y] : blockIdx.y;     // row of this launch (dearest groups of rows first); row_base + r = row of the whole call\n"
         "    const unsigned long long mr_ybase0 = (unsigned long long)(yvals + (size_t)r * n_yvals);\n"
         "    // -> image row (RowBlocks); one range of rows (blk_stride == 0) needs no division, and yrows is a power of two\n"
         "    const double Y = (double)(blk_stride == 0u ? y0 + row_base + r : y0 + ((row_base + r) / blk_rows) * blk_stride + (row_base + r) % blk_rows);\n"
         "    (void)Y; (void)tex; (void)gbits; (void)yrows; (void)tile_list; (void)tile_base;\n";
    if (n_gwords)
        s += "    const unsigned long long mr_gbase0 = (unsigned long long)(gbits + ((size_t)((row_base + r) >> __builtin_ctz(yrows)) * n_tx + tile0) * " + tw + "u);\n";
    if (gw_vgpr)
        s += "    const unsigned mr_gn = (n_tx - tile0 < tiles ? n_tx - tile0 : tiles) * " + tw + "u;       // <= 64: the host bounds `tiles`\n"
             "    const unsigned long long mr_gv = mr_lane < mr_gn ? ((const unsigned long long *)mr_gbase0)[mr_lane] : 0ull;\n" +
             (sub > 1 ? "    const unsigned long long mr_gnz = mr_ballot(mr_gv != 0ull);            // which of the strip's words have a bit set\n" : "");
    s += "    const bool mr_wide = rgb64 == nullptr;                                   // wide variants: element e of lane l is pixel x0 + 4 l + e, else x0 + 64 e + l\n"
         "    const unsigned mr_xl = mr_wide ? 4u * mr_lane : mr_lane, mr_xs = mr_wide ? 1u : 64u;\n"
         "    const unsigned mr_src = (mr_lane * 4u) / 3u, mr_shift = ((mr_lane * 4u) % 3u) * 8u;   // RGB8 packing of one 64-pixel run\n"
         "    const size_t row_px = (size_t)r * w;\n"
         "    for (unsigned t = 0; t < tiles; t++) {\n"
         "    const unsigned x0 = (tile0 + t) * 256u;\n"
         "    if (x0 >= w) break;\n"
         "    // y values, constants, guard words: scalar loads where they are used, from addresses made opaque in every trip (fresh\n"
         "    // copies: an asm output carried around the loop counts as divergent once a lane-dependent branch sits in the loop)\n"
         "/*MR_KBASE*/";
    if (n_gwords && !gw_vgpr) {
        s += "    unsigned long long mr_gbase = mr_gbase0;\n"
             "    asm volatile(\"\" : \"+s\"(mr_gbase));\n";
        if (sub > 1)   // many words, narrow rectangles: lane i of mr_gt0 holds word i of the tile's rectangles (<= 64 together, jit_guard_geom)
            s += "    unsigned long long mr_gt0 = mr_lane < " + tw + "u ? ((const unsigned long long *)mr_gbase)[t * " + tw + "u + mr_lane] : 0ull;\n";
        else           // lane i of mr_gt<j> holds word 64 j + i of this tile (one vector load each); a test takes its word with v_readlane
            for (uint32_t j = 0; j < (n_gwords + 63) / 64; j++)
                s += "    unsigned long long mr_gt" + std::to_string(j) + " = " + std::to_string(64 * j) + "u + mr_lane < " + nw + "u ? ((const unsigned long long *)mr_gbase)[t * " + nw + "u + " +
                     std::to_string(64 * j) + "u + mr_lane] : 0ull;\n";
    } else if (gw_vgpr && sub == 1)
        for (uint32_t j = 0; j < n_gwords; j++)
            s += "    mr_mask gq" + std::to_string(j) + " = mr_lane64(mr_gv, t * " + nw + "u + " + std::to_string(j) + "u);\n";
    if (defer) s += "    ((volatile unsigned *)mr_slow)[mr_wv] = 0u;\n    bool mr_slow_tile = false;\n";
    // what opens a pass of either width: the tables made opaque (LICM would hoist every constant and y value out of the
    // loops and spill them), the pixel coordinates, the outputs
    const std::string opaque =
        "    unsigned long long mr_ybase = mr_ybase0;\n"
        "    asm volatile(\"\" : \"+s\"(mr_ybase));\n"
        "    mr_kptr yv = (mr_kptr)mr_ybase;\n"
        "    const __attribute__((address_space(4))) unsigned *yw = (const __attribute__((address_space(4))) unsigned *)yv;\n"
        "    (void)yv; (void)yw;\n/*MR_KC*/";
    // the guard words of the rectangle at hand, opaque anew in every pass: left visible, all their bit tests are loop
    // invariants too (168 booleans for chess, hoisted and spilled to VGPR lanes)
    std::string gq_pass;
    if (n_gwords && !gw_vgpr && sub > 1) {
        gq_pass = "    asm volatile(\"\" : \"+v\"(mr_gt0));\n"
                  "    const unsigned mr_gsub = " + esub + " * " + nw + "u;           // first word of this pass's rectangle\n"
                  "    const unsigned long long mr_gnzp = mr_ballot(mr_gt0 != 0ull) >> mr_gsub;      // which of its words have a bit set\n"
                  "    (void)mr_gnzp;\n";
        E.gw_lane_base = "mr_gsub";
    } else if (gw_vgpr && sub > 1)
        for (uint32_t j = 0; j < n_gwords; j++) {
            const std::string k = std::to_string(j);
            gq_pass += "    mr_mask gq" + k + " = mr_lane64(mr_gv, (t * " + std::to_string(sub) + "u + " + esub + ") * " + nw + "u + " + k + "u);\n"
                       "    asm volatile(\"\" : \"+s\"(gq" + k + "));\n";
        }
    else if (gw_vgpr)
        for (uint32_t j = 0; j < n_gwords; j++) gq_pass += "    asm volatile(\"\" : \"+s\"(gq" + std::to_string(j) + "));\n";
    else if (n_gwords)
        for (uint32_t j = 0; j < (n_gwords + 63) / 64; j++) {
            const std::string k = std::to_string(j);
            gq_pass += "    asm volatile(\"\" : \"+v\"(mr_gt" + k + "));\n"
                       "    const unsigned long long mr_gnz" + k + " = mr_ballot(mr_gt" + k + " != 0ull);      // which of the tile's words have a bit set\n"
                       "    (void)mr_gnz" + k + ";\n";
        }
    const std::string wide_open =
        "    {\n" + opaque + (sub > 1 ? std::string() : gq_pass) +
        "    const unsigned xa = x0 + mr_xl;                                        // this lane's first pixel\n"
        "    const mr_d X((double)xa, (double)(xa + mr_xs), (double)(xa + 2u * mr_xs), (double)(xa + 3u * mr_xs));\n"
        "    mr_d o0 = 0.0, o1 = 0.0, o2 = 0.0;\n"
        "    float mr_defer = 0.0f;                     // fused Step(Sin) ops count their undecided cases in here\n"
        "    (void)X; (void)mr_defer;\n";
    const std::string defer_pass = defer ?
        "    mr_slow_tile |= mr_ballot(mr_defer != 0.0f) != 0ull || ((volatile unsigned *)mr_slow)[mr_wv] != 0u;      // wave-uniform\n" : "";
    const std::string wide_close = defer_pass +
        "    const unsigned p0 = mr_cast_u8(o0.a) | (mr_cast_u8(o1.a) << 8) | (mr_cast_u8(o2.a) << 16);\n"
        "    const unsigned p1 = mr_cast_u8(o0.b) | (mr_cast_u8(o1.b) << 8) | (mr_cast_u8(o2.b) << 16);\n"
        "    const unsigned p2 = mr_cast_u8(o0.c) | (mr_cast_u8(o1.c) << 8) | (mr_cast_u8(o2.c) << 16);\n"
        "    const unsigned p3 = mr_cast_u8(o0.d) | (mr_cast_u8(o1.d) << 8) | (mr_cast_u8(o2.d) << 16);\n"
        "    if (mr_wide) {\n"
        "        if (rgb8) {\n"
        "            unsigned char *q = rgb8 + (row_px + xa) * 3;                      // this lane's 12 bytes\n"
        "            if (x0 + 256u <= w && ((size_t)(rgb8 + (row_px + x0) * 3) & 3u) == 0u) {     // wave-uniform: a whole tile whose bytes start on a dword\n"
        "                mr_u3 d;\n"
        "                d.a = p0 | (p1 << 24); d.b = (p1 >> 8) | (p2 << 16); d.c = (p2 >> 16) | (p3 << 8);\n"
        "                *(mr_u3 *)q = d;\n"
        "            } else {\n"
        "                const unsigned pk[4] = {p0, p1, p2, p3};\n"
        "                for (unsigned e = 0; e < 4u; e++)\n"
        "                    if (xa + e < w) { q[3 * e] = (unsigned char)pk[e]; q[3 * e + 1] = (unsigned char)(pk[e] >> 8); q[3 * e + 2] = (unsigned char)(pk[e] >> 16); }\n"
        "            }\n"
        "        }\n"
        "    } else {\n"
        "        const unsigned pk[4] = {p0, p1, p2, p3};\n"
        "        const double c0[4] = {o0.a, o0.b, o0.c, o0.d}, c1[4] = {o1.a, o1.b, o1.c, o1.d}, c2[4] = {o2.a, o2.b, o2.c, o2.d};\n"
        "        _Pragma(\"unroll\") for (unsigned e = 0; e < 4u; e++)\n"
        "            mr_store_run(rgb8, rgb64, row_px, x0 + 64u * e, w, mr_lane, mr_src, mr_shift, pk[e], c0[e], c1[e], c2[e]);\n"
        "    }\n"
        "    }\n";
    // one 64-pixel run: f64 planes (24 B per lane) and / or RGB8 (48 lanes assemble a dword each from two neighbours'
    // packed colours; ragged ends and unaligned rows store bytes)
    const std::string store_run =
        "__device__ inline void mr_store_run(unsigned char *__restrict__ rgb8, double *__restrict__ rgb64, size_t row_px, unsigned xw, unsigned w,\n"
        "                                    unsigned lane, unsigned src, unsigned shift, unsigned pk, double c0, double c1, double c2)\n{\n"
        "    const unsigned x = xw + lane;\n"
        "    if (rgb64 && x < w) { const size_t p = (row_px + x) * 3; rgb64[p] = c0; rgb64[p + 1] = c1; rgb64[p + 2] = c2; }\n"
        "    if (rgb8) {\n"
        "        unsigned char *wave_out = rgb8 + (row_px + xw) * 3;\n"
        "        if (xw + 64u <= w && ((size_t)wave_out & 3u) == 0u) {                // wave-uniform\n"
        "            const unsigned pa = (unsigned)__builtin_amdgcn_ds_bpermute((int)(src * 4u), (int)pk);\n"
        "            const unsigned pb = (unsigned)__builtin_amdgcn_ds_bpermute((int)(src * 4u + 4u), (int)pk);\n"
        "            const unsigned dw = (unsigned)((((unsigned long long)pb << 24) | pa) >> shift);\n"
        "            if (lane < 48u) ((unsigned *)wave_out)[lane] = dw;\n"
        "        } else if (x < w) {\n"
        "            unsigned char *q = rgb8 + (row_px + x) * 3;\n"
        "            q[0] = (unsigned char)pk; q[1] = (unsigned char)(pk >> 8); q[2] = (unsigned char)(pk >> 16);\n"
        "        }\n"
        "    }\n"
        "}\n";
    std::string tile_end;         // closes a tile: the work list entry of a tile some Sin of which needs the slow path
    if (defer)
        tile_end = "    if (mr_slow_tile && mr_lane == 0u) tile_list[1u + atomicAdd(&tile_list[0], 1u)] = tile_base + r * n_tx + tile0 + t;\n";

    if (n_gwords) {
        // the variant of a tile with no guard bit set, four pixels per lane
        if (!gw_vgpr && sub > 1)
            s += "    if (mr_ballot(mr_gt0 != 0ull) == 0ull) {\n";
        else if (sub > 1)
            s += "    if (((mr_gnz >> (t * " + tw + "u)) & " + std::to_string((1ull << (sub * n_gwords)) - 1ull) + "ull) == 0ull) {\n";
        else if (gw_vgpr) {
            std::string any = "gq0";
            for (uint32_t j = 1; j < n_gwords; j++) any += " | gq" + std::to_string(j);
            s += "    if ((" + any + ") == 0ull) {\n";
        } else {
            std::string any = "mr_gt0";
            for (uint32_t j = 1; j < (n_gwords + 63) / 64; j++) any += " | mr_gt" + std::to_string(j);
            s += "    if (mr_ballot((" + any + ") != 0ull) == 0ull) {\n";
        }
        E.td = "mr_d"; E.tm = "mr_m";
        E.assume_guards_zero = true;
        s += wide_open;
        E.section(P.pix_ops, P.n_pix_ops, P.n_pix_slots, true, "v");
        s += wide_close;
        E.assume_guards_zero = false;
        s += tile_end + "    continue;\n    }\n";
    }
    if (wide_general) {
        E.td = "mr_d"; E.tm = "mr_m";
        s += wide_open;
        E.section(P.pix_ops, P.n_pix_ops, P.n_pix_slots, true, "v");
        s += wide_close + tile_end;
    } else {
        // one wavefront, four passes of 64 pixels (a loop, not unrolled); the passes leave their packed pixels in LDS
        // (same-wave traffic: no barrier) and a whole aligned tile is stored as a dwordx3 per lane
        E.td = "double"; E.tm = "mr_mask";
        s += "    const bool mr_fast = rgb8 && x0 + 256u <= w && ((size_t)(rgb8 + (row_px + x0) * 3) & 3u) == 0u;      // wave-uniform\n"
             "    _Pragma(\"unroll 1\") for (unsigned e = 0; e < 4u; e++) {\n" + opaque + gq_pass +
             "    const unsigned xw = x0 + 64u * e, x = xw + mr_lane;\n"
             "    const double X = (double)x;\n"
             "    double o0 = 0.0, o1 = 0.0, o2 = 0.0;\n"
             "    float mr_defer = 0.0f;\n"
             "    (void)X; (void)mr_defer;\n";
        E.rplan = &reductions;
        E.section(P.pix_ops, P.n_pix_ops, P.n_pix_slots, true, "v");
        E.rplan = nullptr;
        s += defer_pass +
             "    const unsigned pk = mr_cast_u8(o0) | (mr_cast_u8(o1) << 8) | (mr_cast_u8(o2) << 16);\n"
             "    if (mr_fast) {\n"
             "        mr_tp[mr_wv * 256u + 64u * e + mr_lane] = pk;\n"
             "        mr_store_run(nullptr, rgb64, row_px, xw, w, mr_lane, mr_src, mr_shift, pk, o0, o1, o2);\n"
             "    } else mr_store_run(rgb8, rgb64, row_px, xw, w, mr_lane, mr_src, mr_shift, pk, o0, o1, o2);\n"
             "    }\n"
             "    if (mr_fast) {\n"
             "        __builtin_amdgcn_wave_barrier();                                     // same wavefront wrote them: LDS keeps its order\n"
             "        const mr_u4 p = *(const mr_u4 *)&mr_tp[mr_wv * 256u + 4u * mr_lane];\n"
             "        mr_u3 d;\n"
             "        d.a = p.a | (p.b << 24); d.b = (p.b >> 8) | (p.c << 16); d.c = (p.c >> 16) | (p.d << 8);\n"
             "        *(mr_u3 *)(rgb8 + (row_px + x0 + 4u * mr_lane) * 3) = d;\n"
             "        __builtin_amdgcn_wave_barrier();\n"
             "    }\n" + tile_end;
    }
    s += "    }\n}\n";
    {
        std::string tab = store_run;
        if (!E.ktab_vals.empty()) {
            tab += "__constant__ __attribute__((aligned(64))) double mr_kc_tab[" + std::to_string(E.ktab_vals.size()) + "] = {";
            for (size_t j = 0; j < E.ktab_vals.size(); j++) { tab += (j % 6 ? " " : "\n    "); tab += lit(E.ktab_vals[j]); tab += ","; }
            tab += "\n};\n";
        }
        s.replace(s.find("/*MR_KTAB*/"), 11, tab);
        // The table's address, made opaque once per tile and once per pass (see `opaque`)
        for (size_t at; (at = s.find("/*MR_KBASE*/")) != std::string::npos;)
            s.replace(at, 12, E.ktab_vals.empty() ? "" : "    unsigned long long mr_kbase = (unsigned long long)mr_kc_tab;\n");
        const std::string kc = E.ktab_vals.empty() ? "    asm volatile(\"\" ::: \"memory\");\n" :
                               "    asm volatile(\"\" : \"+s\"(mr_kbase) :: \"memory\");\n"
                               "    const mr_kptr mr_kc = (mr_kptr)mr_kbase;\n";
        for (size_t at; (at = s.find("/*MR_KC*/")) != std::string::npos;) s.replace(at, 9, kc);
    }
    return s;
}

#define RTC_TRY(expr)                                                                              \
    do {                                                                                           \
        hiprtcResult r_ = (expr);                                                                  \
        if (r_ != HIPRTC_SUCCESS)                                                                  \
            throw Error{MARAY_E_HIP, std::string(#expr) + ": " + hiprtcGetErrorString(r_)};         \
    } while (0)

void jit_compile(const std::string &src, std::vector<char> &code, std::string &log)
{
    hiprtcProgram prog;
    const char *headers[] = {maray_embedded_device_math_h, maray_embedded_libm_h, maray_embedded_libm_tables_h};
    const char *names[] = {"device_math.h", "maray_libm.h", "maray_libm_tables.h"};
    RTC_TRY(hiprtcCreateProgram(&prog, src.c_str(), "maray_jit.hip", 3, headers, names));
    const char *olevel = getenv("MARAY_JIT_OPT");          // "-O1" builds faster (1.7 against 2.6 s for chess, kernel 37.4 against 35.6 us)
    std::vector<const char *> opts = {"--offload-arch=gfx950", (olevel && olevel[0] == '-') ? olevel : "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-mllvm", "-structurizecfg-skip-uniform-regions"};
    std::vector<std::string> extra;                        // MARAY_JIT_EXTRA="-mllvm -some-flag ...": measurement knob
    if (const char *e_ = getenv("MARAY_JIT_EXTRA")) { std::istringstream in(e_); for (std::string w; in >> w;) extra.push_back(w); }
    for (const std::string &w : extra) opts.push_back(w.c_str());
    hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    size_t ln = 0;
    hiprtcGetProgramLogSize(prog, &ln);
    log.assign(ln, '\0');
    if (ln) hiprtcGetProgramLog(prog, &log[0]);
    if (rc != HIPRTC_SUCCESS) {
        hiprtcDestroyProgram(&prog);
        throw Error{MARAY_E_HIP, std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(rc) + "\n" + log};
    }
    size_t n = 0;
    RTC_TRY(hiprtcGetCodeSize(prog, &n));
    code.resize(n);
    RTC_TRY(hiprtcGetCode(prog, code.data()));
    hiprtcDestroyProgram(&prog);
}

// ---- code objects: built once per (program, toolchain), kept in the process and on disk ------------------------
//
// The reference's JIT compiles its three modules again on every thread of every render (src/render.rs:158-165).
// Here a program's two code objects (PIXEL and ROW kernels) are a pure function of the generated sources, the
// compiler options and the hiprtc that builds them: they are built once, shared by every context of the process
// (one per device of a multi-GPU render: the second device loads what the first one built), and stored under
// MARAY_CACHE_DIR (default $XDG_CACHE_HOME/maray_amd or ~/.cache/maray_amd; "off" disables) for the next process.
namespace {

uint64_t fnv1a(const void *data, size_t n, uint64_t h)
{
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

// an unsigned field of the kernel's metadata note (msgpack: the value follows its key)
long code_meta_uint(const std::vector<char> &co, const char *key)
{
    const size_t kn = strlen(key);
    const auto it = std::search(co.begin(), co.end(), key, key + kn);
    if (it == co.end() || (size_t)(co.end() - it) < kn + 5) return -1;
    const unsigned char *v = (const unsigned char *)&*it + kn;
    if (v[0] <= 0x7f) return v[0];
    if (v[0] == 0xcc) return v[1];
    if (v[0] == 0xcd) return ((long)v[1] << 8) | v[2];
    if (v[0] == 0xce) return ((long)v[1] << 24) | ((long)v[2] << 16) | ((long)v[3] << 8) | v[4];
    return -1;
}

std::string cache_dir()
{
    const char *e = getenv("MARAY_CACHE_DIR");
    if (e) {
        if (!e[0] || !strcmp(e, "off") || !strcmp(e, "0")) return "";
        return e;
    }
    if (const char *x = getenv("XDG_CACHE_HOME")) if (x[0]) return std::string(x) + "/maray_amd";
    if (const char *h = getenv("HOME")) if (h[0]) return std::string(h) + "/.cache/maray_amd";
    return "";
}

void mkdirs(const std::string &d)
{
    for (size_t i = 1; i <= d.size(); i++)
        if (i == d.size() || d[i] == '/') (void)mkdir(d.substr(0, i).c_str(), 0777);
}

const uint32_t CACHE_MAGIC = 0x3263726du;    // "mrc2": 9 header words (guard geometry in the header)

bool cache_read(const std::string &path, JitCode &c)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint32_t hdr[9];
    bool ok = fread(hdr, 4, 9, f) == 9 && hdr[0] == CACHE_MAGIC && hdr[4] < (1u << 30) && hdr[5] < (1u << 30);
    if (ok) {
        c.n_row_chunks = hdr[1]; c.n_gjobs = hdr[2]; c.waves = (int)hdr[3];
        c.n_gwords = hdr[6]; c.guard_w = hdr[7]; c.guard_h = hdr[8];
        c.pix.resize(hdr[4]); c.rows.resize(hdr[5]);
        ok = fread(c.pix.data(), 1, c.pix.size(), f) == c.pix.size() && fread(c.rows.data(), 1, c.rows.size(), f) == c.rows.size();
        uint64_t sum = 0;
        ok = ok && fread(&sum, 8, 1, f) == 1 && sum == fnv1a(c.rows.data(), c.rows.size(), fnv1a(c.pix.data(), c.pix.size(), 0xcbf29ce484222325ull));
    }
    fclose(f);
    return ok;
}

void cache_write(const std::string &dir, const std::string &path, const JitCode &c)
{
    mkdirs(dir);
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;                                       // a read-only or missing cache directory is not an error
    const uint32_t hdr[9] = {CACHE_MAGIC, c.n_row_chunks, c.n_gjobs, (uint32_t)c.waves, (uint32_t)c.pix.size(), (uint32_t)c.rows.size(),
                             c.n_gwords, c.guard_w, c.guard_h};
    const uint64_t sum = fnv1a(c.rows.data(), c.rows.size(), fnv1a(c.pix.data(), c.pix.size(), 0xcbf29ce484222325ull));
    const bool ok = fwrite(hdr, 4, 9, f) == 9 && fwrite(c.pix.data(), 1, c.pix.size(), f) == c.pix.size() &&
                    fwrite(c.rows.data(), 1, c.rows.size(), f) == c.rows.size() && fwrite(&sum, 8, 1, f) == 1;
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());      // rename: readers never see half a file
}

std::mutex g_code_mutex;
std::map<std::string, std::shared_future<std::shared_ptr<const JitCode>>> g_code;
std::vector<std::string> g_code_order;              // oldest first: the table keeps the 32 most recent programs (the rest live on disk)

// The code key names the code objects of a program: a hash of the two generated sources, the embedded headers and the
// build options -- whatever changes the kernels changes it.  Generating the sources costs ~0.1 s for chess, so the key of
// a program is remembered under a cheaper name: a hash of the program itself (constants, both sections), the generator
// (MARAY_BUILD_ID: a hash of this library's sources, made by the Makefile), every MARAY_JIT_* knob of the environment and
// the compiler's version -- in the process and, next to the code objects, in <cache>/<name>.key.
struct CodeKey { std::string hex, src_pix, src_rows; uint32_t n_row_chunks = 1, n_gjobs = 0; bool have_src = false; };

std::string hex128(uint64_t h1, uint64_t h2)
{
    char buf[40];
    snprintf(buf, sizeof buf, "%016llx%016llx", (unsigned long long)h1, (unsigned long long)h2);
    return buf;
}

std::string hiprtc_path();

std::string key_salt()
{
    int major = 0, minor = 0;
    (void)hiprtcVersion(&major, &minor);          // a process that imported PyTorch first compiles with PyTorch's own hiprtc
    // ... and two builds of one version are two compilers: the library's path, size and modification time name the binary
    std::string rtc_id = hiprtc_path();
    struct stat st;
    if (!rtc_id.empty() && stat(rtc_id.c_str(), &st) == 0) rtc_id += ":" + std::to_string((long long)st.st_size) + ":" + std::to_string((long long)st.st_mtime);
    // the library's build id: what a launch does with the kernels (tiles per wavefront, grid shape) is library code, and a
    // profile stamped with a code key has to mean "these kernels, launched this way"
    return std::string(maray_version()) + "|" + maray_build_id + "|hiprtc " + std::to_string(major) + "." + std::to_string(minor) + " " + rtc_id +
           "|" + JIT_OPTIONS + "|" + (getenv("MARAY_JIT_OPT") ? getenv("MARAY_JIT_OPT") : "") + (getenv("MARAY_JIT_EXTRA") ? std::string("|") + getenv("MARAY_JIT_EXTRA") : std::string());
}

std::string program_name(const maray_program &prog)
{
    std::string salt = key_salt() + "|" + maray_build_id;
    std::vector<std::string> knobs;
    for (char **e = environ; e && *e; e++) if (!strncmp(*e, "MARAY_JIT_", 10)) knobs.push_back(*e);
    std::sort(knobs.begin(), knobs.end());
    for (const std::string &kn : knobs) salt += "|" + kn;
    const uint32_t counts[8] = {prog.version, prog.n_consts, prog.n_row_ops, prog.n_row_slots, prog.n_yvals, prog.n_pix_ops, prog.n_pix_slots, prog.n_app};
    uint64_t h[2] = {0xcbf29ce484222325ull, 0x84222325cbf29ce4ull};
    for (uint64_t &x : h) {
        x = fnv1a(salt.data(), salt.size(), x);
        x = fnv1a(counts, sizeof counts, x);
        x = fnv1a(prog.consts, (size_t)prog.n_consts * sizeof(double), x);
        x = fnv1a(prog.row_ops, (size_t)prog.n_row_ops * sizeof(uint64_t), x);
        x = fnv1a(prog.pix_ops, (size_t)prog.n_pix_ops * sizeof(uint64_t), x);
    }
    return hex128(h[0], h[1]);
}

void key_sources(const maray_program &prog, CodeKey &k)
{
    if (k.have_src) return;
    k.src_pix = jit_source(prog);
    if (prog.n_row_ops) k.src_rows = jit_source_rows(prog, &k.n_row_chunks, &k.n_gjobs);
    k.have_src = true;
}

std::mutex g_name_mutex;
std::map<std::string, std::string> g_names;           // program name -> code key

CodeKey code_key(const maray_program &prog)
{
    CodeKey k;
    const std::string name = program_name(prog), dir = cache_dir();
    {
        std::lock_guard<std::mutex> lk(g_name_mutex);
        auto it = g_names.find(name);
        if (it != g_names.end()) { k.hex = it->second; return k; }
    }
    const std::string path = dir.empty() ? "" : dir + "/" + name + ".key";
    if (!path.empty())
        if (FILE *f = fopen(path.c_str(), "rb")) {
            char buf[33] = {0};
            const bool ok = fread(buf, 1, 32, f) == 32 && strspn(buf, "0123456789abcdef") == 32;
            fclose(f);
            if (ok) k.hex = buf;
        }
    if (k.hex.empty()) {
        key_sources(prog, k);
        const std::string salt = key_salt();
        uint64_t h1 = fnv1a(salt.data(), salt.size(), 0xcbf29ce484222325ull), h2 = fnv1a(salt.data(), salt.size(), 0x84222325cbf29ce4ull);
        for (const std::string *t : {&k.src_pix, &k.src_rows}) { h1 = fnv1a(t->data(), t->size() + 1, h1); h2 = fnv1a(t->data(), t->size() + 1, h2); }
        for (const char *hd : {maray_embedded_device_math_h, maray_embedded_libm_h, maray_embedded_libm_tables_h}) { h1 = fnv1a(hd, strlen(hd), h1); h2 = fnv1a(hd, strlen(hd), h2); }
        k.hex = hex128(h1, h2);
        if (!path.empty()) {
            mkdirs(dir);
            const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
            if (FILE *f = fopen(tmp.c_str(), "wb")) {
                const bool ok = fwrite(k.hex.data(), 1, 32, f) == 32;
                if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
            }
        }
    }
    std::lock_guard<std::mutex> lk(g_name_mutex);
    if (g_names.size() > 256) g_names.clear();
    g_names[name] = k.hex;
    return k;
}

// ---- out-of-process builds ---------------------------------------------------------------------------------------
// hiprtc serialises compiles inside a process (two threads: 7.9 s either way for chess, measured), and an LLVM abort
// inside it takes the process down.  So a program's two modules are built by two helper processes side by side
// (maray_jitc, next to this library; it dlopens the very hiprtc this process has loaded -- the compiler's version is
// part of the code key): chess cold 2.6 -> 1.6 s.  MARAY_JIT_HELPER=0, or a helper that is missing or cannot reach the
// compiler: the module is compiled in-process.  A source that does not compile is an error either way, with the
// compiler's log; a helper that dies while compiling is an error too (MARAY_E_HIP; BACKEND_AUTO then takes the interpreter).
std::string self_dir()
{
    Dl_info info;
    if (!dladdr((const void *)&maray_build_id, &info) || !info.dli_fname) return "";
    const std::string p = info.dli_fname;
    const size_t at = p.rfind('/');
    return at == std::string::npos ? "." : p.substr(0, at);
}

std::string hiprtc_path()
{
    Dl_info info;
    if (!dladdr((const void *)&hiprtcCompileProgram, &info) || !info.dli_fname) return "";
    return info.dli_fname;
}

struct HelperJob {
    pid_t pid = -1;
    std::string dir, src_path, out_path;
};

bool read_file(const std::string &path, std::vector<char> &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    out.clear();
    for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) out.insert(out.end(), buf, buf + n);
    fclose(f);
    return true;
}

// starts the helper on `src`; pid stays -1 when it cannot be started.  Source and result live in a directory of their own
// (mkdtemp, 0700): nobody else can put a file or a link where the helper writes and this process reads.
HelperJob helper_start(const std::string &helper, const std::string &rtc, const std::string &src, const char *tag)
{
    HelperJob j;
    const char *tmp = getenv("TMPDIR");
    char path[512];
    snprintf(path, sizeof path, "%s/maray_jit_%ld_%s_XXXXXX", (tmp && tmp[0]) ? tmp : "/tmp", (long)getpid(), tag);
    if (!mkdtemp(path)) return j;
    j.dir = path;
    j.src_path = j.dir + "/kernel.hip";
    j.out_path = j.dir + "/kernel.out";
    const int fd = open(j.src_path.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
    if (fd < 0) return j;
    const bool ok = write(fd, src.data(), src.size()) == (ssize_t)src.size();
    close(fd);
    if (!ok) return j;
    const char *olevel = getenv("MARAY_JIT_OPT");
    std::vector<char *> argv = {(char *)helper.c_str(), (char *)rtc.c_str(), (char *)j.src_path.c_str(), (char *)j.out_path.c_str()};
    if (olevel && olevel[0] == '-') argv.push_back((char *)olevel);
    argv.push_back(nullptr);
    pid_t pid = -1;
    if (posix_spawn(&pid, helper.c_str(), nullptr, nullptr, argv.data(), environ) == 0) j.pid = pid;
    return j;
}

// What became of a helper.  OK: `code` holds the code object.  REJECTED: the source does not compile, `log` holds the
// compiler's errors.  ABSENT: the helper never got as far as the compiler (not started, could not load hiprtc or read its
// input): the caller compiles in-process.  DIED: the helper was running the compiler and ended by a signal or an
// unexpected status -- an abort inside LLVM is what the helper exists to keep out of the caller, so this is an error,
// never a reason to run the same compile in-process; `log` says how it ended and where its source was kept.
enum HelperEnd { HELPER_OK, HELPER_REJECTED, HELPER_ABSENT, HELPER_DIED };

HelperEnd helper_finish(HelperJob &j, std::vector<char> &code, std::string &log)
{
    HelperEnd end = HELPER_ABSENT;
    if (j.pid > 0) {
        int status = 0;
        pid_t r;
        do r = waitpid(j.pid, &status, 0); while (r < 0 && errno == EINTR);
        std::vector<char> out;
        if (r != j.pid) { end = HELPER_DIED; log = "waitpid failed"; }
        else if (WIFSIGNALED(status)) { end = HELPER_DIED; log = "signal " + std::to_string(WTERMSIG(status)) + (WTERMSIG(status) == SIGABRT ? " (abort)" : ""); }
        else if (!WIFEXITED(status)) { end = HELPER_DIED; log = "wait status " + std::to_string(status); }
        else switch (WEXITSTATUS(status)) {
        case 0:
            if (read_file(j.out_path, out) && out.size() >= 64 && memcmp(out.data(), "\177ELF", 4) == 0) { code.swap(out); end = HELPER_OK; }
            else { end = HELPER_DIED; log = "exit status 0 without a code object"; }
            break;
        case 3:
            if (read_file(j.out_path, out)) { log.assign(out.begin(), out.end()); end = HELPER_REJECTED; }
            else { end = HELPER_DIED; log = "exit status 3 without a compiler log"; }
            break;
        case 2: case 4: case 5: case 127: end = HELPER_ABSENT; break;      // usage / no hiprtc / no input / not executable: the compiler never ran
        default: end = HELPER_DIED; log = "exit status " + std::to_string(WEXITSTATUS(status));
        }
    }
    if (end == HELPER_DIED && !j.src_path.empty()) log += "; source kept in " + j.src_path;      // for the bug report
    else if (!j.src_path.empty()) (void)unlink(j.src_path.c_str());
    if (!j.out_path.empty()) (void)unlink(j.out_path.c_str());
    if (!j.dir.empty() && end != HELPER_DIED) (void)rmdir(j.dir.c_str());
    return end;
}

std::shared_ptr<const JitCode> build_code(const maray_program &prog, CodeKey &k)
{
    auto c = std::make_shared<JitCode>();
    const std::string dir = cache_dir(), path = dir.empty() ? "" : dir + "/" + k.hex + ".mrco";
    if (!path.empty() && cache_read(path, *c)) { c->from_disk = true; return c; }
    key_sources(prog, k);          // (a key that came from its cheaper name has no sources yet)
    std::string log;
    // both modules in helper processes, side by side
    bool have_pix = false, have_rows = false;
    {
        const char *e = getenv("MARAY_JIT_HELPER");
        const std::string helper = self_dir() + "/maray_jitc", rtc = hiprtc_path();
        if (!(e && e[0] == '0') && !rtc.empty() && access(helper.c_str(), X_OK) == 0) {
            HelperJob jp = helper_start(helper, rtc, k.src_pix, "pix"), jr;
            if (prog.n_row_ops) jr = helper_start(helper, rtc, k.src_rows, "rows");
            std::string lp, lr;
            const HelperEnd rp = helper_finish(jp, c->pix, lp), rr = prog.n_row_ops ? helper_finish(jr, c->rows, lr) : HELPER_ABSENT;
            if (rp == HELPER_REJECTED) throw Error{MARAY_E_HIP, "hiprtcCompileProgram (maray_jitc): the PIXEL kernel does not compile\n" + lp};
            if (rr == HELPER_REJECTED) throw Error{MARAY_E_HIP, "hiprtcCompileProgram (maray_jitc): the ROW kernel does not compile\n" + lr};
            if (rp == HELPER_DIED) throw Error{MARAY_E_HIP, "the compiler aborted on the PIXEL kernel (maray_jitc: " + lp + ")"};
            if (rr == HELPER_DIED) throw Error{MARAY_E_HIP, "the compiler aborted on the ROW kernel (maray_jitc: " + lr + ")"};
            have_pix = rp == HELPER_OK; have_rows = rr == HELPER_OK;
        }
    }
    // Occupancy: the generator's own choice (8 or 6 waves per SIMD, jit_source), then 6 / 4 / 2 (<= 80 / 128 / 256 VGPRs)
    // until a build needs no scratch: spilled VGPRs are HBM traffic.
    const int ladder[] = {0, 6, 4, 2};
    for (int i = 0; i < 4; i++) {
        if (!(i == 0 && have_pix)) jit_compile(i == 0 ? k.src_pix : jit_source(prog, ladder[i]), c->pix, log);
        c->waves = ladder[i];
        if (code_meta_uint(c->pix, ".private_segment_fixed_size") <= 0) break;
    }
    if (prog.n_row_ops && !have_rows) jit_compile(k.src_rows, c->rows, log);
    c->n_row_chunks = k.n_row_chunks; c->n_gjobs = k.n_gjobs;
    if (prog.n_row_ops) {          // (the guard plan is a walk over the ROW tape: once here, not in every context's creation)
        const GuardGeom geom = jit_guard_geom(prog);
        c->n_gwords = jit_guard_words(prog); c->guard_w = geom.gw; c->guard_h = geom.gh;
    }
    if (!path.empty()) cache_write(dir, path, *c);
    return c;
}

}   // namespace

std::string jit_code_key(const maray_program &prog) { return code_key(prog).hex; }

bool jit_code_is_cached(const maray_program &prog)
{
    const CodeKey k = code_key(prog);
    {
        std::lock_guard<std::mutex> lk(g_code_mutex);
        if (g_code.count(k.hex)) return true;
    }
    const std::string dir = cache_dir();
    return !dir.empty() && access((dir + "/" + k.hex + ".mrco").c_str(), R_OK) == 0;
}

std::shared_ptr<const JitCode> jit_code_for(const maray_program &prog)
{
    CodeKey k = code_key(prog);
    std::promise<std::shared_ptr<const JitCode>> mine;
    std::shared_future<std::shared_ptr<const JitCode>> fut;
    bool build = false;
    {
        std::lock_guard<std::mutex> lk(g_code_mutex);
        auto it = g_code.find(k.hex);
        if (it != g_code.end()) fut = it->second;
        else {
            fut = mine.get_future().share(); g_code.emplace(k.hex, fut); build = true;
            g_code_order.push_back(k.hex);
            while (g_code_order.size() > 32) {          // contexts hold their code objects themselves (shared_ptr)
                g_code.erase(g_code_order.front());
                g_code_order.erase(g_code_order.begin());
            }
        }
    }
    if (build) {
        try { mine.set_value(build_code(prog, k)); }
        catch (...) {
            mine.set_exception(std::current_exception());
            std::lock_guard<std::mutex> lk(g_code_mutex);
            g_code.erase(k.hex);                       // a failed build is not remembered (the waiters still see its error)
        }
    }
    return fut.get();         // the contexts of a multi-GPU render: the first builds, the others wait here
}

namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            throw Error{MARAY_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)};           \
    } while (0)

struct DevTex { const unsigned char *rgb; unsigned w, h; };

struct JitBackend final : Backend {
    int device = 0;
    maray_program P{};
    std::shared_ptr<const JitCode> code;
    hipModule_t mod = nullptr, mod_rows = nullptr;
    hipFunction_t f_rows = nullptr, f_pix = nullptr, f_order = nullptr;
    unsigned *d_order = nullptr; size_t order_cap = 0;
    uint64_t order_key[3] = {0, 0, 0};                     // the geometry d_order was computed for
    uint64_t seen_key[3] = {0, 0, 0};                      // the geometry of the previous launch
    Backend *slow = nullptr;            // tape interpreter: evaluates the tiles the pixel kernel deferred
    unsigned *d_flags = nullptr; size_t flags_cap = 0;
    DevTex *d_tex = nullptr;
    std::vector<unsigned char *> d_tex_rgb;
    double *d_yvals = nullptr; size_t yvals_cap = 0;          // ROW-stage tables (the context's, not a launch's: launches are ordered by their stream)
    unsigned long long *d_gbits = nullptr; size_t gbits_cap = 0;
    unsigned char *d_rgb8 = nullptr; size_t rgb8_cap = 0;      // time_rows without a caller's buffer
    HostPipe pipe;                      // streams + staging of the host-raster entry points
    hipStream_t own_stream = nullptr;   // = pipe's compute stream
    uint32_t n_row_chunks = 1, n_gwords = 0, n_gjobs = 0, guard_rows = 1, guard_sub = 1;       // guard_sub: guard rectangles per 256-pixel tile
    uint32_t n_cu = 256;
    bool wide_all = false;              // the whole section runs four pixels per lane (jit_wide_general)
    unsigned k_tiles = 0;               // MARAY_JIT_TILES: tiles per wavefront (0 = by launch size), read once when the context is created
    bool has_sin = false;               // some Sin argument is not proven bounded: tiles may be deferred to `slow`
    hipStream_t last_stream = nullptr; bool have_last = false;      // the stream of the last launch (see launch())
    hipEvent_t handover = nullptr;

    ~JitBackend() override {
        (void)hipSetDevice(device);
        delete slow;
        if (mod) (void)hipModuleUnload(mod);
        if (mod_rows) (void)hipModuleUnload(mod_rows);
        (void)hipFree(d_flags);
        (void)hipFree(d_tex);
        for (auto p : d_tex_rgb) (void)hipFree(p);
        (void)hipFree(d_order); (void)hipFree(d_rgb8);
        (void)hipFree(d_yvals); (void)hipFree(d_gbits);
        if (handover) (void)hipEventDestroy(handover);
    }

    void init(int dev, const maray_program &prog, const maray_texture *tex, uint32_t n_tex) {
        device = dev;
        // MARAY_TRACE_INIT=1: where the time of a context's creation goes (stderr)
        const bool trace = getenv("MARAY_TRACE_INIT") && getenv("MARAY_TRACE_INIT")[0] == '1';
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!trace) return;
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "maray init: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
            t_last = now;
        };
        HIP_TRY(hipSetDevice(dev));
        // (hipGetDeviceProperties costs ~5 ms a call: asked once per device and process)
        static std::mutex prop_mutex;
        static std::map<int, std::pair<std::string, int>> props;
        std::pair<std::string, int> info;
        {
            std::lock_guard<std::mutex> lk(prop_mutex);
            auto it = props.find(dev);
            if (it == props.end()) {
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDeviceProperties(&prop, dev));
                it = props.emplace(dev, std::make_pair(std::string(prop.gcnArchName), prop.multiProcessorCount)).first;
            }
            info = it->second;
        }
        if (info.first.rfind("gfx950", 0) != 0)
            throw Error{MARAY_E_NO_DEVICE, "device is " + info.first + ", this library is built for gfx950 only"};
        n_cu = (uint32_t)std::max(1, info.second);
        has_sin = may_defer_tiles(prog);
        // (set below, from the code object's header: the guard plan is not recomputed per context)
        if (const char *e_ = getenv("MARAY_JIT_TILES")) if (atoi(e_) > 0) k_tiles = (unsigned)atoi(e_);
        lap("device");
        code = jit_code_for(prog);                       // built by the first context of the process, or read from the cache
        lap("code objects");
        HIP_TRY(hipModuleLoadData(&mod, code->pix.data()));
        HIP_TRY(hipModuleGetFunction(&f_pix, mod, "maray_jit_pixels"));
        lap("load PIXEL module");
        n_row_chunks = code->n_row_chunks; n_gjobs = code->n_gjobs;
        wide_all = jit_wide_general(prog, code->n_gwords);
        if (has_sin) slow = make_tape_backend(dev, prog, tex, n_tex, false);     // drains the tiles the pixel kernel defers; other programs never defer
        P = prog;
        P.consts = nullptr; P.row_ops = nullptr; P.pix_ops = nullptr;
        if (prog.n_row_ops) {
            HIP_TRY(hipModuleLoadData(&mod_rows, code->rows.data()));
            HIP_TRY(hipModuleGetFunction(&f_rows, mod_rows, "maray_jit_rows"));
            n_gwords = code->n_gwords;
            if (n_gwords) HIP_TRY(hipModuleGetFunction(&f_order, mod_rows, "maray_jit_order"));
            guard_rows = code->guard_h;
            guard_sub = 256u / code->guard_w;
            lap("load ROW module");

        }
        pipe.init(dev);
        lap("host pipe");
        own_stream = pipe.compute_stream();
        HIP_TRY(hipEventCreateWithFlags(&handover, hipEventDisableTiming));
        std::vector<DevTex> descs(n_tex ? n_tex : 1);
        for (uint32_t i = 0; i < n_tex; i++) {
            unsigned char *d = nullptr;
            const size_t bytes = (size_t)tex[i].w * tex[i].h * 3;
            HIP_TRY(hipMalloc((void **)&d, bytes ? bytes : 8));
            if (bytes) HIP_TRY(hipMemcpy(d, tex[i].rgb, bytes, hipMemcpyHostToDevice));
            d_tex_rgb.push_back(d);
            descs[i] = DevTex{d, tex[i].w, tex[i].h};
        }
        HIP_TRY(hipMalloc((void **)&d_tex, descs.size() * sizeof(DevTex)));
        HIP_TRY(hipMemcpy(d_tex, descs.data(), descs.size() * sizeof(DevTex), hipMemcpyHostToDevice));
        lap("streams, events, textures");
    }

    template <typename T>
    void ensure(T *&p, size_t &cap, size_t n) {
        if (n <= cap) return;
        if (p) HIP_TRY(hipFree(p));
        p = nullptr; cap = 0;
        HIP_TRY(hipMalloc((void **)&p, n * sizeof(T)));
        cap = n;
    }

    void launch(uint32_t w, const RowBlocks &rb, unsigned char *d8, double *d64, hipStream_t st, bool rows_pass) {
        const uint32_t rows_total = rb.n_rows, y0 = rb.y0;
        unsigned blk_rows = rb.block_rows, blk_stride = rb.block_stride;
        if (!rows_total || !w) return;
        // The scratch tables (y values, guard bits, row order, work list) belong to the context, not to a launch: launches
        // are ordered by the stream they are issued on.  When a launch arrives on another stream than the last one, that
        // stream is made to wait for everything the previous one holds (one event, only at the hand-over).
        if (have_last && st != last_stream) {
            HIP_TRY(hipEventRecord(handover, last_stream));
            HIP_TRY(hipStreamWaitEvent(st, handover, 0));
        }
        last_stream = st; have_last = true;
        // rows per guard evaluation: a group must not straddle two row blocks (its image rows have to be consecutive)
        unsigned yrows = (guard_rows > 1 && (blk_rows >= rows_total || blk_rows % guard_rows == 0)) ? guard_rows : 1u;
        const uint32_t n_groups = (rows_total + yrows - 1) / yrows;
        // (the ROW stage on a stream of its own under the previous launch's PIXEL kernel was measured and lost: the two kernels
        // share the SIMDs and the events that order the streams cost a signal round trip each)
        if (rows_pass) {
            ensure(d_yvals, yvals_cap, (size_t)rows_total * std::max<uint32_t>(P.n_yvals, 1));
            const size_t had = gbits_cap;
            ensure(d_gbits, gbits_cap, (size_t)n_groups * ((w + 255) / 256) * guard_sub * std::max<uint32_t>(n_gwords, 1));
            // bits past the last guard belong to no job and are never written: zero them once (the pixel kernel tests whole words)
            if (gbits_cap != had) HIP_TRY(hipMemsetAsync(d_gbits, 0, gbits_cap * sizeof(unsigned long long), st));
        }
        if (!d_yvals || !d_gbits) throw Error{MARAY_E_INTERNAL, "launch without a ROW pass before any ROW pass"};
        unsigned n_yvals = P.n_yvals;
        if (rows_pass && P.n_row_ops) {
            unsigned yy0 = y0, rr = rows_total, ww = w;
            unsigned n_tx_ = (w + 255) / 256 * guard_sub;
            // guards: one item per rectangle (group of yrows rows, run of 256 / guard_sub pixels); y values: one per row
            const uint64_t items = std::max<uint64_t>(n_gwords ? (uint64_t)n_groups * n_tx_ : 0, rows_total);
            const unsigned bs = ROW_BLOCK;
            if (items + bs > 0xFFFFFFFFull) throw Error{MARAY_E_ARG, "too many tiles in one launch; render fewer rows per call"};
            void *args[] = {&d_yvals, &d_gbits, &d_tex, &yy0, &rr, &n_yvals, &ww, &n_tx_, &blk_rows, &blk_stride, &yrows};
            unsigned gy = n_row_chunks + n_gjobs;
            HIP_TRY(hipModuleLaunchKernel(f_rows, (unsigned)((items + bs - 1) / bs), gy, 1, bs, 1, 1, 0, st, args, nullptr));
        }
        // launch order of the PIXEL kernel (maray_jit_order): once per geometry, from the guard bits the ROW kernel just wrote;
        // a cached order is used by every launch of that geometry, with or without a ROW pass (time_rows)
        const unsigned *row_order = nullptr;
        if (f_order && n_gwords && rows_total <= 65535 && n_groups <= 4096 && n_groups > 1) {
            const uint64_t key[3] = {((uint64_t)w << 32) | rows_total, ((uint64_t)y0 << 32) | blk_rows, ((uint64_t)blk_stride << 32) | yrows};
            const bool cached = key[0] == order_key[0] && key[1] == order_key[1] && key[2] == order_key[2];
            // The order costs one 43 us kernel per geometry and saves ~2.6 us per launch: it is computed when a geometry
            // comes a second time (an animation, a benchmark loop), never for the tiles of a one-shot render, each of
            // which is a geometry of its own.
            const bool again = key[0] == seen_key[0] && key[1] == seen_key[1] && key[2] == seen_key[2];
            seen_key[0] = key[0]; seen_key[1] = key[1]; seen_key[2] = key[2];
            if (!cached && rows_pass && again) {
                order_key[0] = order_key[1] = order_key[2] = 0;     // no geometry owns d_order until the kernel below is enqueued
                ensure(d_order, order_cap, (size_t)rows_total);
                unsigned rr = rows_total, n_tx_ = (w + 255) / 256 * guard_sub;
                void *oargs[] = {&d_gbits, &d_order, &rr, &n_tx_, &yrows};
                HIP_TRY(hipModuleLaunchKernel(f_order, 1, 1, 1, 256, 1, 1, n_groups * 4, st, oargs, nullptr));
                order_key[0] = key[0]; order_key[1] = key[1]; order_key[2] = key[2];
            }
            if (key[0] == order_key[0] && key[1] == order_key[1] && key[2] == order_key[2]) row_order = d_order;
        }
        const unsigned n_tx = (w + 255) / 256;
        const uint64_t n_tiles = (uint64_t)n_tx * rows_total;
        // Tiles per wavefront.  Every tile of a program without guards costs the same: long strips, nothing to balance (config 2
        // at 8192^2: 2.74 -> 2.88 TB/s with 8 tiles; only for the cheap four-wide sections, a narrow wavefront would live too
        // long).  Else rows cost what they show (sky: nothing, board: 5x the average) and wavefronts are the unit of load
        // balance: a launch that fills the device only a few times over -- chess up to 2048^2 -- is a matter of how long its
        // longest wavefront lives, not of throughput: one tile per wavefront (1024^2 / 2048^2 / 4096^2 / 8192^2 with 1 tile:
        // 20.3 / 18.9 / 48.4 / 149 us per step, with 2: 25.5 / 24.8 / 42.3 / 138); a launch that fills it dozens of times over
        // is a matter of what every wavefront costs before its first pixel, its tail is short against the whole: four tiles
        // (8192^2 / 16384^2 with 2 / 3 / 4 / 5 tiles: 138 / 123 / 119 / 139 and 517 / 487 / 497 / 566 us per step; 4096^2:
        // 42.2 / 55.5 / 44.6 / 46.9)
        const uint64_t device_slots = (uint64_t)n_cu * 4 * 7;
        unsigned tiles = wide_all ? 8 : (n_tiles <= 4 * device_slots ? 1 : n_tiles <= 16 * device_slots ? 2 : 4);
        if (k_tiles) tiles = std::min(64u, k_tiles);
        if (n_gwords && n_gwords <= GW_INLINE_MAX) tiles = std::min(tiles, 64u / (n_gwords * guard_sub));      // a strip's guard words: one per lane
        tiles = std::max(1u, std::min(tiles, n_tx));
        const unsigned gx = (n_tx + 4 * tiles - 1) / (4 * tiles);
        if (n_tiles > 0xFFFFFFFFull) throw Error{MARAY_E_ARG, "too many tiles in one launch; render fewer rows per call"};
        if (has_sin) {
            ensure(d_flags, flags_cap, (size_t)n_tiles + 1);                // work list {count, tile, ...} of deferred tiles
            HIP_TRY(hipMemsetAsync(d_flags, 0, sizeof(unsigned), st));
        }
        for (uint32_t r0 = 0; r0 < rows_total; r0 += 65535) {          // gridDim.y limit
            const uint32_t rows = std::min<uint32_t>(65535, rows_total - r0);
            unsigned char *p8 = d8 ? d8 + (size_t)r0 * w * 3 : nullptr;
            double *p64 = d64 ? d64 + (size_t)r0 * w * 3 : nullptr;
            const double *yv = d_yvals + (size_t)r0 * n_yvals;
            unsigned *fl = d_flags;
            unsigned ww = w, yy0 = y0, tile_base = r0 * n_tx, row_base = r0;
            const unsigned long long *gb = d_gbits;              // indexed by the row of the whole call
            unsigned ntx = n_tx;
            void *args[] = {&p8, &p64, &yv, &d_tex, &fl, &tile_base, &gb, &ntx, &ww, &yy0, &n_yvals, &tiles, &blk_rows, &blk_stride, &row_base, &yrows, &row_order};
            HIP_TRY(hipModuleLaunchKernel(f_pix, gx, rows, 1, 256, 1, 1, 0, st, args, nullptr));
        }
        if (has_sin) slow->render_flagged(w, rb, d8, d64, st, d_flags, d_yvals);   // no-op unless a tile was deferred
    }

    void render_device(uint32_t w, uint32_t, const RowBlocks &rb, void *d8, void *d64, void *stream) override {
        HIP_TRY(hipSetDevice(device));
        launch(w, rb, (unsigned char *)d8, (double *)d64, (hipStream_t)stream, true);
    }

    void render_host_tiles(uint32_t w, uint32_t, const std::vector<RowTile> &tiles, uint32_t row0, uint8_t *rgb8, double *rgb64,
                           const std::function<void(uint32_t, uint32_t)> &done) override {
        HIP_TRY(hipSetDevice(device));
        pipe.run(w, tiles, row0, rgb8, rgb64,
                 [&](const RowBlocks &rb, unsigned char *d8, double *d64, hipStream_t st) { launch(w, rb, d8, d64, st, true); }, done);
    }

    float time_rows(uint32_t w, uint32_t, const RowBlocks &rb, void *d8, void *d64, int reps) override {
        HIP_TRY(hipSetDevice(device));
        const size_t n = (size_t)rb.n_rows * w * 3;
        unsigned char *p8 = (unsigned char *)d8;
        double *p64 = (double *)d64;
        if (!p8 && !p64) { ensure(d_rgb8, rgb8_cap, n); p8 = d_rgb8; }
        launch(w, rb, p8, p64, own_stream, true);
        launch(w, rb, p8, p64, own_stream, true);          // a geometry's second launch computes its row order
        for (int i = 0; i < 5; i++) launch(w, rb, p8, p64, own_stream, false);      // (the timed launches follow launches of their own kind)
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, own_stream));
        for (int i = 0; i < reps; i++) launch(w, rb, p8, p64, own_stream, false);
        HIP_TRY(hipEventRecord(e1, own_stream));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        return ms / (float)(reps > 0 ? reps : 1);
    }

    const char *kernel_name() const override { return "maray_jit_pixels"; }
};

}   // namespace

Backend *make_jit_backend(int device, const maray_program &prog, const maray_texture *tex, uint32_t n_tex)
{
    if (hip_device_count() <= 0) throw Error{MARAY_E_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)"};
    auto *b = new JitBackend();
    try {
        b->init(device, prog, tex, n_tex);
    } catch (...) {
        delete b;
        throw;
    }
    return b;
}

}   // namespace maray
