// row_split.cpp — analysis of the ROW section shared by the evaluators (product code).
//
// The ROW section's outputs are independent: y values that PIXEL ops read as operands, and guards
// that only gate SKIP ops.  Both back-ends cut the tape by outputs (the cone of a set of OUT ops)
// to run parts of it side by side or at a different granularity (guards per rectangle of pixels).
#include <algorithm>
#include <cstdint>
#include <vector>

#include "backend.hpp"
#include "maray_tape.h"

namespace maray {

// Number of leading y values that PIXEL ops read as arithmetic operands; the rest of the table only gates SKIP ops.
uint32_t numeric_yvals(const maray_program &P)
{
    uint32_t n = 0;
    for (uint32_t i = 0; i < P.n_pix_ops; i++) {
        const uint64_t ins = P.pix_ops[i];
        const uint32_t op = MARAY_INS_OP(ins);
        if (op == MARAY_OP_NOP || op == MARAY_OP_SKIPZ || op == MARAY_OP_SKIPNZ || op == MARAY_OP_TEXDIM) continue;
        const uint32_t refs[2] = {MARAY_INS_A(ins), (op >= MARAY_OP_ADD && op <= MARAY_OP_APP) ? MARAY_INS_B(ins) : 0u};
        for (uint32_t r : refs) if (MARAY_REF_KIND(r) == MARAY_K_YVAL) n = std::max(n, MARAY_REF_INDEX(r) + 1);
    }
    return n;
}

RowTapeDeps row_tape_deps(const maray_program &P)
{
    const uint32_t n = P.n_row_ops;
    RowTapeDeps d;
    d.deps.assign(n, {-1, -1});
    d.reads_y.assign(n, 0);
    auto is_y = [](uint32_t ref) { return MARAY_REF_KIND(ref) == MARAY_K_SPEC && MARAY_REF_INDEX(ref) == MARAY_SPEC_Y; };
    std::vector<int32_t> slot_writer(P.n_row_slots ? P.n_row_slots : 1, -1);
    int32_t acc = -1;
    auto producer = [&](uint32_t ref) -> int32_t {
        const uint32_t kind = MARAY_REF_KIND(ref), idx = MARAY_REF_INDEX(ref);
        if (kind == MARAY_K_SLOT) return slot_writer[idx];
        if (kind == MARAY_K_SPEC && idx == MARAY_SPEC_ACC) return acc;
        return -1;
    };
    for (uint32_t j = 0; j < n; j++) {
        const uint64_t ins = P.row_ops[j];
        const uint32_t op = MARAY_INS_OP(ins);
        if (op == MARAY_OP_NOP) continue;
        if (op != MARAY_OP_TEXDIM) {
            d.deps[j][0] = producer(MARAY_INS_A(ins));
            d.reads_y[j] = is_y(MARAY_INS_A(ins)) || (d.deps[j][0] >= 0 && d.reads_y[d.deps[j][0]]);
        }
        if (op == MARAY_OP_SKIPZ || op == MARAY_OP_SKIPNZ) continue;        // reads its guard, leaves ACC and the slots alone
        if (op == MARAY_OP_OUT) { d.outs.push_back(j); continue; }
        if (op >= MARAY_OP_ADD && op <= MARAY_OP_APP) {
            d.deps[j][1] = producer(MARAY_INS_B(ins));
            d.reads_y[j] |= is_y(MARAY_INS_B(ins)) || (d.deps[j][1] >= 0 && d.reads_y[d.deps[j][1]]);
        }
        acc = (int32_t)j;
        if (MARAY_INS_DST(ins) != MARAY_DST_NONE) slot_writer[MARAY_INS_DST(ins)] = (int32_t)j;
    }
    return d;
}

// The ROW tape with everything but the cone of the given OUT ops turned into NOPs.  A SKIP op stays
// when the op that ends its region does (and then its guard is needed too): what a region holds
// feeds nothing outside it, so any part of it can still be skipped as a whole.
std::vector<uint64_t> row_tape_cone(const maray_program &P, const RowTapeDeps &d, const std::vector<uint32_t> &outs, size_t *cost)
{
    const uint32_t n = P.n_row_ops;
    std::vector<uint8_t> need(n, 0);
    for (uint32_t o : outs) need[o] = 1;
    size_t c = 0;
    for (uint32_t j = n; j-- > 0;) {             // producers precede consumers; a region's end follows its SKIP op
        const uint32_t op = MARAY_INS_OP(P.row_ops[j]);
        if (op == MARAY_OP_SKIPZ || op == MARAY_OP_SKIPNZ) {
            const uint32_t end = j + MARAY_INS_AUX(P.row_ops[j]);
            if (end < n && need[end]) need[j] = 1;
        }
        if (!need[j]) continue;
        c++;
        for (int32_t p : d.deps[j]) if (p >= 0) need[p] = 1;
    }
    std::vector<uint64_t> tape(n, 0);           // 0 = NOP
    for (uint32_t j = 0; j < n; j++) if (need[j]) tape[j] = P.row_ops[j];
    if (cost) *cost = c;
    return tape;
}

std::vector<uint64_t> compact_tape(const std::vector<uint64_t> &tape)
{
    const uint32_t n = (uint32_t)tape.size();
    std::vector<uint32_t> pos(n + 1, 0);            // new index of op j (for a NOP: of the next op kept)
    for (uint32_t j = 0; j < n; j++) pos[j + 1] = pos[j] + (MARAY_INS_OP(tape[j]) != MARAY_OP_NOP);
    std::vector<uint64_t> out;
    out.reserve(pos[n]);
    for (uint32_t j = 0; j < n; j++) {
        uint64_t ins = tape[j];
        const uint32_t op = MARAY_INS_OP(ins);
        if (op == MARAY_OP_NOP) continue;
        if (op == MARAY_OP_SKIPZ || op == MARAY_OP_SKIPNZ) {
            const uint32_t end = j + MARAY_INS_AUX(ins);            // kept by construction (row_tape_cone)
            ins = MARAY_INS(op, pos[end] - pos[j], MARAY_INS_DST(ins), MARAY_INS_A(ins), MARAY_INS_B(ins));
        }
        out.push_back(ins);
    }
    return out;
}

// Renumbers the value slots of a (compacted) tape by liveness: a slot is free again after the last op that reads the
// value it holds.  The cone of a few outputs keeps the slot numbers the whole section was allocated with (hundreds, most
// of them idle); an interpreter that keeps slots in LDS wants the few that are live at once.  Returns the slots used.
//
// A value = one definition.  A SKIP op with a dst defines, on the taken path, the value that the op ending its region
// defines on the other: the two (and every SKIP op of a region nested at the same end) are ONE value, and its slot stays
// taken from the outermost of those SKIP ops to the end op at least.  Liveness is taken over the linear order of the
// tape, which is conservative for either path through a region (nothing defined inside one is read after it).
uint32_t renumber_slots(std::vector<uint64_t> &tape)
{
    const uint32_t n = (uint32_t)tape.size();
    std::vector<int32_t> cur(MARAY_DST_NONE + 1, -1);       // original slot -> the value it holds now
    std::vector<int32_t> value_of(n, -1);                   // op -> the value it defines (index of the first op that defines it)
    std::vector<int32_t> last_use(n, -1);                   // value -> last op that needs its slot
    std::vector<int32_t> use_a(n, -1), use_b(n, -1);        // op -> the value its slot operand reads
    std::vector<int32_t> region_value(n, -1);               // end op -> value reserved by the SKIP op(s) of its region
    auto read = [&](uint32_t ref, uint32_t j, std::vector<int32_t> &use) {
        if (MARAY_REF_KIND(ref) != MARAY_K_SLOT) return;
        const int32_t v = cur[MARAY_REF_INDEX(ref)];
        use[j] = v;
        if (v >= 0) last_use[v] = (int32_t)j;
    };
    for (uint32_t j = 0; j < n; j++) {
        const uint64_t ins = tape[j];
        const uint32_t op = MARAY_INS_OP(ins), dst = MARAY_INS_DST(ins);
        if (op == MARAY_OP_NOP) continue;
        if (op != MARAY_OP_TEXDIM) read(MARAY_INS_A(ins), j, use_a);
        if (op >= MARAY_OP_ADD && op <= MARAY_OP_APP) read(MARAY_INS_B(ins), j, use_b);
        if (op == MARAY_OP_OUT || dst == MARAY_DST_NONE) continue;
        if (op == MARAY_OP_SKIPZ || op == MARAY_OP_SKIPNZ) {
            const uint32_t end = j + MARAY_INS_AUX(ins);
            if (end >= n) continue;                          // (validated tapes never do this)
            if (region_value[end] < 0) region_value[end] = (int32_t)j;
            value_of[j] = region_value[end];
            last_use[value_of[j]] = std::max(last_use[value_of[j]], (int32_t)end);
        } else
            value_of[j] = region_value[j] >= 0 ? region_value[j] : (int32_t)j;
        cur[dst] = value_of[j];
    }
    std::vector<std::vector<int32_t>> dies(n);              // op -> values whose slot is free after it
    for (uint32_t v = 0; v < n; v++) if (value_of[v] == (int32_t)v && last_use[v] >= 0) dies[last_use[v]].push_back((int32_t)v);
    std::vector<int32_t> slot_of(n, -1);
    std::vector<uint32_t> free_list;
    uint32_t used = 0;
    for (uint32_t j = 0; j < n; j++) {
        const uint64_t ins = tape[j];
        const uint32_t op = MARAY_INS_OP(ins);
        if (op == MARAY_OP_NOP) continue;
        uint32_t a = MARAY_INS_A(ins), b = MARAY_INS_B(ins), dst = MARAY_INS_DST(ins);
        if (use_a[j] >= 0) a = MARAY_REF(MARAY_K_SLOT, (uint32_t)slot_of[use_a[j]]);
        if (use_b[j] >= 0) b = MARAY_REF(MARAY_K_SLOT, (uint32_t)slot_of[use_b[j]]);
        // operands are read before the result is written: a value that dies here may hand its slot to this op's result,
        // unless it is this op's own value (the end of a region whose result nothing reads afterwards)
        for (int32_t v : dies[j]) if (v != value_of[j]) free_list.push_back((uint32_t)slot_of[v]);
        if (value_of[j] >= 0) {
            const int32_t v = value_of[j];
            if (v == (int32_t)j) {                           // a new value
                if (last_use[v] < 0) dst = MARAY_DST_NONE;   // only ever consumed through ACC here
                else {
                    if (free_list.empty()) free_list.push_back(used++);
                    slot_of[v] = (int32_t)free_list.back();
                    free_list.pop_back();
                    dst = (uint32_t)slot_of[v];
                }
            } else dst = (uint32_t)slot_of[v];               // continues the value its region's SKIP op reserved
        }
        for (int32_t v : dies[j]) if (v == value_of[j]) free_list.push_back((uint32_t)slot_of[v]);
        tape[j] = MARAY_INS(op, MARAY_INS_AUX(ins), dst, a, b);
    }
    return used;
}

// Re-orders a (compacted) cone of the ROW tape and allocates its value slots afresh.  Returns the slots used.
//
// A cone keeps the order the whole section was scheduled in: values that other cones wanted first sit at the front and
// stay live until this cone gets round to them -- 150 slots live at once for a guard whose expression tree needs a
// dozen.  The interpreter keeps slots in LDS (2 KB each per block), so the live set IS its occupancy.  Here the cone is
// scheduled on its own: units (an op, or a SKIP region with everything up to the op that ends it, kept verbatim) in
// depth-first order from each OUT, the operand that needs more slots first (Sethi-Ullman), results handed over through
// ACC when producer and consumer end up adjacent, slots by liveness over the new order.  (Kept only if it needs fewer
// slots than the original order does.)
uint32_t reschedule_tape(std::vector<uint64_t> &tape)
{
    const uint32_t n = (uint32_t)tape.size();
    if (!n) return 0;
    auto is_skip = [&](uint32_t j) { const uint32_t op = MARAY_INS_OP(tape[j]); return op == MARAY_OP_SKIPZ || op == MARAY_OP_SKIPNZ; };
    auto is_out = [&](uint32_t j) { return MARAY_INS_OP(tape[j]) == MARAY_OP_OUT; };
    // units
    std::vector<uint32_t> unit_of(n), ubegin, uend;
    for (uint32_t j = 0; j < n;) {
        uint32_t e = j;
        for (uint32_t k = j; k <= e && k < n; k++)
            if (is_skip(k)) e = std::max(e, std::min(n - 1, k + MARAY_INS_AUX(tape[k])));
        for (uint32_t k = j; k <= e; k++) unit_of[k] = (uint32_t)ubegin.size();
        ubegin.push_back(j); uend.push_back(e);
        j = e + 1;
    }
    const uint32_t nu = (uint32_t)ubegin.size();
    // values: a value is named by the op that ends its definition (a SKIP op with a dst defines its region's value)
    std::vector<int32_t> ua(n, -1), ub(n, -1);
    {
        std::vector<int32_t> cur(MARAY_DST_NONE + 1, -1);
        int32_t acc = -1;
        auto value = [&](uint32_t ref) -> int32_t {
            if (MARAY_REF_KIND(ref) == MARAY_K_SLOT) return cur[MARAY_REF_INDEX(ref)];
            if (MARAY_REF_KIND(ref) == MARAY_K_SPEC && MARAY_REF_INDEX(ref) == MARAY_SPEC_ACC) return acc;
            return -1;
        };
        for (uint32_t j = 0; j < n; j++) {
            const uint64_t ins = tape[j];
            const uint32_t op = MARAY_INS_OP(ins), dst = MARAY_INS_DST(ins);
            if (op == MARAY_OP_NOP) continue;
            if (op != MARAY_OP_TEXDIM) ua[j] = value(MARAY_INS_A(ins));
            if (op >= MARAY_OP_ADD && op <= MARAY_OP_APP) ub[j] = value(MARAY_INS_B(ins));
            if (op == MARAY_OP_OUT) continue;
            if (is_skip(j)) { if (dst != MARAY_DST_NONE) cur[dst] = (int32_t)std::min(n - 1, j + MARAY_INS_AUX(ins)); continue; }
            acc = (int32_t)j;
            if (dst != MARAY_DST_NONE) cur[dst] = (int32_t)j;
        }
    }
    // unit DAG; the original order is a topological order of it
    std::vector<std::vector<uint32_t>> kids(nu);
    for (uint32_t j = 0; j < n; j++)
        for (int32_t v : {ua[j], ub[j]})
            if (v >= 0 && unit_of[v] != unit_of[j]) kids[unit_of[j]].push_back(unit_of[v]);
    std::vector<uint32_t> need(nu, 1);
    for (uint32_t u = 0; u < nu; u++) {
        auto &k = kids[u];
        std::sort(k.begin(), k.end());
        k.erase(std::unique(k.begin(), k.end()), k.end());
        std::stable_sort(k.begin(), k.end(), [&](uint32_t x, uint32_t y) { return need[x] > need[y]; });
        for (uint32_t i = 0; i < k.size(); i++) need[u] = std::max(need[u], need[k[i]] + i);
    }
    std::vector<uint32_t> seq;
    seq.reserve(n);
    {
        std::vector<uint8_t> done(nu, 0);
        std::vector<std::pair<uint32_t, uint32_t>> stack;
        for (uint32_t r = 0; r < nu; r++) {
            if (!is_out(ubegin[r]) || done[r]) continue;
            stack.push_back({r, 0});
            done[r] = 1;
            while (!stack.empty()) {
                auto &top = stack.back();
                if (top.second < kids[top.first].size()) {
                    const uint32_t c = kids[top.first][top.second++];
                    if (!done[c]) { done[c] = 1; stack.push_back({c, 0}); }
                    continue;
                }
                for (uint32_t k = ubegin[top.first]; k <= uend[top.first]; k++) seq.push_back(k);
                stack.pop_back();
            }
        }
    }
    const uint32_t m = (uint32_t)seq.size();
    std::vector<int32_t> newpos(n, -1);
    for (uint32_t p = 0; p < m; p++) newpos[seq[p]] = (int32_t)p;
    // may the op at position p read value v through ACC?  SKIP ops (not taken, or p is not reached) and OUT ops leave ACC alone
    auto acc_ok = [&](uint32_t p, int32_t v) {
        int32_t q = (int32_t)p - 1;
        while (q >= 0 && (is_skip(seq[q]) || is_out(seq[q]))) q--;
        return q >= 0 && (int32_t)seq[q] == v;
    };
    std::vector<uint8_t> needs_slot(n, 0);
    std::vector<int32_t> last_use(n, -1), first_def(n, -1);
    for (uint32_t p = 0; p < m; p++) {
        const uint32_t j = seq[p];
        for (int32_t v : {ua[j], ub[j]}) {
            if (v < 0) continue;
            if (!acc_ok(p, v)) { needs_slot[v] = 1; last_use[v] = (int32_t)p; }
        }
        if (is_out(j)) continue;
        const int32_t v = is_skip(j) ? (int32_t)std::min(n - 1, j + MARAY_INS_AUX(tape[j])) : (int32_t)j;
        if (first_def[v] < 0) first_def[v] = (int32_t)p;
        if (is_skip(j)) continue;
        // a region's value lives to the op that ends it at least
        if (first_def[v] != (int32_t)p) last_use[v] = std::max(last_use[v], (int32_t)p);
    }
    std::vector<std::vector<int32_t>> dies(m);
    for (uint32_t v = 0; v < n; v++) if (needs_slot[v] && last_use[v] >= 0) dies[last_use[v]].push_back((int32_t)v);
    std::vector<int32_t> slot_of(n, -1);
    std::vector<uint32_t> free_list;
    uint32_t used = 0;
    std::vector<uint64_t> out(m);
    for (uint32_t p = 0; p < m; p++) {
        const uint32_t j = seq[p];
        const uint64_t ins = tape[j];
        const uint32_t op = MARAY_INS_OP(ins);
        uint32_t a = MARAY_INS_A(ins), b = MARAY_INS_B(ins), dst = MARAY_DST_NONE;
        const uint32_t acc_ref = MARAY_REF(MARAY_K_SPEC, MARAY_SPEC_ACC);
        if (ua[j] >= 0) a = acc_ok(p, ua[j]) ? acc_ref : MARAY_REF(MARAY_K_SLOT, (uint32_t)slot_of[ua[j]]);
        if (ub[j] >= 0) b = acc_ok(p, ub[j]) ? acc_ref : MARAY_REF(MARAY_K_SLOT, (uint32_t)slot_of[ub[j]]);
        const int32_t v = is_out(j) ? -1 : (is_skip(j) ? (int32_t)std::min(n - 1, j + MARAY_INS_AUX(ins)) : (int32_t)j);
        // operands are read before the result is written: a value that dies here may hand its slot to this op's result
        for (int32_t d : dies[p]) if (d != v) free_list.push_back((uint32_t)slot_of[d]);
        if (v >= 0 && needs_slot[v]) {
            if (slot_of[v] < 0) {
                if (free_list.empty()) free_list.push_back(used++);
                slot_of[v] = (int32_t)free_list.back();
                free_list.pop_back();
            }
            dst = (uint32_t)slot_of[v];
        }
        for (int32_t d : dies[p]) if (d == v) free_list.push_back((uint32_t)slot_of[d]);
        out[p] = MARAY_INS(op, MARAY_INS_AUX(ins), dst, a, b);
    }
    // a cone with many outputs can do better in the order the section was scheduled in: never return the worse of the two
    std::vector<uint64_t> kept = tape;
    const uint32_t used_kept = renumber_slots(kept);
    if (used_kept <= used) { tape.swap(kept); return used_kept; }
    tape.swap(out);
    return used;
}

bool any_guard_reads_y(const maray_program &P)
{
    const RowTapeDeps d = row_tape_deps(P);
    const uint32_t n_ynum = numeric_yvals(P);
    for (uint32_t o : d.outs) if (MARAY_INS_AUX(P.row_ops[o]) >= n_ynum && d.reads_y[o]) return true;
    return false;
}

}   // namespace maray
