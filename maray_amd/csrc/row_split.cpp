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

bool any_guard_reads_y(const maray_program &P)
{
    const RowTapeDeps d = row_tape_deps(P);
    const uint32_t n_ynum = numeric_yvals(P);
    for (uint32_t o : d.outs) if (MARAY_INS_AUX(P.row_ops[o]) >= n_ynum && d.reads_y[o]) return true;
    return false;
}

}   // namespace maray
