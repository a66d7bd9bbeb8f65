// lower.hpp — lowered program container (product code).
#pragma once

#include <array>
#include <cstdint>
#include <vector>

#include "expr.hpp"
#include "maray_hip.h"

namespace maray {

struct Tape {
    std::vector<double> consts;
    std::vector<uint64_t> row_ops;
    std::vector<uint64_t> pix_ops;
    maray_tape_info info;

    maray_program program() const {
        maray_program p;
        p.version = MARAY_TAPE_VERSION;
        p.n_consts = (uint32_t)consts.size();
        p.consts = consts.data();
        p.n_row_ops = (uint32_t)row_ops.size();
        p.row_ops = row_ops.data();
        p.n_row_slots = info.n_row_slots;
        p.n_yvals = info.n_yvals;
        p.n_pix_ops = (uint32_t)pix_ops.size();
        p.pix_ops = pix_ops.data();
        p.n_pix_slots = info.n_pix_slots;
        p.n_app = info.n_app;
        return p;
    }
};

void lower_scene(const Scene &scene, const maray_lower_opts &opts, Tape &out);   // throws Error

// Validate a program handed in through the tape-level ABI (bounds of every
// slot / constant / y-value / output reference); throws Error.
void validate_program(const maray_program &p);

}   // namespace maray
