// jit_parts.hpp — what the three translation units of the specialising back end share (product code):
//   jit_source.cpp (+ jit_emit.hpp)  tape -> HIP source of the ROW and PIXEL kernels
//   jit_build.cpp                    source -> code objects: hiprtc, helper processes, code key, caches
//   jit_backend.cpp                  code objects -> launches (JitBackend)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "backend.hpp"

namespace maray {

// of jit_compile; part of the cache key.  -structurizecfg-skip-uniform-regions: the generated kernels branch on wave-uniform
// conditions throughout (lane masks tested on the scalar unit), and the branch table of a guarded OR-reduction (an asm
// goto inside a loop) only survives when the structurizer leaves uniform regions alone: without the option the back end
// rewrites the table's edges into tests of flags nobody sets
// -O2, not -O3: the two differ in a handful of instructions of chess's PIXEL kernel (170 lines of 13,000 of disassembly) and in nothing
// measurable at run time (frame 27.2 / 27.2 us, config 3b 163-172 / 163-166, config 5 39.1 / 39.1), and -O2 builds that kernel -- the
// long pole of a cold start: the ROW kernel takes as long either way -- a quarter faster: cold ctx_ms 1.68-1.73 -> 1.28-1.38 s
static const char JIT_OPTIONS[] = "--offload-arch=gfx950 -O2 -ffp-contract=off -fno-fast-math -std=c++17 -mllvm -structurizecfg-skip-uniform-regions";

static const uint32_t GW_INLINE_MAX = 12;       // up to this many guard words a rectangle's words are named SGPR pairs; beyond, they stay one per lane (v_readlane per test)
static const unsigned ROW_BLOCK = 256;          // threads per block of the ROW kernel (64 ... 1024 move a chess frame by less than a microsecond)

struct GuardGeom { uint32_t gw, gh; };          // the rectangle a guard is bounded over: gw pixels x gh rows (jit_guard_geom)

std::vector<std::string> jit_option_words();                          // JIT_OPTIONS with MARAY_JIT_OPT / MARAY_JIT_EXTRA applied: what every compile is given (jit_build.cpp)
bool may_defer_tiles(const maray_program &P);                       // some Sin whose argument is not provably bounded: tiles may go to the interpreter
uint32_t jit_guard_words(const maray_program &P);                   // 64-bit words of guard bits per rectangle
GuardGeom jit_guard_geom(const maray_program &P);
bool jit_wide_general(const maray_program &P, uint32_t n_gwords);   // the general section four pixels per lane
bool jit_rows2(const maray_program &P);                             // busy tiles two rows per wavefront (the launch passes rpw = 2 when its guard groups allow)

}   // namespace maray
