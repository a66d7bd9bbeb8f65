// backend.hpp — device evaluator interface behind the tape-level ABI (product code).
#pragma once

#include <array>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "expr.hpp"
#include "maray_hip.h"

namespace maray {

// The image rows of one launch: n_rows rows in blocks of block_rows consecutive ones, block_stride apart.
// Launch row r is image row y0 + (r / block_rows) * block_stride + r % block_rows; outputs are packed in
// launch-row order.  A contiguous range [y0, y1) is one block: {y0, y1 - y0, y1 - y0, 0}.
struct RowBlocks {
    uint32_t y0, n_rows, block_rows, block_stride;
    static RowBlocks range(uint32_t y0, uint32_t y1) { return RowBlocks{y0, y1 - y0, y1 > y0 ? y1 - y0 : 1u, 0u}; }
};

struct RowTile { uint32_t y0, y1; };      // rows [y0, y1)

struct Backend {
    virtual ~Backend() {}
    // the rows of `rb` into device buffers, enqueued on `stream`, no synchronisation
    virtual void render_device(uint32_t w, uint32_t h, const RowBlocks &rb, void *d8, void *d64, void *stream) = 0;
    // The row tiles, in order, into host rasters whose first byte is image row `row0` (blocking; either raster may
    // be null).  Tile k's device -> host copy runs under tile k + 1's kernels (host_pipe.hpp); `done` (may be empty)
    // is called on this thread once a tile's rows are in the raster.
    virtual void render_host_tiles(uint32_t w, uint32_t h, const std::vector<RowTile> &tiles, uint32_t row0, uint8_t *rgb8, double *rgb64,
                                   const std::function<void(uint32_t, uint32_t)> &done) = 0;
    // average ms per launch of the pixel kernel, HIP events on the launch stream
    virtual float time_rows(uint32_t w, uint32_t h, const RowBlocks &rb, void *d8, void *d64, int reps) = 0;
    virtual const char *kernel_name() const = 0;
    // Tape interpreter only: re-evaluate the 256-pixel tiles of the device work list
    // {count, tile, tile, ...} (tile = row_in_launch * ceil(w/256) + x/256), reading the row
    // values from `yvals` instead of running the ROW section.  Enqueued on `stream`.
    virtual void render_flagged(uint32_t, const RowBlocks &, void *, void *, void *, const unsigned *, const double *) {
        throw Error{MARAY_E_INTERNAL, "render_flagged is not supported by this backend"};
    }
};

// host_pipe.cpp: rows [y0, y1) of a w-wide image cut into tiles of a few MiB of output; pinned host memory
std::vector<RowTile> cut_row_tiles(uint32_t w, uint32_t y0, uint32_t y1, bool want8, bool want64);
void *host_alloc_pinned(size_t bytes);          // throws Error
void host_free_pinned(void *p);
void host_register(void *p, size_t bytes);      // throws Error
void host_unregister(void *p);                  // throws Error
bool host_range_is_pinned(const void *p, size_t n);

void set_last_error(const std::string &m);   // thread-local message behind maray_last_error()
int read_whole_file(const char *path, std::vector<uint8_t> &b);                                             // png.cpp; MARAY_E_IO + message on failure
int png_decode(const std::vector<uint8_t> &b, uint8_t **rgb8_out, uint32_t *w_out, uint32_t *h_out);       // png.cpp
int hip_device_count();
Backend *make_tape_backend(int device, const maray_program &prog, const maray_texture *tex, uint32_t n_tex, bool lds_variant);
Backend *make_jit_backend(int device, const maray_program &prog, const maray_texture *tex, uint32_t n_tex);
std::string jit_source(const maray_program &prog, int min_waves = 0);   // PIXEL kernel source (__launch_bounds__(256, min_waves); 0 = the layout's default); throws Error
std::string jit_source_rows(const maray_program &prog, uint32_t *n_chunks_out = nullptr, uint32_t *n_gjobs_out = nullptr);   // ROW kernel source (blockIdx.y = chunk)
void jit_compile(const std::string &src, std::vector<char> &code, std::string &log);     // hiprtc, gfx950; throws Error
// The two code objects of a program (jit_build.cpp): built once per process, kept on disk under MARAY_CACHE_DIR.
struct JitCode {
    std::vector<char> pix, rows;        // maray_jit_pixels; maray_jit_rows + maray_jit_order (empty without a ROW section)
    uint32_t n_row_chunks = 1, n_gjobs = 0;
    uint32_t n_gwords = 0, guard_w = 256, guard_h = 1;      // guard words per rectangle and the rectangle they are bounded over (what the sources were generated for)
    bool rows2 = false;                 // the PIXEL kernel's busy tiles take two rows per wavefront (jit_rows2): launches may pass rpw = 2
    int waves = 8;                      // the __launch_bounds__ occupancy the PIXEL kernel was built for
    bool from_disk = false;
};
std::shared_ptr<const JitCode> jit_code_for(const maray_program &prog);      // needs no GPU; throws Error
std::string jit_code_key(const maray_program &prog);                        // 128-bit hash (hex) of generated sources + toolchain: the cache's file name
bool jit_code_is_cached(const maray_program &prog);                         // in this process or on disk: no hiprtc build needed
void validate_program(const maray_program &p);

// ---- ROW-tape analysis shared by the evaluators (row_split.cpp) ------------------------------------
// Number of leading y values that PIXEL ops read as arithmetic operands; the rest of the table only gates SKIP ops.
uint32_t numeric_yvals(const maray_program &P);
struct RowTapeDeps {
    std::vector<std::array<int32_t, 2>> deps;   // per op: the ops that produce its operands (-1: none)
    std::vector<uint32_t> outs;                 // OUT ops, tape order
    std::vector<uint8_t> reads_y;               // per op: SPEC Y is somewhere in its cone
};
RowTapeDeps row_tape_deps(const maray_program &P);
// The ROW tape with everything but the cone of the given OUT ops turned into NOPs (SKIP regions kept when their end is).
std::vector<uint64_t> row_tape_cone(const maray_program &P, const RowTapeDeps &d, const std::vector<uint32_t> &outs, size_t *cost);
// The same tape without its NOPs (SKIP op counts adjusted).
std::vector<uint64_t> compact_tape(const std::vector<uint64_t> &tape);
// Renumbers the value slots of a compacted tape by liveness; returns the number of slots it then uses.
uint32_t renumber_slots(std::vector<uint64_t> &tape);
uint32_t reschedule_tape(std::vector<uint64_t> &tape);   // cone re-ordered depth first + slots by liveness
// Does any guard (a y value that only gates SKIP ops) have SPEC Y in its cone?  If none does, guards may be evaluated
// once for a group of rows (YMIN / YMAX, include/maray_tape.h).
bool any_guard_reads_y(const maray_program &P);

}   // namespace maray

struct maray_ctx {
    maray::Backend *backend = nullptr;
    uint32_t n_tex = 0;
};
