// jit_backend.cpp — the specialised kernels of a program on one device (product code): code objects -> modules, tables,
// launches.  The GPU analogue of the reference's
//
//   JitBackend   <- wasm_par_gen_to_image                         src/render.rs:102-192
//
// The sources come from jit_source.cpp, the code objects from jit_build.cpp (jit_parts.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "host_pipe.hpp"
#include "jit_parts.hpp"
#include "maray_hip.h"

namespace maray {


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
    std::unique_ptr<HostPipe> pipe;     // streams + staging of the host-raster entry points (from the device's pool: host_pipe.hpp)
    hipStream_t own_stream = nullptr;   // = pipe's compute stream
    uint32_t n_row_chunks = 1, n_gwords = 0, n_gjobs = 0, guard_rows = 1, guard_sub = 1;       // guard_sub: guard rectangles per 256-pixel tile
    uint32_t n_cu = 256;
    bool wide_all = false;              // the whole section runs four pixels per lane (jit_wide_general)
    bool rows2 = false;                 // busy tiles two rows per wavefront (JitCode::rows2)
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
        if (pipe) { (void)hipStreamSynchronize(pipe->compute_stream()); host_pipe_release(std::move(pipe)); }
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
        rows2 = code->rows2;
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
        pipe = host_pipe_acquire(dev);
        lap("host pipe");
        own_stream = pipe->compute_stream();
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
        // Tiles per wavefront.  Every tile of a program without guards costs the same: nothing to balance, and a strip as long as
        // the launch is deep -- a wavefront's tiles come one after the other, so a launch that fills the device once is as
        // long as its longest strip (config 2, RGB8, us per frame with 1 / 2 / 4 / 8 / 16 tiles: 1024^2 3.03 / 3.14 / 3.53 (whole
        // rows); 2048^2 6.86 / 6.70 / 6.99 / 7.63; 4096^2 19.4 / 18.4 / 18.4 / 21.5 / 20.2; 8192^2 65.7 / 63.0 / 63.0 / 60.1 / 69.5; only
        // for the cheap four-wide sections, a narrow wavefront would live too long).  Else rows cost what they show (sky: nothing, board: 5x the average) and wavefronts are the unit of load
        // balance: a launch that fills the device only a few times over -- chess up to 2048^2 -- is a matter of how long its
        // longest wavefront lives, not of throughput: one tile per wavefront (1024^2 / 2048^2 / 4096^2 / 8192^2 with 1 tile:
        // 20.3 / 18.9 / 48.4 / 149 us per step, with 2: 25.5 / 24.8 / 42.3 / 138); a launch that fills it dozens of times over
        // is a matter of what every wavefront costs before its first pixel, its tail is short against the whole: four tiles
        // (8192^2 / 16384^2 with 2 / 3 / 4 / 5 tiles: 138 / 123 / 119 / 139 and 517 / 487 / 497 / 566 us per step; 4096^2:
        // 42.2 / 55.5 / 44.6 / 46.9)
        const uint64_t device_slots = (uint64_t)n_cu * 4 * 7;
        unsigned tiles = wide_all ? (n_tiles <= device_slots ? 1 : n_tiles <= 4 * device_slots ? 2 : n_tiles <= 16 * device_slots ? 4 : 8)
                                  : (n_tiles <= 4 * device_slots ? 1 : n_tiles <= 16 * device_slots ? 2 : 4);
        // two rows per wavefront: when the launch's guard groups have an even number of rows (a pair then lies inside one group);
        // the strip is half as long, so that a wavefront owns as many pixels as it would with one row
        const unsigned rpw = (rows2 && yrows >= 2 && yrows % 2 == 0) ? 2u : 1u;
        if (rpw == 2 && tiles > 1) tiles /= 2;
        if (k_tiles) tiles = std::min(64u, k_tiles);
        if (n_gwords && n_gwords <= GW_INLINE_MAX) tiles = std::min(tiles, 64u / (n_gwords * guard_sub));      // a strip's guard words: one per lane
        tiles = std::max(1u, std::min(tiles, n_tx));
        const unsigned gx = (n_tx + 4 * tiles - 1) / (4 * tiles);
        if (n_tiles > 0xFFFFFFFFull) throw Error{MARAY_E_ARG, "too many tiles in one launch; render fewer rows per call"};
        if (has_sin) {
            ensure(d_flags, flags_cap, (size_t)n_tiles + 1);                // work list {count, tile, ...} of deferred tiles
            HIP_TRY(hipMemsetAsync(d_flags, 0, sizeof(unsigned), st));
        }
        const uint32_t rows_per_grid = 65534u * rpw;                   // gridDim.y limit (an even number of rows either way: pairs, 32-row groups)
        for (uint32_t r0 = 0; r0 < rows_total; r0 += rows_per_grid) {
            uint32_t rows = std::min<uint32_t>(rows_per_grid, rows_total - r0);
            unsigned char *p8 = d8 ? d8 + (size_t)r0 * w * 3 : nullptr;
            double *p64 = d64 ? d64 + (size_t)r0 * w * 3 : nullptr;
            const double *yv = d_yvals + (size_t)r0 * n_yvals;
            unsigned *fl = d_flags;
            unsigned ww = w, yy0 = y0, tile_base = r0 * n_tx, row_base = r0;
            const unsigned long long *gb = d_gbits;              // indexed by the row of the whole call
            unsigned ntx = n_tx;
            unsigned rpw_ = rpw;
            void *args[] = {&p8, &p64, &yv, &d_tex, &fl, &tile_base, &gb, &ntx, &ww, &yy0, &n_yvals, &tiles, &blk_rows, &blk_stride, &row_base, &yrows, &row_order, &rows, &rpw_};
            HIP_TRY(hipModuleLaunchKernel(f_pix, gx, (rows + rpw - 1) / rpw, 1, 256, 1, 1, 0, st, args, nullptr));
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
        pipe->run(w, tiles, row0, rgb8, rgb64,
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
