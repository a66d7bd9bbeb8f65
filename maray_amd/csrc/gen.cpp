// gen.cpp — render façade: gen_to_image / gen (product code).
//
//   maray_gen_to_image <- gen_to_image   src/lib.rs:1177-1195 (the RenderMethod match)
//   maray_gen          <- gen            src/lib.rs:1199-1213
//   ReportState        <- Report::start / Report::update   src/report.rs:26-56
//
// Multi-GPU: pixels are independent (src/render.rs:85), so the image is cut into row tiles dealt
// round-robin to the devices.  One host thread per device creates that device's context (the
// specialised kernels are built once, by the first thread, and loaded by the others:
// jit_code_for) and renders its tiles straight into the caller's raster: host-side gather, no
// collective.  The raster is pinned for the duration of the call when it is not already, so every
// device writes its rows by DMA (host_pipe.hpp); the progress callback runs on the calling thread,
// like the reference's collector loop (src/render.rs:63-82).
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "backend.hpp"
#include "lower.hpp"
#include "maray_hip.h"

using namespace maray;

namespace {

struct ReportState {   // src/report.rs:6-13, :37-56
    maray_report r;
    uint32_t last_row = 0;
    std::chrono::steady_clock::time_point last_t;
    explicit ReportState(maray_report rep) : r(rep), last_t(std::chrono::steady_clock::now()) {}
    bool update(uint32_t row) {
        switch (r.kind) {
        case MARAY_REPORT_ROW:
            if (r.value && row >= last_row + r.value) { last_row += r.value; return true; }
            return false;
        case MARAY_REPORT_DURATION_MS: {
            auto now = std::chrono::steady_clock::now();
            if (now >= last_t + std::chrono::milliseconds(r.value)) { last_t = now; return true; }
            return false;
        }
        default: return false;
        }
    }
};

struct Progress {
    std::mutex m;
    std::condition_variable cv;
    uint64_t rows_done = 0;
    uint32_t max_row = 0;
    bool any = false;
    uint32_t finished = 0;      // workers that have returned
    int failed = 0;
    std::string err;
};

// Pins the caller's raster for the time of a render unless it is pinned already; a raster that cannot be pinned
// (registration refused, or too small to be worth a system call) is filled through the contexts' staging rings.
struct ScopedPin {
    void *p = nullptr;
    ScopedPin(uint8_t *raster, size_t bytes) {
        const char *e = getenv("MARAY_GEN_REGISTER");                 // "0": never register (measurement knob)
        if ((e && e[0] == '0') || bytes < ((size_t)4 << 20) || host_range_is_pinned(raster, bytes)) return;
        try { host_register(raster, bytes); p = raster; } catch (const Error &) {}
    }
    ~ScopedPin() { if (p) { try { host_unregister(p); } catch (const Error &) {} } }
};

int fail_with(int code, const std::string &m) { set_last_error(m); return code; }

template <typename F>
int gen_guard(F f)
{
    try { return f(); }
    catch (const Error &e) { return fail_with(e.code, e.msg); }
    catch (const std::bad_alloc &) { return fail_with(MARAY_E_INTERNAL, "out of memory"); }
    catch (const std::exception &e) { return fail_with(MARAY_E_INTERNAL, e.what()); }
    catch (...) { return fail_with(MARAY_E_INTERNAL, "unknown error"); }
}

}   // namespace

extern "C" int maray_lower(const maray_scene *, const maray_lower_opts *, maray_tape **);

extern "C" int maray_gen_to_image(const maray_scene *s, const maray_texture *tex, uint32_t n_tex,
                                  const maray_gen_opts *opts, maray_report report, maray_report_fn fn, void *user,
                                  uint8_t *rgb8, uint32_t w, uint32_t h)
{
    if (!s || !rgb8) return fail_with(MARAY_E_ARG, "null argument");
    return gen_guard([&]() -> int {
    if (w > MARAY_DOMAIN_MAX || h > MARAY_DOMAIN_MAX) return fail_with(MARAY_E_LIMIT, "image exceeds " + std::to_string(MARAY_DOMAIN_MAX) + " pixels in x or y");
    if (!w || !h) return MARAY_OK;
    maray_tape *tape = nullptr;
    int rc = maray_lower(s, nullptr, &tape);
    if (rc) return rc;
    struct TapeHolder { maray_tape *t; ~TapeHolder() { maray_tape_free(t); } } hold{tape};
    maray_program prog;
    maray_tape_program(tape, &prog);

    int n_dev_avail = 0;
    maray_hip_device_count(&n_dev_avail);
    uint32_t n_dev = opts && opts->n_devices ? opts->n_devices : (uint32_t)(n_dev_avail > 0 ? n_dev_avail : 1);
    // MARAY_GEN_WRAP_DEVICES=1 (tests): worker d drives physical device d mod visible -- the whole multi-device machinery
    // (a thread and a context per worker, one shared build, one registered raster written by every context) on a one-GPU box
    const char *env_wrap = getenv("MARAY_GEN_WRAP_DEVICES");
    const bool wrap = env_wrap && env_wrap[0] == '1' && n_dev_avail > 0;
    if (n_dev_avail > 0 && n_dev > (uint32_t)n_dev_avail && !wrap)
        return fail_with(MARAY_E_NO_DEVICE, "asked for " + std::to_string(n_dev) + " devices, " + std::to_string(n_dev_avail) + " visible");
    if (n_dev > (h + 7) / 8) n_dev = (h + 7) / 8;
    // Row tiles: ~24 MiB of raster each (the DMA engine's rate; the first copy starts early; host_pipe.cpp), but at least
    // four per device, dealt round-robin: the cost of a row depends on what it shows (the kernels skip work tile by tile),
    // so contiguous bands would leave the device with the busiest band behind.  Multiples of 32 rows (guard groups) where
    // the image has that many.
    uint32_t tile_rows = opts && opts->tile_rows ? opts->tile_rows : 0;
    if (!tile_rows) {
        auto whole_groups = [](uint64_t r) { return r >= 32 ? r / 32 * 32 : std::max<uint64_t>(8, r / 8 * 8); };
        const uint64_t by_bytes = whole_groups(((uint64_t)24 << 20) / ((uint64_t)w * 3));
        const uint64_t by_share = whole_groups(((uint64_t)h / (4ull * n_dev) + 7));
        tile_rows = (uint32_t)std::min(by_bytes, n_dev > 1 ? by_share : by_bytes);
    }
    std::vector<std::vector<uint32_t>> share(n_dev);          // per device: y0, y1 pairs
    {
        uint32_t k = 0;
        for (uint64_t y = 0; y < h; y += tile_rows, k++) {
            share[k % n_dev].push_back((uint32_t)y);
            share[k % n_dev].push_back((uint32_t)std::min<uint64_t>(h, y + tile_rows));
        }
    }
    maray_ctx_opts co;
    memset(&co, 0, sizeof co);
    co.backend = opts ? opts->backend : MARAY_BACKEND_AUTO;
    co.hint_mpixels = (uint32_t)std::min<uint64_t>(0xFFFFFFFFu, (((uint64_t)w * h / n_dev) >> 20) + 1);

    ScopedPin pin(rgb8, (size_t)w * h * 3);
    Progress P;
    struct TileUser { Progress *P; };
    TileUser tu{&P};
    auto on_tile = [](void *u, uint32_t y0, uint32_t y1) {
        Progress &Q = *((TileUser *)u)->P;
        std::lock_guard<std::mutex> lk(Q.m);
        Q.rows_done += y1 - y0;
        Q.max_row = std::max(Q.max_row, y1 - 1);
        Q.any = true;
        Q.cv.notify_all();
    };
    auto worker = [&](uint32_t d) {
        maray_ctx *ctx = nullptr;
        int r = maray_hip_ctx_create(wrap ? (int)(d % (uint32_t)n_dev_avail) : (int)d, &prog, tex, n_tex, &co, &ctx);
        if (!r && !share[d].empty())
            r = maray_hip_render_tiles(ctx, w, h, share[d].data(), (uint32_t)(share[d].size() / 2), rgb8, on_tile, &tu);
        const std::string msg = r ? maray_last_error() : "";      // this thread's message, re-raised on the calling thread
        maray_hip_ctx_free(ctx);
        std::lock_guard<std::mutex> lk(P.m);
        if (r && !P.failed) { P.failed = r; P.err = msg; }
        P.finished++;
        P.cv.notify_all();
    };
    std::vector<std::thread> th;
    for (uint32_t d = 0; d < n_dev; d++) th.emplace_back(worker, d);

    {   // collector loop of the reference (src/render.rs:63-82), on the calling thread
        ReportState rs(report);
        std::unique_lock<std::mutex> lk(P.m);
        while (P.finished < n_dev) {
            P.cv.wait_for(lk, std::chrono::milliseconds(10));
            if (P.any && fn && !P.failed && P.rows_done < h && rs.update(P.max_row)) {
                const double prog_frac = (double)P.max_row / (double)h;
                lk.unlock();
                fn(user, rgb8, w, h, prog_frac);
                lk.lock();
            }
            P.any = false;
        }
    }
    for (auto &t : th) t.join();
    if (P.failed) return fail_with(P.failed, P.err);
    set_last_error("");
    return MARAY_OK;
    });
}

namespace {
struct GenUser { std::string path; };
void gen_report(void *user, uint8_t *rgb8, uint32_t w, uint32_t h, double progress)
{
    GenUser *g = (GenUser *)user;
    fprintf(stderr, "%.2f %%\n", 100.0 * progress);   // src/lib.rs:1205
    fflush(stderr);
    maray_png_write(g->path.c_str(), rgb8, w, h);     // partial image re-saved (:1207)
}
}   // namespace

extern "C" int maray_gen(const maray_scene *s, const maray_texture *tex, uint32_t n_tex,
                         const maray_gen_opts *opts, maray_report report, const char *png_path)
{
    if (!s || !png_path) return fail_with(MARAY_E_ARG, "null argument");
    return gen_guard([&]() -> int {
    uint32_t w = 0, h = 0;
    maray_scene_size(s, &w, &h);
    if (w > MARAY_DOMAIN_MAX || h > MARAY_DOMAIN_MAX) return fail_with(MARAY_E_LIMIT, "image exceeds " + std::to_string(MARAY_DOMAIN_MAX) + " pixels in x or y");
    const size_t bytes = (size_t)w * h * 3;
    // RgbImage::new (:1210) -- in pinned memory, so that the devices write it by DMA; zeroed like the reference's
    struct Raster {
        uint8_t *p = nullptr; bool pinned = false;
        ~Raster() { if (pinned) host_free_pinned(p); else free(p); }
    } img;
    try { img.p = (uint8_t *)host_alloc_pinned(bytes); img.pinned = true; }
    catch (const Error &) { img.p = (uint8_t *)malloc(bytes ? bytes : 1); }      // no device: the render below reports it
    if (!img.p) return fail_with(MARAY_E_INTERNAL, "out of memory");
    memset(img.p, 0, bytes);
    GenUser g{png_path};
    int rc = maray_gen_to_image(s, tex, n_tex, opts, report, gen_report, &g, img.p, w, h);
    if (rc) return rc;
    return maray_png_write(png_path, img.p, w, h);   // :1212
    });
}
