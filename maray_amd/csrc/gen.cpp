// gen.cpp — render façade: gen_to_image / gen (product code).
//
//   maray_gen_to_image <- gen_to_image   src/lib.rs:1177-1195 (the RenderMethod match)
//   maray_gen          <- gen            src/lib.rs:1199-1213
//   ReportState        <- Report::start / Report::update   src/report.rs:26-56
//
// Multi-GPU: pixels are independent (src/render.rs:85), so the image is cut
// into row tiles dealt round-robin to the devices, each rendered by its own
// context on its own host thread straight into the caller's raster (host-side
// gather, no collective).  The progress callback runs on the calling thread.
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
    uint32_t rows_done = 0;
    uint32_t max_row = 0;
    bool any = false;
    int failed = 0;
    Error err{0, ""};
};

}   // namespace

extern "C" int maray_lower(const maray_scene *, const maray_lower_opts *, maray_tape **);

extern "C" int maray_gen_to_image(const maray_scene *s, const maray_texture *tex, uint32_t n_tex,
                                  const maray_gen_opts *opts, maray_report report, maray_report_fn fn, void *user,
                                  uint8_t *rgb8, uint32_t w, uint32_t h)
{
    if (!s || !rgb8) return MARAY_E_ARG;
    maray_tape *tape = nullptr;
    int rc = maray_lower(s, nullptr, &tape);
    if (rc) return rc;
    maray_program prog;
    maray_tape_program(tape, &prog);

    int n_dev_avail = 0;
    maray_hip_device_count(&n_dev_avail);
    uint32_t n_dev = opts && opts->n_devices ? opts->n_devices : (uint32_t)(n_dev_avail > 0 ? n_dev_avail : 1);
    if (n_dev > h && h) n_dev = h;
    const uint32_t tile_rows = opts && opts->tile_rows ? opts->tile_rows : 256;
    maray_ctx_opts co;
    memset(&co, 0, sizeof co);
    co.backend = opts ? opts->backend : MARAY_BACKEND_AUTO;

    std::vector<maray_ctx *> ctxs(n_dev, nullptr);
    for (uint32_t d = 0; d < n_dev; d++) {
        rc = maray_hip_ctx_create((int)d, &prog, tex, n_tex, &co, &ctxs[d]);
        if (rc) {
            const std::string keep = maray_last_error();
            for (auto c : ctxs) maray_hip_ctx_free(c);
            maray_tape_free(tape);
            set_last_error(keep);
            return rc;
        }
    }

    Progress P;
    // Row tiles are dealt to the devices round-robin: the cost of a row depends on what it shows (the
    // kernels skip work wave by wave), so contiguous bands would leave the device with the busiest band behind.
    auto worker = [&](uint32_t d) {
        for (uint64_t y64 = (uint64_t)d * tile_rows; y64 < h; y64 += (uint64_t)n_dev * tile_rows) {
            const uint32_t y = (uint32_t)y64;
            const uint32_t y1 = (uint32_t)std::min<uint64_t>(h, y64 + tile_rows);
            int r = maray_hip_render_rows(ctxs[d], w, h, y, y1, rgb8 + (size_t)y * w * 3, nullptr);
            std::lock_guard<std::mutex> lk(P.m);
            if (r) { P.failed = r; P.err = Error{r, maray_last_error()}; P.cv.notify_all(); return; }
            P.rows_done += y1 - y;
            P.max_row = std::max(P.max_row, y1 - 1);
            P.any = true;
            P.cv.notify_all();
            if (P.failed) return;
        }
    };
    std::vector<std::thread> th;
    for (uint32_t d = 0; d < n_dev; d++) th.emplace_back(worker, d);

    {   // collector loop of the reference (src/render.rs:63-82), on the calling thread
        ReportState rs(report);
        std::unique_lock<std::mutex> lk(P.m);
        while (P.rows_done < h && !P.failed) {
            P.cv.wait_for(lk, std::chrono::milliseconds(10));
            if (P.any && fn && rs.update(P.max_row)) {
                const double prog_frac = (double)P.max_row / (double)h;
                lk.unlock();
                fn(user, rgb8, w, h, prog_frac);
                lk.lock();
            }
            P.any = false;
        }
    }
    for (auto &t : th) t.join();
    for (auto c : ctxs) maray_hip_ctx_free(c);
    maray_tape_free(tape);
    if (P.failed) {
        set_last_error(P.err.msg);   // the worker's thread-local message, re-raised on the calling thread
        return P.failed;
    }
    return MARAY_OK;
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
    if (!s || !png_path) return MARAY_E_ARG;
    uint32_t w = 0, h = 0;
    maray_scene_size(s, &w, &h);
    std::vector<uint8_t> img((size_t)w * h * 3, 0);   // RgbImage::new (:1210)
    GenUser g{png_path};
    int rc = maray_gen_to_image(s, tex, n_tex, opts, report, gen_report, &g, img.data(), w, h);
    if (rc) return rc;
    return maray_png_write(png_path, img.data(), w, h);   // :1212
}
