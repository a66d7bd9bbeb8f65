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
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "backend.hpp"
#include "host_pipe.hpp"
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
// Registering costs 2.4 ms per 48 MiB the first time the library sees a range and ~0.1 ms every later time (measured:
// pageable calls 3.4 ms, then 1.0-1.2 ms against 1.03 into memory that stays pinned), so the registration is NOT kept
// between calls: it would save a tenth of a millisecond and leave a DMA target behind in memory the caller may free.
struct ScopedPin {
    void *p = nullptr;
    void pin(uint8_t *raster, size_t bytes) {
        const char *e = getenv("MARAY_GEN_REGISTER");                 // "0": never register (measurement knob)
        if ((e && e[0] == '0') || bytes < ((size_t)4 << 20) || host_range_is_pinned(raster, bytes)) return;
        try { host_register(raster, bytes); p = raster; } catch (const Error &) {}
    }
    ~ScopedPin() { if (p) { try { host_unregister(p); } catch (const Error &) {} } }
};

int fail_with(int code, const std::string &m) { set_last_error(m); return code; }

// ---- what the façade remembers between calls ------------------------------------------------------------------------
// gen_to_image is called once per image (src/lib.rs:1177-1195), an animation calls it in a loop (examples/test*.rs), and
// the reference pays its set-up -- fix_color, and for the JIT three cranelift compiles per thread -- on every call.  Here
// a call's set-up is the lowering (chess: ~0.1 s of host work) and a context per device (streams, two module loads,
// tables: ~50 ms), against a millisecond of rendering: so a scene's tape and its contexts stay, under the hash of the
// scene's encoding + the textures' bytes + the back-end asked for.  The last MARAY_GEN_CACHE programs (default 4; 0: none)
// are kept; maray_gen_cache_clear() frees them.  A context serves one call at a time: a second call for the same program
// while the first is still rendering gets contexts of its own.
struct IdleCtx {
    maray_ctx *ctx = nullptr;
    uint32_t hint_mpixels = 0;                // what MARAY_BACKEND_AUTO was told when it chose this context's evaluator (0: "many")
};
struct GenEntry {
    std::string key;
    maray_tape *tape = nullptr;
    std::multimap<int, IdleCtx> idle;         // per PHYSICAL device: contexts nobody is rendering with
    bool evicted = false;
    ~GenEntry() { for (auto &kv : idle) maray_hip_ctx_free(kv.second.ctx); maray_tape_free(tape); }
};
std::mutex g_gen_mutex;
// most recently used first.  Never destroyed: at process exit the HIP runtime may be gone before a static's destructor
// runs, and freeing a context then would call into it (device memory goes with the process anyway).
std::list<std::shared_ptr<GenEntry>> &g_gen = *new std::list<std::shared_ptr<GenEntry>>();

size_t gen_cache_capacity()
{
    static const size_t cap = [] {
        const char *e = getenv("MARAY_GEN_CACHE");
        return e ? (size_t)std::max(0, atoi(e)) : (size_t)4;
    }();
    return cap;
}

std::shared_ptr<GenEntry> gen_cache_find(const std::string &key)
{
    std::lock_guard<std::mutex> lk(g_gen_mutex);
    for (auto it = g_gen.begin(); it != g_gen.end(); ++it)
        if ((*it)->key == key) { g_gen.splice(g_gen.begin(), g_gen, it); return g_gen.front(); }
    return nullptr;
}

// `fresh` becomes the entry of its key unless another call got there first (then that one is used and `fresh` goes)
std::shared_ptr<GenEntry> gen_cache_insert(std::shared_ptr<GenEntry> fresh)
{
    std::vector<std::shared_ptr<GenEntry>> dead;          // freed outside the lock (contexts synchronise their streams)
    {
        std::lock_guard<std::mutex> lk(g_gen_mutex);
        for (auto &e : g_gen) if (e->key == fresh->key) return e;
        if (!gen_cache_capacity()) return fresh;          // not kept: goes with the call
        g_gen.push_front(fresh);
        while (g_gen.size() > gen_cache_capacity()) { g_gen.back()->evicted = true; dead.push_back(g_gen.back()); g_gen.pop_back(); }
    }
    return fresh;
}

// A context of `dev` that an earlier call left.  Keyed by the physical device, not by the worker that used it: after a
// call whose workers shared a device (MARAY_GEN_WRAP_DEVICES) worker 1's context lives on device 0, and a later call
// that gives worker 1 a device of its own must not be handed it.  Under MARAY_BACKEND_AUTO a context stands for the
// choice made for the FIRST call's size: an interpreter context taken for a thumbnail is not what a 16384^2 frame
// wants, so a kept interpreter context is dropped (the caller makes a new one, AUTO decides again) when this call is
// to render more than it was chosen for.
IdleCtx gen_cache_take(GenEntry &e, int dev, bool auto_backend, uint32_t hint_mpixels)
{
    maray_ctx *stale = nullptr;
    IdleCtx c;
    {
        std::lock_guard<std::mutex> lk(g_gen_mutex);
        auto it = e.idle.find(dev);
        if (it == e.idle.end()) return c;
        const IdleCtx ic = it->second;
        e.idle.erase(it);
        const bool interp = strncmp(maray_hip_kernel_name(ic.ctx), "maray_jit", 9) != 0;
        if (auto_backend && interp && ic.hint_mpixels != 0 && (hint_mpixels == 0 || hint_mpixels > ic.hint_mpixels)) stale = ic.ctx;
        else c = ic;
    }
    if (stale) maray_hip_ctx_free(stale);
    return c;
}

void gen_cache_give(GenEntry &e, int dev, maray_ctx *c, uint32_t hint_mpixels)
{
    {
        std::lock_guard<std::mutex> lk(g_gen_mutex);
        if (!e.evicted && gen_cache_capacity() && e.idle.count(dev) < 4) { e.idle.insert({dev, IdleCtx{c, hint_mpixels}}); return; }
    }
    maray_hip_ctx_free(c);
}

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
namespace maray {
void scene_cache_key(const maray_scene *s, uint64_t out[2]);      // api.cpp
void hash128(const void *data, size_t n, uint64_t h[2]);
}

extern "C" void maray_gen_cache_clear(void)
{
    std::list<std::shared_ptr<GenEntry>> dead;
    {
        std::lock_guard<std::mutex> lk(g_gen_mutex);
        for (auto &e : g_gen) e->evicted = true;
        dead.swap(g_gen);
    }
}

extern "C" int maray_gen_cache_info(char *out, size_t cap)
{
    if (!out || !cap) return fail_with(MARAY_E_ARG, "null argument");
    std::string t;
    {
        std::lock_guard<std::mutex> lk(g_gen_mutex);
        for (auto &e : g_gen)
            for (auto &kv : e->idle)
                t += e->key + " device " + std::to_string(kv.first) + " kernel " + maray_hip_kernel_name(kv.second.ctx) + " hint_mpixels " +
                     std::to_string(kv.second.hint_mpixels) + "\n";
    }
    snprintf(out, cap, "%s", t.c_str());
    return MARAY_OK;
}

extern "C" int maray_gen_to_image(const maray_scene *s, const maray_texture *tex, uint32_t n_tex,
                                  const maray_gen_opts *opts, maray_report report, maray_report_fn fn, void *user,
                                  uint8_t *rgb8, uint32_t w, uint32_t h)
{
    if (!s || !rgb8) return fail_with(MARAY_E_ARG, "null argument");
    return gen_guard([&]() -> int {
    if (w > MARAY_DOMAIN_MAX || h > MARAY_DOMAIN_MAX) return fail_with(MARAY_E_LIMIT, "image exceeds " + std::to_string(MARAY_DOMAIN_MAX) + " pixels in x or y");
    if (!w || !h) return MARAY_OK;
    // the program of this call: remembered from an earlier one, or lowered now
    std::string key;
    {
        uint64_t h[2];
        scene_cache_key(s, h);
        for (uint32_t i = 0; i < n_tex; i++) {
            const uint32_t dims[2] = {tex[i].w, tex[i].h};
            hash128(dims, sizeof dims, h);
            if (!tex[i].rgb && (uint64_t)tex[i].w * tex[i].h) return fail_with(MARAY_E_ARG, "null texture raster");
            hash128(tex[i].rgb, (size_t)tex[i].w * tex[i].h * 3, h);
        }
        char buf[64];
        snprintf(buf, sizeof buf, "%016llx%016llx/%u/%u", (unsigned long long)h[0], (unsigned long long)h[1], n_tex, opts ? opts->backend : (uint32_t)MARAY_BACKEND_AUTO);
        key = buf;
    }
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
    auto device_of = [&](uint32_t d) { return wrap ? (int)(d % (uint32_t)n_dev_avail) : (int)d; };

    // The program of this call: remembered from an earlier one, or lowered now.  A first call's set-up is two things that do
    // not depend on each other -- the lowering (host arithmetic: 20 ms for chess) and what a render needs from the device
    // whatever the program is (streams, events, the copy stream, the tiles' device slots, the raster's pages pinned: 10-15 ms
    // of driver calls) -- so the second runs on threads of its own while this one lowers (the reference compiles its three
    // modules on every thread of every call, src/render.rs:158-165).
    ScopedPin pin;
    std::shared_ptr<GenEntry> entry = gen_cache_find(key);
    if (!entry) {
        std::vector<std::thread> warm;
        struct Joiner { std::vector<std::thread> &v; ~Joiner() { for (auto &t : v) if (t.joinable()) t.join(); } } joiner{warm};
        if (n_dev_avail > 0) {
            warm.emplace_back([&] { pin.pin(rgb8, (size_t)w * h * 3); });
            for (uint32_t d = 0; d < n_dev; d++)
                warm.emplace_back(host_pipe_prewarm, device_of(d), w, std::min(tile_rows, h), (int)std::min<size_t>(3, share[d].size() / 2));
        }
        maray_tape *tape = nullptr;
        const int rc = maray_lower(s, nullptr, &tape);
        for (auto &t : warm) t.join();
        warm.clear();
        if (rc) return rc;
        auto fresh = std::make_shared<GenEntry>();
        fresh->key = key; fresh->tape = tape;
        entry = gen_cache_insert(fresh);
    } else pin.pin(rgb8, (size_t)w * h * 3);
    maray_program prog;
    maray_tape_program(entry->tape, &prog);

    maray_ctx_opts co;
    memset(&co, 0, sizeof co);
    co.backend = opts ? opts->backend : MARAY_BACKEND_AUTO;
    co.hint_mpixels = (uint32_t)std::min<uint64_t>(0xFFFFFFFFu, (((uint64_t)w * h / n_dev) >> 20) + 1);

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
        // worker d's context: the one an earlier call left (same program, same worker slot), or a new one
        const int dev = device_of(d);
        const IdleCtx kept = gen_cache_take(*entry, dev, co.backend == MARAY_BACKEND_AUTO, co.hint_mpixels);
        maray_ctx *ctx = kept.ctx;
        int r = ctx ? MARAY_OK : maray_hip_ctx_create(dev, &prog, tex, n_tex, &co, &ctx);
        if (!r && !share[d].empty())
            r = maray_hip_render_tiles(ctx, w, h, share[d].data(), (uint32_t)(share[d].size() / 2), rgb8, on_tile, &tu);
        const std::string msg = r ? maray_last_error() : "";      // this thread's message, re-raised on the calling thread
        if (r) maray_hip_ctx_free(ctx);                           // (a context that failed is not kept)
        else gen_cache_give(*entry, dev, ctx, kept.ctx ? kept.hint_mpixels : co.hint_mpixels);      // (what it was CHOSEN for)
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
