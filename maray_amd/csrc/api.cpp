// api.cpp — extern "C" boundary of libmaray_hip.so (product code).
// Every function converts maray::Error into a code + thread-local message and
// never lets an exception cross the ABI.
#include <pthread.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "backend.hpp"
#include "expr.hpp"
#include "lower.hpp"
#include "maray_hip.h"

using namespace maray;

struct maray_scene {
    Scene s;
    // 128-bit name of the scene's program: what maray_gen_to_image remembers a scene's tape and contexts under.  A scene read from
    // bytes is named by the hash of those bytes (without the header: the size is not part of a program); a rescale of a named
    // scene folds its factors into the name (load, rescale, render is what the CLI and an embedding do: re-encoding 88,000
    // nodes to name them cost the first call of chess 3-4 ms); every other call that changes the scene drops the name, and
    // the next use hashes the scene's encoding (what `save` would write).  Two names for one program are a miss in the
    // cache, nothing worse.
    std::mutex key_mutex;
    bool key_valid = false;
    uint64_t key[2] = {0, 0};
};
struct maray_tape { Tape t; };

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &m) { g_err = m; return code; }

// Deeply nested expressions recurse deeply; run the recursive passes on a
// thread with a large (lazily committed) stack.
struct BigStack {
    std::function<void()> fn;
    Error err{0, ""};
    bool threw = false, bad_alloc = false;
};
void *big_stack_main(void *p)
{
    BigStack *b = (BigStack *)p;
    try { b->fn(); }
    catch (const Error &e) { b->err = e; b->threw = true; }
    catch (const std::bad_alloc &) { b->bad_alloc = true; }
    catch (const std::exception &e) { b->err = Error{MARAY_E_INTERNAL, e.what()}; b->threw = true; }
    return nullptr;
}
void run_big_stack(std::function<void()> fn)
{
    BigStack b;
    b.fn = std::move(fn);
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, (size_t)1 << 30);
    pthread_t th;
    if (pthread_create(&th, &at, big_stack_main, &b) != 0) {
        pthread_attr_destroy(&at);
        big_stack_main(&b);
    } else {
        pthread_attr_destroy(&at);
        pthread_join(th, nullptr);
    }
    if (b.bad_alloc) throw Error{MARAY_E_INTERNAL, "out of memory"};
    if (b.threw) throw b.err;
}

template <typename F>
int guard(F f)
{
    try { f(); g_err.clear(); return MARAY_OK; }
    catch (const Error &e) { return fail(e.code, e.msg); }
    catch (const std::bad_alloc &) { return fail(MARAY_E_INTERNAL, "out of memory"); }
    catch (const std::exception &e) { return fail(MARAY_E_INTERNAL, e.what()); }
    catch (...) { return fail(MARAY_E_INTERNAL, "unknown error"); }
}

#define REQUIRE(c, what) do { if (!(c)) throw Error{MARAY_E_ARG, what}; } while (0)

}   // namespace

namespace maray {

void set_last_error(const std::string &m) { g_err = m; }

// Four independent 64-bit lanes, 32 bytes a step (a cache key, not a defence): a render call hashes its textures' bytes
// on every call -- content, not pointers: a caller may repaint a texture in place -- so the hash has to run at memory
// speed; one multiply chain per 8 bytes did 4 GB/s (1.5 ms for config 5's 6 MiB, more than the render), four chains
// side by side do ~4x that.
void hash128(const void *data, size_t n, uint64_t h[2])
{
    const unsigned char *p = (const unsigned char *)data;
    uint64_t a = h[0] ^ (n * 0x9e3779b97f4a7c15ull), b = h[1] + n, c = ~h[0] + 0x2545f4914f6cdd1dull, d = h[1] ^ 0xd6e8feb86659fd93ull;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        memcpy(w, p + i, 32);
        a = (a ^ w[0]) * 0xff51afd7ed558ccdull; a = (a << 29) | (a >> 35);
        b = (b + w[1]) * 0xc4ceb9fe1a85ec53ull; b ^= b >> 31;
        c = (c ^ w[2]) * 0x9fb21c651e98df25ull; c = (c << 31) | (c >> 33);
        d = (d + w[3]) * 0xd6e8feb86659fd93ull; d ^= d >> 29;
    }
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        a = (a ^ w) * 0xff51afd7ed558ccdull; a = (a << 29) | (a >> 35);
        b = (b + w) * 0xc4ceb9fe1a85ec53ull; b ^= b >> 31;
    }
    uint64_t w = 0;
    memcpy(&w, p + i, n - i);
    a = (a ^ w) * 0xff51afd7ed558ccdull; a ^= a >> 33;
    b = (b + w) * 0xc4ceb9fe1a85ec53ull; b ^= b >> 29;
    a ^= (c * 0x9e3779b97f4a7c15ull) ^ (c >> 32); b += (d * 0xff51afd7ed558ccdull) ^ (d >> 31);
    a *= 0x9e3779b97f4a7c15ull; a ^= a >> 32;
    h[0] = a * 0x9e3779b97f4a7c15ull; h[1] = b ^ a;
}

void scene_cache_key(const maray_scene *cs, uint64_t out[2])
{
    maray_scene *s = const_cast<maray_scene *>(cs);              // (the key is a cache inside the handle)
    std::lock_guard<std::mutex> lk(s->key_mutex);
    if (!s->key_valid) {
        std::vector<uint8_t> b;
        run_big_stack([&] { scene_encode(s->s, b); });
        // (the header's size is not part of a scene's program: maray_gen_to_image takes the size as arguments)
        s->key[0] = 0x6d61726179ull; s->key[1] = 3;
        hash128(b.data() + 8, b.size() >= 8 ? b.size() - 8 : 0, s->key);
        s->key_valid = true;
    }
    out[0] = s->key[0]; out[1] = s->key[1];
}

void validate_program(const maray_program &p)
{
    if (p.version != MARAY_TAPE_VERSION) throw Error{MARAY_E_ARG, "tape version mismatch"};
    if ((p.n_consts && !p.consts) || (p.n_row_ops && !p.row_ops) || (p.n_pix_ops && !p.pix_ops))
        throw Error{MARAY_E_ARG, "null section pointer"};
    if (p.n_row_slots > MARAY_MAX_SLOTS || p.n_pix_slots > MARAY_MAX_SLOTS) throw Error{MARAY_E_LIMIT, "too many slots"};
    const uint32_t n_ynum = numeric_yvals(p);       // y values past these only gate SKIP ops (guards)
    auto check = [&](const uint64_t *ops, uint32_t n, uint32_t n_slots, bool pixel) {
        std::vector<uint8_t> written(n_slots, 0);
        bool have_acc = false;
        struct Region { uint32_t end; std::vector<uint8_t> before; uint32_t dst; };
        std::vector<Region> regions;               // open SKIPZ / SKIPNZ regions, innermost last
        for (uint32_t i = 0; i < n; i++) {
            const uint64_t ins = ops[i];
            const uint32_t op = MARAY_INS_OP(ins), aux = MARAY_INS_AUX(ins), dst = MARAY_INS_DST(ins);
            if (op >= MARAY_OP_COUNT) throw Error{MARAY_E_ARG, "invalid opcode at op " + std::to_string(i)};
            auto close_regions = [&]() {
                // values written only inside a skipped region do not exist on the skip path
                while (!regions.empty() && regions.back().end == i) {
                    written = regions.back().before;
                    if (regions.back().dst != MARAY_DST_NONE) written[regions.back().dst] = 1;
                    if (dst != regions.back().dst) throw Error{MARAY_E_ARG, "skip region does not end in its own destination at op " + std::to_string(i)};
                    regions.pop_back();
                }
            };
            if (op == MARAY_OP_NOP) { close_regions(); continue; }
            const int arity = (op == MARAY_OP_TEXDIM) ? 0 : (op >= MARAY_OP_ADD && op <= MARAY_OP_APP) ? 2 : 1;   // STEPSIN, OUT, MOV, SKIP* and the unary ops read `a` only
            const uint32_t refs[2] = {MARAY_INS_A(ins), MARAY_INS_B(ins)};
            for (int k = 0; k < arity; k++) {
                const uint32_t kind = MARAY_REF_KIND(refs[k]), idx = MARAY_REF_INDEX(refs[k]);
                bool ok = true;
                switch (kind) {
                case MARAY_K_SLOT: ok = idx < n_slots && written[idx]; break;
                case MARAY_K_CONST: ok = idx < p.n_consts; break;
                case MARAY_K_YVAL: ok = pixel && idx < p.n_yvals; break;
                default: ok = idx <= MARAY_SPEC_YMIN && (idx != MARAY_SPEC_ACC || have_acc) && (idx != MARAY_SPEC_X || pixel) &&
                              (idx < MARAY_SPEC_XMAX || !pixel);      // the span specials are for the ROW section
                }
                if (!ok) throw Error{MARAY_E_ARG, "operand out of range or read before write at op " + std::to_string(i)};
            }
            if (op == MARAY_OP_OUT) {
                if (aux >= (pixel ? 3u : p.n_yvals)) throw Error{MARAY_E_ARG, "output index out of range at op " + std::to_string(i)};
                if (!regions.empty()) throw Error{MARAY_E_ARG, "OUT inside a skip region at op " + std::to_string(i)};
                continue;
            }
            if (op == MARAY_OP_SKIPZ || op == MARAY_OP_SKIPNZ) {
                if (aux == 0 || (uint64_t)i + aux >= n) throw Error{MARAY_E_ARG, "skip count out of range at op " + std::to_string(i)};
                if (!regions.empty() && i + aux > regions.back().end) throw Error{MARAY_E_ARG, "skip regions overlap at op " + std::to_string(i)};
                if (dst != MARAY_DST_NONE && dst >= n_slots) throw Error{MARAY_E_ARG, "dst slot out of range at op " + std::to_string(i)};
                // a guard (a y value nothing reads as an operand) is a bound, "0 => the boolean is 0 over the span": the
                // evaluators keep it as one bit (!= 0) or do not keep its value at all; it can gate a SKIPZ only
                if (op == MARAY_OP_SKIPNZ && MARAY_REF_KIND(refs[0]) == MARAY_K_YVAL && MARAY_REF_INDEX(refs[0]) >= n_ynum)
                    throw Error{MARAY_E_ARG, "SKIPNZ on a guard y value at op " + std::to_string(i) + " (guards gate SKIPZ only)"};
                const uint32_t last = MARAY_INS_OP(ops[i + aux]);
                if (!(last == MARAY_OP_MUL || last == MARAY_OP_MIN || last == MARAY_OP_MAX))
                    throw Error{MARAY_E_ARG, "skip region must end in Mul / Min / Max at op " + std::to_string(i)};
                regions.push_back(Region{i + aux, written, dst});
                continue;
            }
            if ((op == MARAY_OP_APP || op == MARAY_OP_TEXDIM) && aux >= p.n_app)
                throw Error{MARAY_E_ARG, "App id above n_app at op " + std::to_string(i)};
            if (dst != MARAY_DST_NONE) {
                if (dst >= n_slots) throw Error{MARAY_E_ARG, "dst slot out of range at op " + std::to_string(i)};
                written[dst] = 1;
            }
            have_acc = true;
            close_regions();
        }
        if (!regions.empty()) throw Error{MARAY_E_ARG, "unterminated skip region"};
    };
    check(p.row_ops, p.n_row_ops, p.n_row_slots, false);
    check(p.pix_ops, p.n_pix_ops, p.n_pix_slots, true);
}

}   // namespace maray

extern "C" {

const char *maray_last_error(void) { return g_err.c_str(); }
const char *maray_version(void) { return "maray_amd 0.1 (gfx950; tape v2; mirrors maray 0.3.8)"; }

// ---- scenes ------------------------------------------------------------------
static void name_from_bytes(maray_scene *s, const uint8_t *buf, size_t len)
{
    std::lock_guard<std::mutex> lk(s->key_mutex);
    s->key[0] = 0x6d61726179ull; s->key[1] = 3;
    hash128(buf + (len >= 8 ? 8 : len), len >= 8 ? len - 8 : 0, s->key);
    s->key_valid = true;
}

int maray_scene_from_bytes(const uint8_t *buf, size_t len, maray_scene **out)
{
    return guard([&] {
        REQUIRE(buf && out, "null argument");
        *out = nullptr;
        maray_scene *s = new maray_scene();
        try { run_big_stack([&] { scene_decode(buf, len, s->s); }); }
        catch (...) { delete s; throw; }
        name_from_bytes(s, buf, len);
        *out = s;
    });
}

int maray_scene_open(const char *path, maray_scene **out)
{
    return guard([&] {
        REQUIRE(path && out, "null argument");
        *out = nullptr;
        FILE *f = fopen(path, "rb");
        if (!f) throw Error{MARAY_E_IO, std::string("cannot open ") + path};
        std::vector<uint8_t> b;
        uint8_t tmp[65536];
        size_t n;
        while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) b.insert(b.end(), tmp, tmp + n);
        fclose(f);
        maray_scene *s = new maray_scene();
        try { run_big_stack([&] { scene_decode(b.data(), b.size(), s->s); }); }
        catch (...) { delete s; throw; }
        name_from_bytes(s, b.data(), b.size());
        *out = s;
    });
}

void maray_scene_free(maray_scene *s) { delete s; }

int maray_scene_size(const maray_scene *s, uint32_t *w, uint32_t *h)
{
    return guard([&] { REQUIRE(s && w && h, "null argument"); *w = s->s.w; *h = s->s.h; });
}

int maray_scene_set_size(maray_scene *s, uint32_t w, uint32_t h)
{
    return guard([&] { REQUIRE(s, "null argument"); s->key_valid = false; s->s.w = w; s->s.h = h; });
}

int maray_scene_is_legacy(const maray_scene *s, int *legacy)
{
    return guard([&] { REQUIRE(s && legacy, "null argument"); *legacy = s->s.legacy ? 1 : 0; });
}

int maray_scene_node_count(const maray_scene *s, int c, uint64_t *n)
{
    return guard([&] {
        REQUIRE(s && n && c >= 0 && c < 3, "bad argument");
        run_big_stack([&] { *n = scene_node_count(s->s, c); });
    });
}

int maray_scene_encode(const maray_scene *s, uint8_t *out, size_t cap, size_t *len_out)
{
    return guard([&] {
        REQUIRE(s && len_out, "null argument");
        std::vector<uint8_t> b;
        run_big_stack([&] { scene_encode(s->s, b); });
        *len_out = b.size();
        if (out && cap >= b.size()) memcpy(out, b.data(), b.size());
    });
}

int maray_scene_save(const maray_scene *s, const char *path)
{
    return guard([&] {
        REQUIRE(s && path, "null argument");
        std::vector<uint8_t> b;
        run_big_stack([&] { scene_encode(s->s, b); });
        FILE *f = fopen(path, "wb");
        if (!f) throw Error{MARAY_E_IO, std::string("cannot create ") + path};
        size_t n = fwrite(b.data(), 1, b.size(), f);
        fclose(f);
        if (n != b.size()) throw Error{MARAY_E_IO, "short write"};
    });
}

int maray_scene_fix_color(maray_scene *s)
{
    return guard([&] { REQUIRE(s, "null argument"); s->key_valid = false; run_big_stack([&] { scene_fix_color(s->s); }); });
}

int maray_scene_rescale(maray_scene *s, uint32_t sx, uint32_t sy)
{
    return guard([&] {
        REQUIRE(s, "null argument");
        scene_rescale(s->s, sx, sy);
        std::lock_guard<std::mutex> lk(s->key_mutex);
        if (s->key_valid) { const uint32_t step[3] = {0x52455343u, sx, sy}; hash128(step, sizeof step, s->key); }      // "RESC", factors
    });
}

int maray_scene_simplify(maray_scene *s)
{
    return guard([&] { REQUIRE(s, "null argument"); s->key_valid = false; run_big_stack([&] { scene_simplify(s->s); }); });
}

int maray_scene_simplify_ex(maray_scene *s, uint32_t flags)
{
    return guard([&] {
        REQUIRE(s, "null argument"); s->key_valid = false;
        REQUIRE((flags & ~(uint32_t)MARAY_SIMPLIFY_MERGE_DIVISORS) == 0, "unknown simplify flag");
        run_big_stack([&] { scene_simplify(s->s, flags); });
    });
}

int maray_scene_compress(maray_scene *s, uint32_t *n_vars3)
{
    return guard([&] { REQUIRE(s, "null argument"); s->key_valid = false; run_big_stack([&] { scene_compress(s->s, n_vars3); }); });
}

int maray_scene_display_len(maray_scene *s, int c, uint64_t *len)
{
    return guard([&] {
        REQUIRE(s && len && c >= 0 && c < 3, "bad argument"); s->key_valid = false;
        run_big_stack([&] { *len = scene_display_len(s->s, c); });
    });
}

// ---- lowering -----------------------------------------------------------------
int maray_lower(const maray_scene *s, const maray_lower_opts *opts, maray_tape **out)
{
    return guard([&] {
        REQUIRE(s && out, "null argument");
        *out = nullptr;
        maray_lower_opts o;
        memset(&o, 0, sizeof o);
        o.hoist_rows = 1;
        if (opts) o = *opts;
        maray_tape *t = new maray_tape();
        try {
            run_big_stack([&] { lower_scene(s->s, o, t->t); });
            validate_program(t->t.program());
        } catch (...) { delete t; throw; }
        *out = t;
    });
}

void maray_tape_free(maray_tape *t) { delete t; }

int maray_tape_program(const maray_tape *t, maray_program *out)
{
    return guard([&] { REQUIRE(t && out, "null argument"); *out = t->t.program(); });
}

int maray_tape_get_info(const maray_tape *t, maray_tape_info *out)
{
    return guard([&] { REQUIRE(t && out, "null argument"); *out = t->t.info; });
}

// ---- device ---------------------------------------------------------------------
int maray_hip_device_count(int *n)
{
    return guard([&] { REQUIRE(n, "null argument"); *n = hip_device_count(); });
}

int maray_hip_ctx_create(int device, const maray_program *prog, const maray_texture *tex, uint32_t n_tex,
                         const maray_ctx_opts *opts, maray_ctx **out)
{
    return guard([&] {
        REQUIRE(prog && out, "null argument");
        REQUIRE(n_tex == 0 || tex, "null texture table");
        *out = nullptr;
        validate_program(*prog);
        if (prog->n_app > 5u * n_tex)   // reference: rt.functions[id] panics (src/lib.rs:665)
            throw Error{MARAY_E_APP_RANGE, "scene calls App id " + std::to_string(prog->n_app - 1) + " but only " +
                        std::to_string(5u * n_tex) + " texture functions exist"};
        for (uint32_t i = 0; i < n_tex; i++) REQUIRE(tex[i].rgb || (uint64_t)tex[i].w * tex[i].h == 0, "null texture raster");
        const uint32_t backend = opts ? opts->backend : MARAY_BACKEND_TAPE;
        Backend *b = nullptr;
        switch (backend) {
        case MARAY_BACKEND_TAPE: b = make_tape_backend(device, *prog, tex, n_tex, true); break;
        case MARAY_BACKEND_TAPE_SMEM: b = make_tape_backend(device, *prog, tex, n_tex, false); break;
        case MARAY_BACKEND_JIT: b = make_jit_backend(device, *prog, tex, n_tex); break;
        case MARAY_BACKEND_AUTO: {
            // One-shot economics.  The specialised kernels render a frame tens of times faster than the interpreter, but
            // hiprtc needs seconds to build them (chess: 9 k pixel + 17 k row ops -> 1.6 s with the two modules built side
            // by side; 1,000 triangles, 36 k + 37 k ops -> 15-20 s; before the lowering privatised shared values a 49 k-op
            // scene took 6.5 min), the interpreter needs none.  So: take them when the code objects are already in the cache, or when the
            // caller says how much it will render (hint_mpixels) and the interpreter would need longer for that than
            // the build takes, or when it does not say (a context kept for many frames).  Estimates, measured on chess
            // and the triangle soups: build 0.5 s + 0.4 ms per pixel op + 0.24 ms per row op;
            // interpreter 4.4 us per megapixel and executed op, of which guards skip ~95 % when the program has any.
            const char *force = getenv("MARAY_AUTO");          // "jit" / "tape": override (measurements)
            bool want_jit = prog->n_pix_ops <= 25000;
            if (want_jit && opts && opts->hint_mpixels) {
                // cached code objects cost a context ~0.02 s more than the interpreter's (two module loads; the key comes from the
                // program's name, jit_build.cpp) -- 0.2 s when the process has to generate the sources for the key first
                const double build_s = jit_code_is_cached(*prog) ? 0.02 : 0.5 + 4.0e-4 * prog->n_pix_ops + 2.4e-4 * prog->n_row_ops;
                const bool guarded = prog->n_yvals > numeric_yvals(*prog);
                const double interp_s = (double)opts->hint_mpixels * prog->n_pix_ops * 4.4e-6 * (guarded ? 0.05 : 1.0);
                want_jit = interp_s > build_s;
            }
            if (force && !strcmp(force, "jit")) want_jit = true;
            if (force && !strcmp(force, "tape")) want_jit = false;
            if (!want_jit) { b = make_tape_backend(device, *prog, tex, n_tex, false); break; }
            try { b = make_jit_backend(device, *prog, tex, n_tex); }
            catch (const Error &e) {
                if (e.code == MARAY_E_NO_DEVICE) throw;
                b = make_tape_backend(device, *prog, tex, n_tex, false);   // still the HIP path, never a CPU fallback
            }
            break;
        }
        default: throw Error{MARAY_E_ARG, "unknown backend"};
        }
        maray_ctx *c = new maray_ctx();
        c->backend = b;
        c->n_tex = n_tex;
        *out = c;
    });
}

void maray_hip_ctx_free(maray_ctx *c)
{
    if (!c) return;
    delete c->backend;
    delete c;
}

static void check_rows(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1)
{
    if (y0 > y1 || y1 > h) throw Error{MARAY_E_ARG, "row range out of bounds"};
    // the lowering's interval analysis (bounded Sin arguments, row bounds) covers x, y < MARAY_DOMAIN_MAX
    if (w > MARAY_DOMAIN_MAX || y1 > MARAY_DOMAIN_MAX)
        throw Error{MARAY_E_LIMIT, "image exceeds " + std::to_string(MARAY_DOMAIN_MAX) + " pixels in x or y"};
}

int maray_hip_render_rows(maray_ctx *c, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint8_t *rgb8, double *rgb64)
{
    return guard([&] {
        REQUIRE(c && c->backend, "null context");
        check_rows(w, h, y0, y1);
        if (y0 == y1 || w == 0 || (!rgb8 && !rgb64)) return;
        c->backend->render_host_tiles(w, h, cut_row_tiles(w, y0, y1, rgb8 != nullptr, rgb64 != nullptr), y0, rgb8, rgb64, nullptr);
    });
}

int maray_hip_render_tiles(maray_ctx *c, uint32_t w, uint32_t h, const uint32_t *tiles_y0y1, uint32_t n_tiles,
                           uint8_t *rgb8_image, maray_tile_fn fn, void *user)
{
    return guard([&] {
        REQUIRE(c && c->backend, "null context");
        REQUIRE(n_tiles == 0 || tiles_y0y1, "null tile list");
        if (n_tiles == 0 || w == 0 || !rgb8_image) return;
        std::vector<RowTile> tiles(n_tiles);
        for (uint32_t i = 0; i < n_tiles; i++) {
            tiles[i] = RowTile{tiles_y0y1[2 * i], tiles_y0y1[2 * i + 1]};
            check_rows(w, h, tiles[i].y0, tiles[i].y1);
            REQUIRE(tiles[i].y1 > tiles[i].y0, "empty tile");
        }
        std::function<void(uint32_t, uint32_t)> done;
        if (fn) done = [&](uint32_t a, uint32_t b) { fn(user, a, b); };
        c->backend->render_host_tiles(w, h, tiles, 0, rgb8_image, nullptr, done);
    });
}

int maray_host_alloc(size_t bytes, void **out)
{
    return guard([&] {
        REQUIRE(out, "null argument");
        *out = host_alloc_pinned(bytes);
    });
}

void maray_host_free(void *p) { host_free_pinned(p); }

int maray_host_register(void *p, size_t bytes)
{
    return guard([&] { REQUIRE(p && bytes, "null argument"); host_register(p, bytes); });
}

int maray_host_unregister(void *p)
{
    return guard([&] { REQUIRE(p, "null argument"); host_unregister(p); });
}

int maray_hip_render_rows_device(maray_ctx *c, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                                 void *d_rgb8, void *d_rgb64, void *stream)
{
    return guard([&] {
        REQUIRE(c && c->backend, "null context");
        check_rows(w, h, y0, y1);
        if (y0 == y1 || w == 0) return;
        c->backend->render_device(w, h, RowBlocks::range(y0, y1), d_rgb8, d_rgb64, stream);
    });
}

int maray_hip_render_blocks_device(maray_ctx *c, uint32_t w, uint32_t h, uint32_t y0, uint32_t block_rows, uint32_t block_stride,
                                   uint32_t n_blocks, void *d_rgb8, void *d_rgb64, void *stream)
{
    return guard([&] {
        REQUIRE(c && c->backend, "null context");
        if (n_blocks == 0 || block_rows == 0 || w == 0) return;
        REQUIRE(n_blocks == 1 || block_stride >= block_rows, "row blocks overlap: block_stride < block_rows");
        const uint64_t last = (uint64_t)y0 + (uint64_t)(n_blocks - 1) * block_stride + block_rows;       // one past the last row
        REQUIRE(last <= 0xFFFFFFFFull && (uint64_t)n_blocks * block_rows <= 0xFFFFFFFFull, "row blocks out of range");
        check_rows(w, h, y0, (uint32_t)last);
        c->backend->render_device(w, h, RowBlocks{y0, n_blocks * block_rows, block_rows, n_blocks == 1 ? 0u : block_stride}, d_rgb8, d_rgb64, stream);
    });
}

int maray_hip_time_rows(maray_ctx *c, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                        void *d_rgb8, void *d_rgb64, int reps, float *ms_avg)
{
    return guard([&] {
        REQUIRE(c && c->backend && ms_avg, "null argument");
        check_rows(w, h, y0, y1);
        REQUIRE(y1 > y0 && w > 0 && reps > 0, "empty launch");
        *ms_avg = c->backend->time_rows(w, h, RowBlocks::range(y0, y1), d_rgb8, d_rgb64, reps);
    });
}

int maray_hip_time_blocks(maray_ctx *c, uint32_t w, uint32_t h, uint32_t y0, uint32_t block_rows, uint32_t block_stride,
                          uint32_t n_blocks, void *d_rgb8, void *d_rgb64, int reps, float *ms_avg)
{
    return guard([&] {
        REQUIRE(c && c->backend && ms_avg, "null argument");
        REQUIRE(n_blocks > 0 && block_rows > 0 && w > 0 && reps > 0, "empty launch");
        REQUIRE(n_blocks == 1 || block_stride >= block_rows, "row blocks overlap: block_stride < block_rows");
        const uint64_t last = (uint64_t)y0 + (uint64_t)(n_blocks - 1) * block_stride + block_rows;
        REQUIRE(last <= 0xFFFFFFFFull && (uint64_t)n_blocks * block_rows <= 0xFFFFFFFFull, "row blocks out of range");
        check_rows(w, h, y0, (uint32_t)last);
        *ms_avg = c->backend->time_rows(w, h, RowBlocks{y0, n_blocks * block_rows, block_rows, n_blocks == 1 ? 0u : block_stride}, d_rgb8, d_rgb64, reps);
    });
}

int maray_jit_source(const maray_program *prog, char **src_out)
{
    return guard([&] {
        REQUIRE(prog && src_out, "null argument");
        const std::string s = jit_source(*prog);
        *src_out = (char *)malloc(s.size() + 1);
        if (!*src_out) throw Error{MARAY_E_INTERNAL, "out of memory"};
        memcpy(*src_out, s.c_str(), s.size() + 1);
    });
}

int maray_jit_source_rows(const maray_program *prog, char **src_out, uint32_t *n_chunks)
{
    return guard([&] {
        REQUIRE(prog && src_out, "null argument");
        const std::string s = jit_source_rows(*prog, n_chunks);
        *src_out = (char *)malloc(s.size() + 1);
        if (!*src_out) throw Error{MARAY_E_INTERNAL, "out of memory"};
        memcpy(*src_out, s.c_str(), s.size() + 1);
    });
}

int maray_row_cone(const maray_program *prog, uint32_t first_out, uint32_t n_out, uint64_t **ops_out, uint32_t *n_ops_out, uint32_t *n_slots_out)
{
    return guard([&] {
        REQUIRE(prog && ops_out && n_ops_out && n_slots_out, "null argument");
        validate_program(*prog);
        const RowTapeDeps deps = row_tape_deps(*prog);
        std::vector<uint32_t> outs;
        for (uint32_t o : deps.outs) {
            const uint32_t k = MARAY_INS_AUX(prog->row_ops[o]);
            if (k >= first_out && k - first_out < n_out) outs.push_back(o);
        }
        std::vector<uint64_t> t = compact_tape(row_tape_cone(*prog, deps, outs, nullptr));
        *n_slots_out = getenv("MARAY_TAPE_KEEP_ORDER") ? renumber_slots(t) : reschedule_tape(t);
        *n_ops_out = (uint32_t)t.size();
        *ops_out = (uint64_t *)malloc(t.size() * 8 + 8);
        if (!*ops_out) throw Error{MARAY_E_INTERNAL, "out of memory"};
        memcpy(*ops_out, t.data(), t.size() * 8);
    });
}

int maray_jit_build(const maray_program *prog, void **code_out, size_t *len_out)
{
    return guard([&] {
        REQUIRE(prog && code_out && len_out, "null argument");
        // both kernels, through the code object cache (this is also how a cache is warmed ahead of a render)
        const std::vector<char> &code = jit_code_for(*prog)->pix;
        *code_out = malloc(code.size() ? code.size() : 1);
        if (!*code_out) throw Error{MARAY_E_INTERNAL, "out of memory"};
        memcpy(*code_out, code.data(), code.size());
        *len_out = code.size();
    });
}

int maray_jit_code_key(const maray_program *prog, char *out33)
{
    return guard([&] {
        REQUIRE(prog && out33, "null argument");
        const std::string k = jit_code_key(*prog);
        memcpy(out33, k.c_str(), std::min<size_t>(k.size(), 32) + 1);
        out33[32] = 0;
    });
}

int maray_jit_code_cached(const maray_program *prog, int *cached)
{
    return guard([&] { REQUIRE(prog && cached, "null argument"); *cached = jit_code_is_cached(*prog) ? 1 : 0; });
}

const char *maray_hip_kernel_name(const maray_ctx *c) { return (c && c->backend) ? c->backend->kernel_name() : ""; }

void maray_free(void *p) { free(p); }

}   // extern "C"
