/*
 * maray_hip.h — C ABI of libmaray_hip.so: the MI355X (gfx950) per-pixel
 * expression evaluator that drops in behind maray's render surface.
 *
 * Every entry point cites the reference interface (advancedresearch/maray
 * 0.3.8) it replaces or mirrors.  Conventions:
 *   - plain pointers and sizes only; opaque handles are owned by the library
 *     and released with the matching *_free;
 *   - output buffers are owned and sized by the caller; inputs are borrowed
 *     for the duration of the call only;
 *   - every function returns 0 on success or a negative MARAY_E_* code and
 *     never throws/aborts across the boundary; maray_last_error() returns a
 *     thread-local message;
 *   - calls are blocking; a maray_ctx is bound to one device and is not
 *     thread-safe (multi-GPU = one ctx per device);
 *   - there is NO CPU fallback: the render entry points fail with
 *     MARAY_E_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef MARAY_HIP_H
#define MARAY_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "maray_tape.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    MARAY_OK = 0,
    MARAY_E_ARG = -1,        /* bad argument */
    MARAY_E_IO = -2,         /* file error (open/save return anyhow::Error) */
    MARAY_E_DECODE = -3,     /* bincode decode error */
    MARAY_E_ALIASED = -4,    /* scene depends on Cache id-aliasing (reference result is order dependent) */
    MARAY_E_CYCLE = -5,      /* Let variable defined in terms of itself (reference overflows its stack) */
    MARAY_E_APP_RANGE = -6,  /* App id outside Runtime.functions (reference panics, src/lib.rs:665) */
    MARAY_E_LIMIT = -7,      /* tape limits exceeded (slots / constants / y values) */
    MARAY_E_NO_DEVICE = -8,  /* no usable gfx950 device, or HIP runtime error */
    MARAY_E_HIP = -9,        /* HIP runtime / hiprtc error */
    MARAY_E_INTERNAL = -10
};

const char *maray_last_error(void);
const char *maray_version(void);

/* ---- scene files ------------------------------------------------------------
 * `open` (src/lib.rs:1227-1235) / `save` (src/lib.rs:1216-1224): bincode 1.3.3
 * default options of `([u32;2], [Expr;3])`.  The reader auto-detects the
 * current tag numbering (src/lib.rs:101-149) and the legacy one used by
 * data/chess.maray (no Arc variant; tags one lower): it accepts the numbering
 * that consumes the buffer exactly. */
typedef struct maray_scene maray_scene;

int maray_scene_open(const char *path, maray_scene **out);
int maray_scene_from_bytes(const uint8_t *buf, size_t len, maray_scene **out);
void maray_scene_free(maray_scene *s);
int maray_scene_size(const maray_scene *s, uint32_t *w, uint32_t *h);
int maray_scene_set_size(maray_scene *s, uint32_t w, uint32_t h);
int maray_scene_is_legacy(const maray_scene *s, int *legacy);
/* Tree node count of channel c (definitions of Let variables counted once). */
int maray_scene_node_count(const maray_scene *s, int c, uint64_t *n);
/* `save`: encode in the current numbering.  *len_out = bytes needed; writes
 * only if cap is large enough. */
int maray_scene_encode(const maray_scene *s, uint8_t *out, size_t cap, size_t *len_out);
int maray_scene_save(const maray_scene *s, const char *path);
/* var_fixer::fix_color (src/var_fixer.rs:74-82).  maray_lower applies it
 * itself; exposed so the pre-pass can be tested on its own. */
int maray_scene_fix_color(maray_scene *s);
/* `Expr::scale`-style resampling (src/lib.rs:804-806) applied to the whole
 * scene including Let definitions: X -> X * (1/sx), Y -> Y * (1/sy), and the
 * header size multiplied by (sx, sy).  Exact when sx, sy are powers of two. */
int maray_scene_rescale(maray_scene *s, uint32_t sx, uint32_t sy);
/* `Expr::simplify` on each channel (src/lib.rs:601-604: constant_reduction at the root, src/constant_reduction.rs:9-185,
 * then the rewrite rules of src/simplify.rs:129-327).  Authoring-time, not on the render path; the reference applies it
 * before `save` (examples/chess.rs:43). */
int maray_scene_simplify(maray_scene *s);
/* The same with options.  MARAY_SIMPLIFY_MERGE_DIVISORS: one rewrite the reference does not have, `(a/p) * 1/q -> a / (p*q)`.
 * The reference's rules do not terminate on such a term -- src/simplify.rs:276-283 turns `(a/p) * 1/q` into `(a * 1/q) / p`,
 * whose numerator is the quotient a/q again, and the two divisors change places for ever (the reference overflows its
 * stack; maray_scene_simplify reports MARAY_E_LIMIT) -- and examples/chess.rs:43 builds such terms from every grid cell,
 * so the current revision cannot regenerate its own data/chess.maray (written by an older one: legacy tags, and naturals
 * that are products of such divisors).  With the flag the example's pipeline runs. */
enum { MARAY_SIMPLIFY_MERGE_DIVISORS = 1 };
int maray_scene_simplify_ex(maray_scene *s, uint32_t flags);
/* `Expr::compress` on each channel (src/lib.rs:610-614: `flatten`, src/compressor.rs:167-211, then the greedy
 * Let-introducing loop of `compress`, :214-236, driven by the printed length of every candidate term exactly as the
 * reference's `Display` prints it).  Changes no value.  n_vars3 (may be NULL): variables introduced per channel. */
int maray_scene_compress(maray_scene *s, uint32_t *n_vars3);
/* `format!("{}", color[c]).chars().count()` (impl Display for Expr, src/lib.rs:196-366): what compress weighs. */
int maray_scene_display_len(maray_scene *s, int c, uint64_t *len);

/* ---- lowering: Expr -> tape ---------------------------------------------------
 * Replaces what the reference does per pixel in `Expr::eval2` + `Cache`
 * (src/lib.rs:623-670, src/cache.rs) with a one-time pass: fix_color, inline
 * Let/Var/Arc/Decor into one hash-consed DAG over R, G and B, fold constant
 * sub-trees with IEEE-exact ops only (never sin/exp/ln), hoist Y-only
 * sub-expressions into the ROW section, schedule, allocate value slots. */
typedef struct maray_tape maray_tape;

typedef struct maray_lower_opts {
    uint32_t hoist_rows;   /* 1 = hoist Y-only sub-expressions into the ROW section (default), 0 = off */
    uint32_t plain_cse;    /* 1 = hash-cons without commutative operand canonicalisation (reproduces the
                              op census of SURVEY.md §8(d)); 0 = canonicalise a+b/b+a etc. (default) */
    uint32_t no_fuse;      /* 1 = keep Step(Sin(a)) as two ops; 0 = fuse into MARAY_OP_STEPSIN (default) */
    uint32_t no_skips;     /* 1 = do not emit SKIPZ / SKIPNZ wave-level short circuits; 0 = emit them (default) */
    uint32_t no_row_guards; /* 1 = no row-level SKIPZ ops (guards that are y values: bounds of a boolean over a whole row) */
    uint32_t no_private_regions; /* 1 = row regions share hash-consed values with the rest of the tape (computed ahead of the
                              SKIP op, unconditionally); 0 = each row region re-derives the x-dependent values it reads (default) */
    uint32_t no_rebalance; /* 1 = keep chains of one boolean connective (max(t1, max(t2, ...))) as written; 0 = rebuild them as
                              balanced trees, whose sub-trees get row-level SKIP ops of their own (default) */
    uint32_t no_y_spans;   /* 1 = guards bound a boolean over a span of x on one row only (they read Y); 0 = guards monotone in y
                              too are bounded over the rows [YMIN, YMAX] as well, i.e. over a rectangle of pixels (default) */
} maray_lower_opts;

typedef struct maray_tape_info {
    uint32_t n_consts, n_row_ops, n_row_slots, n_yvals, n_pix_ops, n_pix_slots, n_app;
    /* algorithmic op counts (unique non-constant ops after hash-consing and
     * constant folding): total and by dependence class */
    uint32_t alg_ops, alg_ops_xy, alg_ops_x, alg_ops_y, alg_ops_uniform;
    uint32_t folded_ops;       /* constant ops folded on the host */
    uint32_t dag_nodes;        /* unique DAG nodes including leaves */
    uint32_t acc_operands;     /* operand reads served by ACC */
    uint32_t skip_ops;         /* SKIPZ / SKIPNZ ops in the PIXEL section */
    uint32_t bool_ops;         /* PIXEL ops whose value is provably +0.0 or 1.0 */
    uint32_t sin_ops, sin_bounded;   /* Sin/StepSin ops, and how many have a proven-bounded argument */
    uint32_t private_regions;  /* row regions that got private copies of the shared x-dependent values they read */
    uint32_t rebalanced_chains; /* boolean OR / AND chains rebuilt as balanced trees */
    uint32_t op_histogram[MARAY_OP_COUNT];   /* PIXEL section */
} maray_tape_info;

int maray_lower(const maray_scene *s, const maray_lower_opts *opts, maray_tape **out);
void maray_tape_free(maray_tape *t);
int maray_tape_program(const maray_tape *t, maray_program *out);   /* pointers valid until maray_tape_free */
int maray_tape_get_info(const maray_tape *t, maray_tape_info *out);

/* ---- tape-level device ABI (what a Rust `RenderMethod::Hip` arm binds) --------
 * Replaces par_gen_to_image / wasm_par_gen_to_image (src/render.rs:35-192) and
 * the wasmer JIT (src/wasm.rs:136-158). */
typedef struct maray_ctx maray_ctx;

typedef struct maray_texture {   /* one `RgbImage` of textures::Textures (src/textures.rs:9-12) */
    const uint8_t *rgb;          /* interleaved RGB8, row-major, w*h*3 bytes */
    uint32_t w, h;
} maray_texture;

enum {
    MARAY_BACKEND_TAPE = 0,      /* tape interpreter kernel: tape + constants staged in LDS */
    MARAY_BACKEND_TAPE_SMEM = 1, /* tape interpreter kernel: tape streamed through the scalar cache */
    MARAY_BACKEND_JIT = 2,       /* tape specialised to straight-line HIP via hiprtc (GPU analogue of src/wasm.rs) */
    MARAY_BACKEND_AUTO = 3       /* the specialised kernels when their code objects are in the cache (MARAY_CACHE_DIR) or when
                                    building them (estimated from the tape's size) costs less than what they save over
                                    hint_mpixels of rendering; else the scalar-cache tape interpreter, which needs no build */
};

typedef struct maray_ctx_opts {
    uint32_t backend;       /* MARAY_BACKEND_* */
    uint32_t hint_mpixels;  /* MARAY_BACKEND_AUTO: how many megapixels (2^20 pixels, rounded up) the caller is about to render with
                               this context; 0 = unknown / many.  The specialised kernels are worth their hiprtc build only
                               when that time is earned back (or when the code object cache already holds them). */
    uint32_t reserved[6];
} maray_ctx_opts;

int maray_hip_device_count(int *n);
/* Uploads tape, constants and textures to HBM once. */
int maray_hip_ctx_create(int device, const maray_program *prog, const maray_texture *tex, uint32_t n_tex,
                         const maray_ctx_opts *opts, maray_ctx **out);
void maray_hip_ctx_free(maray_ctx *c);

/* Evaluate rows [y0, y1) of a w x h image: p = [x as f64, y as f64],
 * x in [0,w) (src/render.rs:88-95).  rgb8: (y1-y0)*w*3 bytes, interleaved RGB,
 * row-major, values cast like Rust `as u8` (saturating, NaN -> 0); rgb64:
 * (y1-y0)*w*3 doubles (the pre-cast f64 values), either may be NULL.
 * Host-pointer form (device -> host copy included). */
int maray_hip_render_rows(maray_ctx *c, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                          uint8_t *rgb8, double *rgb64);
/* Several row ranges ("tiles", rows [y0,y1) each, given as n_tiles pairs y0,y1), rendered in the given order into ONE
 * caller-owned RGB8 raster of the whole image (rgb8_image = first byte of image row 0; w*h*3 bytes).  This is what
 * one worker of a multi-GPU render calls for its share of the image (the host-side gather of SURVEY.md 8(e), the
 * collector thread of src/render.rs:54-83): tile k's device -> host copy runs under tile k+1's kernels, the call
 * joins at the end.  fn (may be NULL) is called on the calling thread after each tile's rows have landed.
 * A raster in pinned memory (maray_host_alloc / maray_host_register) is written by DMA directly at PCIe rate; a
 * pageable one is filled through a pinned staging ring of the context (an extra host copy). */
typedef void (*maray_tile_fn)(void *user, uint32_t y0, uint32_t y1);
int maray_hip_render_tiles(maray_ctx *c, uint32_t w, uint32_t h, const uint32_t *tiles_y0y1, uint32_t n_tiles,
                           uint8_t *rgb8_image, maray_tile_fn fn, void *user);
/* Pinned (page-locked, DMA-able from every device) host memory for output rasters: what backs the `RgbImage` a
 * caller hands to gen_to_image (src/lib.rs:1210) when it wants the PCIe rate.  maray_host_register pins memory the
 * caller already owns (e.g. a Rust Vec<u8>) for the time between the two calls. */
int maray_host_alloc(size_t bytes, void **out);
void maray_host_free(void *p);
int maray_host_register(void *p, size_t bytes);
int maray_host_unregister(void *p);
/* Device-pointer form: outputs stay in HBM; enqueued on `stream`
 * (a hipStream_t, NULL = the null stream); returns without synchronising. */
int maray_hip_render_rows_device(maray_ctx *c, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                                 void *d_rgb8, void *d_rgb64, void *stream);
/* Several row ranges in one launch: n_blocks blocks of block_rows consecutive rows, the first at y0, block_stride
 * rows apart (block_stride >= block_rows); outputs packed block after block.  This is how one rank of a multi-GPU
 * render takes its interleaved share of an image (blocks dealt round-robin, src/render.rs:35-99 deals rows the same
 * way to Rayon workers) without paying a launch per block. */
int maray_hip_render_blocks_device(maray_ctx *c, uint32_t w, uint32_t h, uint32_t y0, uint32_t block_rows, uint32_t block_stride,
                                   uint32_t n_blocks, void *d_rgb8, void *d_rgb64, void *stream);
/* Time `reps` launches of the pixel kernel for rows [y0,y1) with HIP events on
 * the launch stream; returns the average milliseconds per launch. */
int maray_hip_time_rows(maray_ctx *c, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                        void *d_rgb8, void *d_rgb64, int reps, float *ms_avg);
/* The same for the launch maray_hip_render_blocks_device issues (one rank's interleaved share of an image). */
int maray_hip_time_blocks(maray_ctx *c, uint32_t w, uint32_t h, uint32_t y0, uint32_t block_rows, uint32_t block_stride,
                          uint32_t n_blocks, void *d_rgb8, void *d_rgb64, int reps, float *ms_avg);
/* The part of a program's ROW section that y values [first_out, first_out + n_out) depend on, as the evaluators cut
 * it to run parts of the section side by side or per rectangle of pixels: a tape of its own (NOPs removed, SKIP
 * regions kept, value slots renumbered by liveness; *n_slots_out = slots it uses).  Free *ops_out with maray_free. */
int maray_row_cone(const maray_program *prog, uint32_t first_out, uint32_t n_out, uint64_t **ops_out, uint32_t *n_ops_out,
                   uint32_t *n_slots_out);
/* MARAY_BACKEND_JIT, offline: the HIP source generated for a tape (free with
 * maray_free) and the gfx950 code object hiprtc builds from it (needs no GPU). */
int maray_jit_source(const maray_program *prog, char **src_out);
/* ... and of the ROW kernel (one work-item per row; *n_chunks = how many independent chunks the
 * ROW section was cut into, evaluated side by side as blockIdx.y; may be NULL). */
int maray_jit_source_rows(const maray_program *prog, char **src_out, uint32_t *n_chunks);
int maray_jit_build(const maray_program *prog, void **code_out, size_t *len_out);
/* The specialised kernels of a program are built once and kept, in the process and under MARAY_CACHE_DIR (default
 * $XDG_CACHE_HOME/maray_amd or ~/.cache/maray_amd; "off" disables): the reference's JIT compiles its modules again on
 * every thread of every render (src/render.rs:158-165).  The key is a 128-bit hash of the generated sources, the
 * compiler options and the hiprtc version: 32 hex digits + NUL into out33.  Needs no GPU. */
int maray_jit_code_key(const maray_program *prog, char *out33);
int maray_jit_code_cached(const maray_program *prog, int *cached);
/* Name of the dominant kernel (for matching rocprofv3 rows). */
const char *maray_hip_kernel_name(const maray_ctx *c);

/* ---- render façade: gen_to_image / gen (src/lib.rs:1177-1213) -----------------*/
enum {                         /* RenderMethod (src/lib.rs:1155-1173), extended */
    MARAY_METHOD_HIP = 3       /* 0..2 are the reference's CPU methods, not provided here */
};
enum { MARAY_REPORT_NONE = 0, MARAY_REPORT_ROW = 1, MARAY_REPORT_DURATION_MS = 2 };   /* Report (src/report.rs:17-24) */

typedef struct maray_report {
    uint32_t kind;    /* MARAY_REPORT_* */
    uint32_t value;   /* rows, or milliseconds */
} maray_report;

/* F: Fn(&mut RgbImage, f64) (src/lib.rs:1185): image so far + progress y/h.
 * Invoked on the calling thread between row tiles. */
typedef void (*maray_report_fn)(void *user, uint8_t *rgb8, uint32_t w, uint32_t h, double progress);

typedef struct maray_gen_opts {
    uint32_t backend;        /* MARAY_BACKEND_* */
    uint32_t n_devices;      /* 0 = all visible devices; image rows are tiled across them */
    uint32_t tile_rows;      /* rows per launch (0 = default) */
    uint32_t reserved[5];
} maray_gen_opts;

/* gen_to_image: fills the caller's w*h*3 RGB8 buffer. */
int maray_gen_to_image(const maray_scene *s, const maray_texture *tex, uint32_t n_tex,
                       const maray_gen_opts *opts, maray_report report, maray_report_fn fn, void *user,
                       uint8_t *rgb8, uint32_t w, uint32_t h);
/* gen_to_image keeps what a call sets up -- the scene's tape and one context per device -- for the next call with the
 * same scene, textures and back-end (the last MARAY_GEN_CACHE programs, default 4, 0 = none): an animation that calls
 * it in a loop (examples/test*.rs) pays the lowering and the context creation once, not per frame.  This frees them. */
void maray_gen_cache_clear(void);
/* What is kept, one line per idle context: "<program key> device <d> kernel <name> hint_mpixels <n>" (NUL-terminated,
 * cut to cap).  Under MARAY_BACKEND_AUTO a kept interpreter context is replaced when a later call renders more than
 * the call it was chosen for; contexts are kept per physical device. */
int maray_gen_cache_info(char *out, size_t cap);
/* gen: render and write a PNG (progress callback prints "%.2f %%" to stderr
 * and re-saves the partial image, like src/lib.rs:1203-1208). */
int maray_gen(const maray_scene *s, const maray_texture *tex, uint32_t n_tex,
              const maray_gen_opts *opts, maray_report report, const char *png_path);

/* PNG I/O used by gen and the CLI (`img.save`, `image::open(..).to_rgb8()`). */
int maray_png_write(const char *path, const uint8_t *rgb8, uint32_t w, uint32_t h);
int maray_png_read(const char *path, uint8_t **rgb8_out, uint32_t *w, uint32_t *h);   /* free with maray_free */
/* `image::open(file).unwrap().to_rgb8()` (examples/maray.rs:58-65): a texture in any format this library reads -- told from
 * the file's first bytes like the `image` crate does: PNG, BMP, PNM (P1 - P6), TGA, QOI, farbfeld, GIF (the first frame),
 * TIFF (grey / RGB strips) -- as RGB8 (alpha dropped, grey replicated, 16-bit samples rounded).  JPEG / WebP:
 * MARAY_E_DECODE naming the format. */
int maray_image_read(const char *path, uint8_t **rgb8_out, uint32_t *w, uint32_t *h);   /* free with maray_free */
void maray_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
