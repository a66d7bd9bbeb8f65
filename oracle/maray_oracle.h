/*
 * maray_oracle.h — CPU ORACLE for the maray per-pixel expression evaluator.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the reported CPU baseline.  Nothing under
 * maray_amd/ links, loads or calls it.
 *
 * It is a plain-C restatement of the reference (advancedresearch/maray 0.3.8)
 * interpreter path, function by function:
 *
 *   oracle_scene_from_bytes  <- maray::open            src/lib.rs:1227-1235
 *                               serde schema of Expr   src/lib.rs:101-149
 *                               Context                src/lib.rs:51-55
 *                               Token                  src/token.rs:10-38
 *   oracle_fix_color         <- var_fixer::fix_color   src/var_fixer.rs:74-82
 *                               VarFixer::fix          src/var_fixer.rs:25-70
 *   oracle_eval2             <- Expr::eval2            src/lib.rs:623-670
 *   (dep_x)                  <- Expr::dep_x            src/lib.rs:675-706
 *   (cache)                  <- Cache                  src/cache.rs:6-42
 *   oracle_render_rows       <- par_gen_to_image       src/render.rs:35-99
 *                               single_gen_to_image    src/render.rs:4-32
 *   textures                 <- textures::fun_*        src/textures.rs:14-65
 *
 * Third-party arithmetic the reference reaches through Rust `std`
 * (f64::sin/exp/ln, src/lib.rs:648-650) is the platform libm; this oracle calls
 * the same libm (glibc `sin`, `exp`, `log`) so that it computes what the
 * reference would compute if it were built and run on this very host.
 *
 * Pinning: checked against every known answer of the reference's own
 * `it_works` test (src/lib.rs:1241-1285), the `test_var_fixer` vectors
 * (src/lib.rs:1517-1691) and the data/chess.maray -> images/chess.png fixture
 * (>= 99.98 %, mismatches confined to rows 512 and 704) in
 * tests/test_oracle_*.py.  `exp`/`ln` and max/min on NaN/±0 have no known
 * answer in the reference: parity unpinned for those beyond libm itself.
 */
#ifndef MARAY_ORACLE_H
#define MARAY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scene oracle_scene;

typedef struct {
    const uint8_t *rgb; /* interleaved RGB8, row-major */
    uint32_t w, h;
} oracle_texture;

/* Decode `([u32;2], [Expr;3])` (bincode 1.3.3 default options).  `legacy`:
 * 0 = current tag numbering, 1 = legacy (current - 1, no Arc variant),
 * -1 = auto (accept the numbering that consumes the buffer exactly).
 * Returns NULL on error (message via oracle_last_error). */
oracle_scene *oracle_scene_from_bytes(const uint8_t *buf, size_t len, int legacy);
oracle_scene *oracle_scene_open(const char *path);
void oracle_scene_free(oracle_scene *s);
const char *oracle_last_error(void);

void oracle_scene_size(const oracle_scene *s, uint32_t *w, uint32_t *h);
/* 1 if the file was read with the legacy numbering. */
int oracle_scene_is_legacy(const oracle_scene *s);
/* Number of tree nodes in channel c (0..2), counting definitions of Let
 * variables once (the file's own node count). */
uint64_t oracle_scene_node_count(const oracle_scene *s, int c);
/* Histogram of node tags (22 entries, current numbering) for channel c. */
void oracle_scene_tag_histogram(const oracle_scene *s, int c, uint64_t out[22]);

/* Apply var_fixer::fix_color in place (renderers do this first,
 * src/render.rs:14,51,117).  Idempotent flag kept in the scene. */
void oracle_fix_color(oracle_scene *s);
/* Re-encode channel c in the *current* numbering (used to check fix_color
 * against the reference's expected trees).  Returns bytes written or the
 * needed size if cap is too small. */
size_t oracle_scene_encode_channel(const oracle_scene *s, int c, uint8_t *out, size_t cap);

/* Expr::eval2 of channel c at (x, y) with a fresh Cache and an empty outer
 * Context — what `Expr::eval` (src/lib.rs:617-620) does, in 2-D. */
double oracle_eval2(const oracle_scene *s, int c, double x, double y,
                    const oracle_texture *tex, uint32_t n_tex);

/* Render rows [y0, y1) of a w x h image.  Mirrors par_gen_to_image: per row a
 * fresh Cache, per pixel clear_dep_x, p = [x as f64, y as f64], R then G then
 * B, `as u8`.  rgb8: (y1-y0)*w*3 bytes or NULL; rgb64: (y1-y0)*w*3 doubles
 * (the pre-cast values) or NULL.  threads <= 1 -> single thread.  Calls
 * oracle_fix_color first if not yet applied.  Returns 0 on success. */
int oracle_render_rows(oracle_scene *s, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                       const oracle_texture *tex, uint32_t n_tex, int threads,
                       uint8_t *rgb8, double *rgb64);

/* CPU baseline only: the scene as a straight-line C translation unit — a stand-in for the
 * reference's wasmer JIT (src/wasm.rs:89-158; one un-shared function per channel, host-import
 * style out-of-line calls).  malloc'd; release with oracle_free.  Exports
 * `int jit_render_rows(uint32_t w, uint32_t y0, uint32_t y1, int threads, uint8_t *rgb8)`. */
char *oracle_scene_emit_c(oracle_scene *s);
void oracle_free(void *p);

/* Rust `f64 as u8` (saturating, NaN -> 0, truncation toward zero). */
uint8_t oracle_cast_u8(double v);

/* Scalar op semantics exposed for per-op sweeps. */
double oracle_op_unary(int tag, double a);   /* tag in current numbering */
double oracle_op_binary(int tag, double a, double b);

#ifdef __cplusplus
}
#endif
#endif
