/*
 * maray_tape.h — the flat f64 op tape: the wire format between the host
 * lowering (Expr -> tape) and the MI355X evaluators.  Part of the C ABI,
 * versioned by MARAY_TAPE_VERSION.
 *
 * A program has two straight-line sections that share one constant pool:
 *
 *   ROW section   evaluated once per image row (Y-only sub-expressions hoisted
 *                 out of the pixel loop); its OUT ops write the row's "y
 *                 values" yv[0..n_yvals).  May be empty.
 *   PIXEL section evaluated once per pixel; its OUT ops write channel 0,1,2
 *                 (R,G,B: the three `Color` expressions, src/lib.rs:48).
 *
 * The opcode set is the computing variants of `Expr` (src/lib.rs:101-149) —
 * Neg Abs Recip Sqrt Step Sin Exp Ln Add Mul Max Min App — with the semantics
 * of `Expr::eval2` (src/lib.rs:623-670); X, Y, Tau, E, Nat, Var, Let, Arc and
 * Decor are resolved by the lowering and never appear as ops.
 *
 * One op = one 64-bit little-endian word:
 *
 *   bits  0..6   opcode   (MARAY_OP_*)
 *   bits  7..19  aux      (OUT: output index; APP/TEXDIM: function id; SKIPZ/SKIPNZ: op count;
 *                         SIN/STEPSIN: bit 0 = MARAY_AUX_SIN_BOUNDED)
 *   bits 20..31  dst      (value slot written, or MARAY_DST_NONE)
 *   bits 32..47  a        (operand reference)
 *   bits 48..63  b        (operand reference; 0 for unary ops)
 *
 * Operand reference = kind << 14 | index:
 *
 *   kind 0 SLOT   per-item value slot `index` (< n_slots)
 *   kind 1 CONST  consts[index]
 *   kind 2 YVAL   yv[index] of the current row (PIXEL section only)
 *   kind 3 SPEC   index 0 = X (pixel x as f64), 1 = Y (row y as f64),
 *                 2 = ACC (result of the immediately preceding op),
 *                 3 = XMAX, 4 = XMIN (ROW section only): the largest and the smallest x, as f64, of the
 *                 span of pixels that the row's SKIP guards are being evaluated for.  Only y values
 *                 that gate SKIP ops depend on them (they bound a boolean over the span: guard == 0
 *                 proves the boolean 0 for every x in [XMIN, XMAX]); y values read as operands never
 *                 do.  An evaluator that runs the ROW section once per row passes 0 and w - 1; one
 *                 that evaluates the guards per tile of a row passes the tile's ends and skips more.
 *                 5 = YMAX, 6 = YMIN (ROW section only): the same for rows.  A guard whose cone reads
 *                 YMIN / YMAX and not Y holds for every pixel of [XMIN, XMAX] x [YMIN, YMAX]; an
 *                 evaluator may compute it once for several rows.  A guard that reads Y is exact
 *                 for that row and valid for that row only.  Evaluating per row: YMIN = YMAX = Y.
 *
 * Every op also leaves its result in ACC.  An op with dst == MARAY_DST_NONE
 * is consumed only through ACC by the next op.
 */
#ifndef MARAY_TAPE_H
#define MARAY_TAPE_H

#include <stdint.h>

#define MARAY_TAPE_VERSION 2u   /* 2: SPEC XMIN, YMAX, YMIN */

enum {
    MARAY_OP_NOP = 0,
    MARAY_OP_MOV = 1,     /* dst = a */
    MARAY_OP_NEG = 2,     /* src/lib.rs:640 */
    MARAY_OP_ABS = 3,     /* :641 */
    MARAY_OP_RECIP = 4,   /* :642  1.0 / a */
    MARAY_OP_SQRT = 5,    /* :643 */
    MARAY_OP_STEP = 6,    /* :644-647  a >= 0 ? 1 : 0 */
    MARAY_OP_SIN = 7,     /* :648  glibc 2.35 sin, bit-exact */
    MARAY_OP_EXP = 8,     /* :649 */
    MARAY_OP_LN = 9,      /* :650 */
    MARAY_OP_ADD = 10,    /* :651 */
    MARAY_OP_MUL = 11,    /* :653 */
    MARAY_OP_MAX = 12,    /* :655  f64::max (NaN-ignoring) */
    MARAY_OP_MIN = 13,    /* :657 */
    MARAY_OP_APP = 14,    /* :664-668 with textures::functions (src/textures.rs:54-65), aux = id, id % 5 < 3 */
    MARAY_OP_TEXDIM = 15, /* App with id % 5 in {3,4}: image width / height (src/textures.rs:40-50), aux = id */
    MARAY_OP_OUT = 16,    /* output[aux] = a */
    MARAY_OP_STEPSIN = 17, /* Step(Sin(a)) fused: :644-648; only the sign of glibc's sin is computed */
    /* Wave-level short circuit of boolean algebra (values that are exactly +0.0 or 1.0):
     * SKIPZ  a, n: if a == 0.0 on every lane of the wavefront, the result of the AND (Mul / Min)
     *              that ends the next n ops is +0.0 whatever its other operand is: write +0.0 to dst
     *              (if any) and ACC, and skip those n ops.  Otherwise no effect (ACC unchanged).
     * SKIPNZ a, n: same for an OR (Max) when a == 1.0 on every lane: the result is 1.0.
     * n = aux.  The skipped ops define no value that is read after them, and contain no OUT.
     * An evaluator may ignore both ops (never skip): results are identical. */
    MARAY_OP_SKIPZ = 18,
    MARAY_OP_SKIPNZ = 19,
    MARAY_OP_COUNT = 20
};

/* SIN / STEPSIN aux bit 0: the lowering proved by interval arithmetic that the
 * argument is finite with |a| < 105414350 (glibc's reduce_sincos range) for every
 * pixel with x, y < MARAY_DOMAIN_MAX, so the huge-argument path cannot be taken. */
#define MARAY_AUX_SIN_BOUNDED 1u
#define MARAY_DOMAIN_MAX 1048576u

#define MARAY_DST_NONE 0xFFFu
#define MARAY_MAX_SLOTS 0xFFFu
#define MARAY_MAX_INDEX 0x3FFFu

enum { MARAY_K_SLOT = 0, MARAY_K_CONST = 1, MARAY_K_YVAL = 2, MARAY_K_SPEC = 3 };
enum { MARAY_SPEC_X = 0, MARAY_SPEC_Y = 1, MARAY_SPEC_ACC = 2, MARAY_SPEC_XMAX = 3, MARAY_SPEC_XMIN = 4, MARAY_SPEC_YMAX = 5, MARAY_SPEC_YMIN = 6 };

#define MARAY_REF(kind, index) ((uint32_t)(((kind) << 14) | ((index) & 0x3FFFu)))
#define MARAY_REF_KIND(r) (((r) >> 14) & 3u)
#define MARAY_REF_INDEX(r) ((r) & 0x3FFFu)

#define MARAY_INS(op, aux, dst, a, b)                                                    \
    ((uint64_t)((op) & 0x7Fu) | ((uint64_t)((aux) & 0x1FFFu) << 7) |                      \
     ((uint64_t)((dst) & 0xFFFu) << 20) | ((uint64_t)((a) & 0xFFFFu) << 32) |            \
     ((uint64_t)((b) & 0xFFFFu) << 48))
#define MARAY_INS_OP(i) ((uint32_t)((i) & 0x7Fu))
#define MARAY_INS_AUX(i) ((uint32_t)(((i) >> 7) & 0x1FFFu))
#define MARAY_INS_DST(i) ((uint32_t)(((i) >> 20) & 0xFFFu))
#define MARAY_INS_A(i) ((uint32_t)(((i) >> 32) & 0xFFFFu))
#define MARAY_INS_B(i) ((uint32_t)(((i) >> 48) & 0xFFFFu))

/* A lowered program, as plain pointers + sizes (what the tape-level ABI takes). */
typedef struct maray_program {
    uint32_t version;        /* MARAY_TAPE_VERSION */
    uint32_t n_consts;
    const double *consts;
    uint32_t n_row_ops;      /* ROW section */
    const uint64_t *row_ops;
    uint32_t n_row_slots;
    uint32_t n_yvals;        /* outputs of the ROW section */
    uint32_t n_pix_ops;      /* PIXEL section */
    const uint64_t *pix_ops;
    uint32_t n_pix_slots;
    uint32_t n_app;          /* 1 + highest App id used (0 if none) */
} maray_program;

#endif
