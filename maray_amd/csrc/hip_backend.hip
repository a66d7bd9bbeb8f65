// hip_backend.hip — gfx950 evaluators behind the tape-level C ABI (product code).
//
// Replaces the per-pixel loop of par_gen_to_image / wasm_par_gen_to_image
// (src/render.rs:85-97, :168-183): one work-item per pixel, the tape walked
// once per wavefront with wave-uniform dispatch (the op word lives in SGPRs,
// the switch is a scalar branch, every lane executes the same VALU op).
//
// Data layout in HBM (uploaded once at ctx creation):
//   tape      n_ops  x u64      (ROW section, PIXEL section)
//   consts    n_consts x f64
//   textures  RGB8 rasters + a MarayTex descriptor table
//   yvals     rows x n_yvals f64   (written by the ROW kernel, read as scalars)
//   spill     (n_slots - n_lds_slots) x grid_threads f64, only if the value
//             slots do not fit in LDS
// Outputs: rgb8 (rows*w*3 bytes) and/or rgb64 (rows*w*3 doubles), interleaved.
//
// LDS layout of the pixel kernel (dynamic):
//   [ tape (TAPE_LDS only) | consts (TAPE_LDS only) | slots: n_lds_slots x 256 f64 ]
// Slot s of thread t is at slots[s*256 + t]: ds_read_b64/ds_write_b64 hit 64
// consecutive 8-byte words per wave, conflict-free.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "backend.hpp"
#include "device_math.h"
#include "host_pipe.hpp"
#include "maray_hip.h"

namespace maray {

namespace {

constexpr int BLOCK = 256;

struct KArgs {
    const uint64_t *tape;      // section to run
    const double *consts;
    const double *yvals;       // PIXEL: rows x n_yvals (row-major); ROW: unused
    double *yout;              // ROW: rows x n_yvals
    double *spill;             // [slot - n_lds_slots][grid_threads]
    const MarayTex *tex;
    const unsigned *tile_list;    // optional work list {count, tile, tile, ...}: only those tiles are evaluated
    unsigned char *rgb8;
    double *rgb64;
    uint32_t n_ops, n_consts, n_yvals, n_slots, n_lds_slots;
    uint32_t w, y0, rows;      // image width, first row, row count of this launch
    uint32_t blk_rows, blk_stride;   // launch row r is image row y0 + (r / blk_rows) * blk_stride + r % blk_rows (RowBlocks)
    uint32_t tiles_per_row, n_tiles;
    // Guards evaluated per rectangle of guard_rows rows x 256 / guard_sub pixels (guard_w32 != 0): bit g of a rectangle's words =
    // guard guard_first + g.  PIXEL reads them instead of y values; the GUARDS kernel (job j = 8 guards = one byte,
    // tape[job_off[j] .. + job_len[j])) writes them.
    uint32_t *gbits;
    uint32_t *queue;           // GUARDS: next (job, block of items) unit to hand out (zeroed before the launch)
    const uint32_t *job_id;    // GUARDS: jobs are listed longest first; job_id[k] = which byte of the guard bits job k writes
    const uint64_t *xtape;     // PIXEL: the section pre-decoded for run_xtape, or null (generic loop)
    uint32_t x_slot;           // run_xtape: slots x_slot, x_slot + 1, x_slot + 2 hold X, Y and the results nothing reads
    const uint32_t *job_off, *job_len;
    uint32_t guard_first, guard_w32, guard_rows;
    uint32_t guard_sub;            // guard rectangles per 256-pixel tile (1, 2 or 4: a rectangle is 256 / guard_sub pixels wide)
};

typedef const __attribute__((address_space(4))) uint64_t *k_u64_ptr;   // constant address space: scalar loads
typedef const __attribute__((address_space(4))) double *k_f64_ptr;

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

enum { MODE_PIXEL = 0, MODE_ROW = 1, MODE_GUARDS = 2 };

// What one work-item evaluates the tape for.
struct Item {
    const double *yrow;            // PIXEL: the row's y values
    const uint32_t *gk;            // PIXEL: the guard words of the pixel's rectangle, or null (guards are y values)
    double X, Y;                   // pixel / row coordinates
    double xmin, xmax, ymin, ymax; // ROW sections: the span of pixels the guards are bounded over
    double *yout;                  // MODE_ROW: OUT k -> yout[k]
};

// One pass over a tape section for the calling work-item.
//   MODE_PIXEL : item is a pixel, OUT k -> out[k] (k = 0,1,2)
//   MODE_ROW   : item is an image row, OUT k -> yout[k]
//   MODE_GUARDS: item is a rectangle of pixels, OUT k (a guard) -> bit (k - guard_first) % 8 of `gacc`
template <bool TAPE_LDS, int MODE>
__device__ __forceinline__ void run_tape(const KArgs &A, const uint64_t *tape_g, uint32_t n_ops, const uint64_t *tape_lds, const double *consts_lds,
                                         double *slots, double *spill_base, uint32_t spill_stride,
                                         const Item &I, double &o0, double &o1, double &o2, uint32_t &gacc)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t n_lds = A.n_lds_slots;
    const double X = I.X, Y = I.Y;
    double acc = 0.0;
    k_u64_ptr tape_k = (k_u64_ptr)tape_g;
    k_f64_ptr consts_k = (k_f64_ptr)A.consts;
    k_f64_ptr yrow_k = (k_f64_ptr)I.yrow;
    const __attribute__((address_space(4))) uint32_t *gk_k = (const __attribute__((address_space(4))) uint32_t *)I.gk;

    auto fetch = [&](uint32_t ref) -> double {
        const uint32_t kind = ref >> 14, idx = ref & 0x3FFFu;   // wave-uniform
        if (kind == MARAY_K_SLOT) return idx < n_lds ? slots[idx * BLOCK + tid] : spill_base[(size_t)(idx - n_lds) * spill_stride];
        if (kind == MARAY_K_CONST) return TAPE_LDS ? consts_lds[idx] : consts_k[idx];
        if (kind == MARAY_K_YVAL) {
            if (MODE == MODE_PIXEL && I.gk && idx >= A.guard_first) {      // a guard of this pixel's rectangle: one bit
                const uint32_t g = idx - A.guard_first;
                return ((gk_k[g >> 5] >> (g & 31u)) & 1u) ? 1.0 : 0.0;
            }
            return yrow_k[idx];
        }
        switch (idx) {
        case MARAY_SPEC_X: return X;
        case MARAY_SPEC_Y: return Y;
        case MARAY_SPEC_ACC: return acc;
        case MARAY_SPEC_XMAX: return I.xmax;
        case MARAY_SPEC_XMIN: return I.xmin;
        case MARAY_SPEC_YMAX: return I.ymax;
        default: return I.ymin;
        }
    };

    for (uint32_t pc = 0; pc < n_ops; ++pc) {
        uint32_t lo, hi;
        if (TAPE_LDS) {
            const uint64_t ins = tape_lds[pc];           // same LDS address in every lane: broadcast read
            lo = uni((uint32_t)ins); hi = uni((uint32_t)(ins >> 32));
        } else {
            const uint64_t ins = tape_k[pc];             // s_load through the scalar cache
            lo = uni((uint32_t)ins); hi = uni((uint32_t)(ins >> 32));
        }
        const uint32_t op = lo & 0x7Fu, aux = (lo >> 7) & 0x1FFFu, dst = lo >> 20;
        const uint32_t ra = hi & 0xFFFFu, rb = hi >> 16;
        double r;
        switch (op) {
        case MARAY_OP_MOV: r = fetch(ra); break;
        case MARAY_OP_NEG: r = mr_neg(fetch(ra)); break;
        case MARAY_OP_ABS: r = mr_abs(fetch(ra)); break;
        case MARAY_OP_RECIP: r = mr_recip(fetch(ra)); break;
        case MARAY_OP_SQRT: r = mr_sqrt(fetch(ra)); break;
        case MARAY_OP_STEP: r = mr_step(fetch(ra)); break;
        case MARAY_OP_SIN: r = mr_sin(fetch(ra)); break;
        case MARAY_OP_STEPSIN: r = mr_stepsin(fetch(ra)); break;
        case MARAY_OP_EXP: r = mr_exp(fetch(ra)); break;
        case MARAY_OP_LN: r = mr_ln(fetch(ra)); break;
        case MARAY_OP_ADD: { const double a = fetch(ra), b = fetch(rb); r = a + b; break; }
        case MARAY_OP_MUL: { const double a = fetch(ra), b = fetch(rb); r = a * b; break; }
        case MARAY_OP_MAX: { const double a = fetch(ra), b = fetch(rb); r = mr_max(a, b); break; }
        case MARAY_OP_MIN: { const double a = fetch(ra), b = fetch(rb); r = mr_min(a, b); break; }
        case MARAY_OP_APP: { const double a = fetch(ra), b = fetch(rb); r = mr_app(A.tex, aux, a, b); break; }
        case MARAY_OP_TEXDIM: r = mr_texdim(A.tex, aux); break;
        case MARAY_OP_SKIPZ: case MARAY_OP_SKIPNZ: {
            // wave-level short circuit of boolean algebra: every lane of this wavefront agrees that the
            // AND (OR) ending `aux` ops further down is 0 (1) -> define its value and jump over the region
            const double gv = fetch(ra);
            const bool decided = (op == MARAY_OP_SKIPZ) ? (__builtin_amdgcn_ballot_w64(gv != 0.0) == 0ull)
                                                        : (__builtin_amdgcn_ballot_w64(gv != 1.0) == 0ull);
            if (decided) {
                acc = (op == MARAY_OP_SKIPZ) ? 0.0 : 1.0;
                if (dst != MARAY_DST_NONE) {
                    if (dst < n_lds) slots[dst * BLOCK + tid] = acc;
                    else spill_base[(size_t)(dst - n_lds) * spill_stride] = acc;
                }
                pc += aux;
            }
            continue;
        }
        case MARAY_OP_OUT: {
            const double v = fetch(ra);
            if (MODE == MODE_ROW) I.yout[aux] = v;
            else if (MODE == MODE_GUARDS) { if (v != 0.0) gacc |= 1u << ((aux - A.guard_first) & 7u); }
            else if (aux == 0) o0 = v;
            else if (aux == 1) o1 = v;
            else o2 = v;
            continue;                                     // OUT leaves ACC and slots untouched
        }
        default: continue;                                // NOP
        }
        acc = r;
        if (dst != MARAY_DST_NONE) {
            if (dst < n_lds) slots[dst * BLOCK + tid] = r;
            else spill_base[(size_t)(dst - n_lds) * spill_stride] = r;
        }
    }
}


// ---- PIXEL section, pre-decoded ("xtape") --------------------------------------------------------
// The generic loop above spends ~20 scalar branches per op finding out what its operands are (kind of a, kind of b,
// which special, LDS or spill, dst or none): every condition is wave-uniform, so each is a real branch, and branches
// -- not arithmetic -- are what an op costs.  For the PIXEL section the host rewrites each op once into a word
// whose opcode already says where the operands live, and the loop is one jump table:
//   operand classes  S: value slot in LDS (the item's X, Y -- and in the ROW sections its span XMIN .. YMAX -- are parked
//                       in reserved slots: x_slot + 0 X, 1 Y, 2 trash, 3 XMIN, 4 XMAX, 5 YMIN, 6 YMAX)
//                    A: ACC          U: wave-uniform table: constant `index`, or with bit 15 set y value `index & 0x7fff`
//   dst: a slot, or the reserved trash slot (no "if (dst != none)")
// Same layout as a tape word (op 7 | aux 13 | dst 12 | a 16 | b 16).  Programs whose slots do not fit LDS keep the
// generic loop.
enum {
    XS = 0, XA = 1, XU = 2,                       // operand classes
    XG = 3,                                       // SKIP guard: a bit of the rectangle's guard words
    X_NOP = 0,
    X_BIN = 16,                                   // + 9 * {ADD, MUL, MAX, MIN} + 3 * class(a) + class(b)
    X_UN = 52,                                    // + 3 * {NEG, ABS, STEP, MOV} + class(a)
    X_HEAVY = 64,                                 // + {RECIP, SQRT, SIN, STEPSIN, EXP, LN}; class(a) in the b field
    X_APP = 72,                                   // + 3 * class(a) + class(b)
    X_TEXDIM = 81,
    X_OUT = 82,                                   // + class(a)
    X_SKIPZ = 85, X_SKIPNZ = 89,                  // + class(guard) (XS, XA, XU = a y value bounded over the row, XG)
};

// The ops with data-dependent branches inside (range reduction, table lookups, bounds checks) are CALLED from the
// pre-decoded loop: with them inlined, the loop body is one region holding divergent branches, the compiler
// structurizes all of it, and every case -- the plain Add as well -- leaves through a ladder of flag tests instead of
// one jump to the loop's latch.  Out of line, every branch of the loop is a scalar branch and the switch stays a switch.
__device__ __attribute__((noinline)) double xt_recip(double a) { return mr_recip(a); }
__device__ __attribute__((noinline)) double xt_sqrt(double a) { return mr_sqrt(a); }
__device__ __attribute__((noinline)) double xt_sin(double a) { return mr_sin(a); }
__device__ __attribute__((noinline)) double xt_stepsin(double a) { return mr_stepsin(a); }
__device__ __attribute__((noinline)) double xt_exp(double a) { return mr_exp(a); }
__device__ __attribute__((noinline)) double xt_ln(double a) { return mr_ln(a); }
__device__ __attribute__((noinline)) double xt_app(const MarayTex *tex, uint32_t id, double a, double b) { return mr_app(tex, id, a, b); }

template <bool TAPE_LDS, int MODE>
__device__ __forceinline__ void run_xtape(const KArgs &A, const uint64_t *xtape, uint32_t n_ops, const uint64_t *tape_lds, const double *consts_lds,
                                          double *slots, const Item &I, double &o0, double &o1, double &o2, uint32_t &gacc)
{
    const uint32_t tid = threadIdx.x;
    double acc = 0.0;
    k_u64_ptr tape_k = (k_u64_ptr)xtape;
    k_f64_ptr consts_k = (k_f64_ptr)A.consts;
    k_f64_ptr yrow_k = (k_f64_ptr)I.yrow;
    const __attribute__((address_space(4))) uint32_t *gk_k = (const __attribute__((address_space(4))) uint32_t *)I.gk;
#define FS(i) slots[(i) * BLOCK + tid]
    auto FU = [&](uint32_t i) -> double {          // bit 15 of the index: a y value of the row, else a constant
        const uint32_t k = i & 0x7FFFu;
        if (TAPE_LDS) {                            // both tables are read, the bit picks: no branch
            const double c = consts_lds[(i & 0x8000u) ? 0u : k];
            const double y = yrow_k[(i & 0x8000u) ? k : 0u];
            return (i & 0x8000u) ? y : c;
        }
        k_f64_ptr base = (i & 0x8000u) ? yrow_k : consts_k;          // one scalar select of the base, one scalar load
        return base[k];
    };
    auto FG = [&](uint32_t k, uint32_t i) -> double { return k == XS ? FS(i) : (k == XA ? acc : FU(i)); };   // heavy ops only
    // the ROW kernels run a few wavefronts per CU (their slots fill the LDS): nothing hides the scalar load of the next
    // word, so it is issued one op ahead (the PIXEL kernel has the occupancy and gains nothing from it)
    constexpr bool AHEAD = !TAPE_LDS && MODE != MODE_PIXEL;
    uint64_t ahead = (AHEAD && n_ops) ? tape_k[0] : 0;
    for (uint32_t pc = 0; pc < n_ops; ++pc) {
        uint32_t lo, hi;
        if (TAPE_LDS) {
            const uint64_t ins = tape_lds[pc];
            lo = uni((uint32_t)ins); hi = uni((uint32_t)(ins >> 32));
        } else if (AHEAD) {
            lo = uni((uint32_t)ahead); hi = uni((uint32_t)(ahead >> 32));
            ahead = tape_k[pc + 1 < n_ops ? pc + 1 : pc];
        } else {
            const uint64_t ins = tape_k[pc];
            lo = uni((uint32_t)ins); hi = uni((uint32_t)(ins >> 32));
        }
        const uint32_t op = lo & 0x7Fu, aux = (lo >> 7) & 0x1FFFu, dst = lo >> 20;
        const uint32_t ia = hi & 0xFFFFu, ib = hi >> 16;
        double r;
        // Dispatch: a branch table.  The back end has no jump tables, a `switch` is a tree of six compare-and-branch
        // levels, each on another line of the instruction cache; here the opcode indexes a table of s_branch
        // instructions that follows the s_setpc (s_getpc returns the address of the instruction after itself: the
        // table starts 12 bytes on, one 4-byte s_branch per opcode).
        {
            const uint32_t joff = (op < 93u ? op : 0u) * 4u + 12u;
            asm goto("s_getpc_b64 s[20:21]\n\ts_add_u32 s20, s20, %0\n\ts_addc_u32 s21, s21, 0\n\ts_setpc_b64 s[20:21]"
                     "\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l2\n\ts_branch %l3\n\ts_branch %l4\n\ts_branch %l5\n\ts_branch %l6\n\ts_branch %l7\n\ts_branch %l8\n\ts_branch %l9\n\ts_branch %l10\n\ts_branch %l11\n\ts_branch %l12\n\ts_branch %l13\n\ts_branch %l14\n\ts_branch %l15\n\ts_branch %l16\n\ts_branch %l17\n\ts_branch %l18\n\ts_branch %l19\n\ts_branch %l20\n\ts_branch %l21\n\ts_branch %l22\n\ts_branch %l23\n\ts_branch %l24\n\ts_branch %l25\n\ts_branch %l26\n\ts_branch %l27\n\ts_branch %l28\n\ts_branch %l29\n\ts_branch %l30\n\ts_branch %l31\n\ts_branch %l32\n\ts_branch %l33\n\ts_branch %l34\n\ts_branch %l35\n\ts_branch %l36\n\ts_branch %l37\n\ts_branch %l38\n\ts_branch %l39\n\ts_branch %l40\n\ts_branch %l41\n\ts_branch %l42\n\ts_branch %l43\n\ts_branch %l44\n\ts_branch %l45\n\ts_branch %l46\n\ts_branch %l47\n\ts_branch %l48\n\ts_branch %l49\n\ts_branch %l50\n\ts_branch %l51\n\ts_branch %l52\n\ts_branch %l53\n\ts_branch %l54\n\ts_branch %l55\n\ts_branch %l1\n\ts_branch %l1\n\ts_branch %l56\n\ts_branch %l57\n\ts_branch %l58\n\ts_branch %l59\n\ts_branch %l60\n\ts_branch %l61\n\ts_branch %l62\n\ts_branch %l63\n\ts_branch %l64\n\ts_branch %l65\n\ts_branch %l66\n\ts_branch %l66\n\ts_branch %l66\n\ts_branch %l67\n\ts_branch %l67\n\ts_branch %l67\n\ts_branch %l67\n\ts_branch %l67\n\ts_branch %l67\n\ts_branch %l67\n\ts_branch %l67"
                     : : "s"(joff) : "s20", "s21", "scc" : XL_NOP, XL_ADD0, XL_ADD1, XL_ADD2, XL_ADD3, XL_ADD4, XL_ADD5, XL_ADD6, XL_ADD7, XL_ADD8, XL_MUL0, XL_MUL1, XL_MUL2, XL_MUL3, XL_MUL4, XL_MUL5, XL_MUL6, XL_MUL7, XL_MUL8, XL_MAX0, XL_MAX1, XL_MAX2, XL_MAX3, XL_MAX4, XL_MAX5, XL_MAX6, XL_MAX7, XL_MAX8, XL_MIN0, XL_MIN1, XL_MIN2, XL_MIN3, XL_MIN4, XL_MIN5, XL_MIN6, XL_MIN7, XL_MIN8, XL_NEG0, XL_NEG1, XL_NEG2, XL_ABS0, XL_ABS1, XL_ABS2, XL_STEP0, XL_STEP1, XL_STEP2, XL_MOV0, XL_MOV1, XL_MOV2, XL_H0, XL_H1, XL_H2, XL_H3, XL_H4, XL_H5, XL_APP0, XL_APP1, XL_APP2, XL_APP3, XL_APP4, XL_APP5, XL_APP6, XL_APP7, XL_APP8, XL_TEXDIM, XL_OUT, XL_SKIP);
            goto XL_NEXT;                                 // (not reached: the asm always jumps)
        }
        XL_ADD0: { const double a = FS(ia), b = FS(ib); r = a + b; goto XL_WRITE; }
        XL_ADD1: { const double a = FS(ia), b = acc; r = a + b; goto XL_WRITE; }
        XL_ADD2: { const double a = FS(ia), b = FU(ib); r = a + b; goto XL_WRITE; }
        XL_ADD3: { const double a = acc, b = FS(ib); r = a + b; goto XL_WRITE; }
        XL_ADD4: { const double a = acc, b = acc; r = a + b; goto XL_WRITE; }
        XL_ADD5: { const double a = acc, b = FU(ib); r = a + b; goto XL_WRITE; }
        XL_ADD6: { const double a = FU(ia), b = FS(ib); r = a + b; goto XL_WRITE; }
        XL_ADD7: { const double a = FU(ia), b = acc; r = a + b; goto XL_WRITE; }
        XL_ADD8: { const double a = FU(ia), b = FU(ib); r = a + b; goto XL_WRITE; }
        XL_MUL0: { const double a = FS(ia), b = FS(ib); r = a * b; goto XL_WRITE; }
        XL_MUL1: { const double a = FS(ia), b = acc; r = a * b; goto XL_WRITE; }
        XL_MUL2: { const double a = FS(ia), b = FU(ib); r = a * b; goto XL_WRITE; }
        XL_MUL3: { const double a = acc, b = FS(ib); r = a * b; goto XL_WRITE; }
        XL_MUL4: { const double a = acc, b = acc; r = a * b; goto XL_WRITE; }
        XL_MUL5: { const double a = acc, b = FU(ib); r = a * b; goto XL_WRITE; }
        XL_MUL6: { const double a = FU(ia), b = FS(ib); r = a * b; goto XL_WRITE; }
        XL_MUL7: { const double a = FU(ia), b = acc; r = a * b; goto XL_WRITE; }
        XL_MUL8: { const double a = FU(ia), b = FU(ib); r = a * b; goto XL_WRITE; }
        XL_MAX0: { const double a = FS(ia), b = FS(ib); r = mr_max(a, b); goto XL_WRITE; }
        XL_MAX1: { const double a = FS(ia), b = acc; r = mr_max(a, b); goto XL_WRITE; }
        XL_MAX2: { const double a = FS(ia), b = FU(ib); r = mr_max(a, b); goto XL_WRITE; }
        XL_MAX3: { const double a = acc, b = FS(ib); r = mr_max(a, b); goto XL_WRITE; }
        XL_MAX4: { const double a = acc, b = acc; r = mr_max(a, b); goto XL_WRITE; }
        XL_MAX5: { const double a = acc, b = FU(ib); r = mr_max(a, b); goto XL_WRITE; }
        XL_MAX6: { const double a = FU(ia), b = FS(ib); r = mr_max(a, b); goto XL_WRITE; }
        XL_MAX7: { const double a = FU(ia), b = acc; r = mr_max(a, b); goto XL_WRITE; }
        XL_MAX8: { const double a = FU(ia), b = FU(ib); r = mr_max(a, b); goto XL_WRITE; }
        XL_MIN0: { const double a = FS(ia), b = FS(ib); r = mr_min(a, b); goto XL_WRITE; }
        XL_MIN1: { const double a = FS(ia), b = acc; r = mr_min(a, b); goto XL_WRITE; }
        XL_MIN2: { const double a = FS(ia), b = FU(ib); r = mr_min(a, b); goto XL_WRITE; }
        XL_MIN3: { const double a = acc, b = FS(ib); r = mr_min(a, b); goto XL_WRITE; }
        XL_MIN4: { const double a = acc, b = acc; r = mr_min(a, b); goto XL_WRITE; }
        XL_MIN5: { const double a = acc, b = FU(ib); r = mr_min(a, b); goto XL_WRITE; }
        XL_MIN6: { const double a = FU(ia), b = FS(ib); r = mr_min(a, b); goto XL_WRITE; }
        XL_MIN7: { const double a = FU(ia), b = acc; r = mr_min(a, b); goto XL_WRITE; }
        XL_MIN8: { const double a = FU(ia), b = FU(ib); r = mr_min(a, b); goto XL_WRITE; }
        XL_NEG0: { const double a = FS(ia); r = mr_neg(a); goto XL_WRITE; }
        XL_NEG1: { const double a = acc; r = mr_neg(a); goto XL_WRITE; }
        XL_NEG2: { const double a = FU(ia); r = mr_neg(a); goto XL_WRITE; }
        XL_ABS0: { const double a = FS(ia); r = mr_abs(a); goto XL_WRITE; }
        XL_ABS1: { const double a = acc; r = mr_abs(a); goto XL_WRITE; }
        XL_ABS2: { const double a = FU(ia); r = mr_abs(a); goto XL_WRITE; }
        XL_STEP0: { const double a = FS(ia); r = mr_step(a); goto XL_WRITE; }
        XL_STEP1: { const double a = acc; r = mr_step(a); goto XL_WRITE; }
        XL_STEP2: { const double a = FU(ia); r = mr_step(a); goto XL_WRITE; }
        XL_MOV0: { const double a = FS(ia); r = a; goto XL_WRITE; }
        XL_MOV1: { const double a = acc; r = a; goto XL_WRITE; }
        XL_MOV2: { const double a = FU(ia); r = a; goto XL_WRITE; }
        XL_H0: r = xt_recip(FG(ib, ia)); goto XL_WRITE;
        XL_H1: r = xt_sqrt(FG(ib, ia)); goto XL_WRITE;
        XL_H2: r = xt_sin(FG(ib, ia)); goto XL_WRITE;
        XL_H3: r = xt_stepsin(FG(ib, ia)); goto XL_WRITE;
        XL_H4: r = xt_exp(FG(ib, ia)); goto XL_WRITE;
        XL_H5: r = xt_ln(FG(ib, ia)); goto XL_WRITE;
        XL_APP0: { const double a = FS(ia), b = FS(ib); r = xt_app(A.tex, aux, a, b); goto XL_WRITE; }
        XL_APP1: { const double a = FS(ia), b = acc; r = xt_app(A.tex, aux, a, b); goto XL_WRITE; }
        XL_APP2: { const double a = FS(ia), b = FU(ib); r = xt_app(A.tex, aux, a, b); goto XL_WRITE; }
        XL_APP3: { const double a = acc, b = FS(ib); r = xt_app(A.tex, aux, a, b); goto XL_WRITE; }
        XL_APP4: { const double a = acc, b = acc; r = xt_app(A.tex, aux, a, b); goto XL_WRITE; }
        XL_APP5: { const double a = acc, b = FU(ib); r = xt_app(A.tex, aux, a, b); goto XL_WRITE; }
        XL_APP6: { const double a = FU(ia), b = FS(ib); r = xt_app(A.tex, aux, a, b); goto XL_WRITE; }
        XL_APP7: { const double a = FU(ia), b = acc; r = xt_app(A.tex, aux, a, b); goto XL_WRITE; }
        XL_APP8: { const double a = FU(ia), b = FU(ib); r = xt_app(A.tex, aux, a, b); goto XL_WRITE; }
        XL_TEXDIM: r = mr_texdim(A.tex, aux); goto XL_WRITE;
        XL_OUT: {
            const double v = FG(op - X_OUT, ia);
            if (MODE == MODE_ROW) I.yout[aux] = v;
            else if (MODE == MODE_GUARDS) { if (v != 0.0) gacc |= 1u << ((aux - A.guard_first) & 7u); }
            else if (aux == 0) o0 = v; else if (aux == 1) o1 = v; else o2 = v;
            goto XL_NEXT;                                 // OUT leaves ACC and slots untouched
        }
        XL_SKIP: {
            const bool nz = op >= X_SKIPNZ;
            const uint32_t k = op - (nz ? X_SKIPNZ : X_SKIPZ);
            bool decided;
            if (k == XG) decided = MODE == MODE_PIXEL && ((gk_k[ia >> 5] >> (ia & 31u)) & 1u) == (nz ? 1u : 0u);       // one bit, wave-uniform
            else if (k == XU) decided = FU(ia) == (nz ? 1.0 : 0.0);                                // a y value: uniform
            else {
                const double gv = k == XS ? FS(ia) : acc;
                decided = nz ? (__builtin_amdgcn_ballot_w64(gv != 1.0) == 0ull) : (__builtin_amdgcn_ballot_w64(gv != 0.0) == 0ull);
            }
            if (decided) {
                acc = nz ? 1.0 : 0.0;
                FS(dst) = acc;
                pc += aux;
                if (AHEAD) ahead = tape_k[pc + 1 < n_ops ? pc + 1 : pc];
            }
            goto XL_NEXT;
        }
        XL_NOP: goto XL_NEXT;
    XL_WRITE:
        acc = r;
        FS(dst) = r;
    XL_NEXT:;
    }
#undef FS
}

__device__ __forceinline__ void park_specials(const KArgs &A, double *slots, const Item &I)
{
    const uint32_t t = threadIdx.x, b = A.x_slot;
    slots[(b + 1) * BLOCK + t] = I.Y;
    slots[(b + 3) * BLOCK + t] = I.xmin; slots[(b + 4) * BLOCK + t] = I.xmax;
    slots[(b + 5) * BLOCK + t] = I.ymin; slots[(b + 6) * BLOCK + t] = I.ymax;
}

// PIXEL kernel.  Block = 256 consecutive pixels of one row ("tile"); blocks
// stride over tiles so the LDS staging of the tape is paid once per block.
template <bool TAPE_LDS>
__global__ void __launch_bounds__(BLOCK) maray_tape_pixels(const KArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint64_t *tape_lds = nullptr;
    const double *consts_lds = nullptr;
    double *slots;
    if (TAPE_LDS) {
        uint64_t *tl = (uint64_t *)smem;
        double *cl = (double *)(smem + (size_t)A.n_ops * 8);
        const uint64_t *src = A.xtape ? A.xtape : A.tape;
        for (uint32_t i = threadIdx.x; i < A.n_ops; i += BLOCK) tl[i] = src[i];
        for (uint32_t i = threadIdx.x; i < A.n_consts; i += BLOCK) cl[i] = A.consts[i];
        tape_lds = tl; consts_lds = cl;
        slots = (double *)(smem + ((size_t)A.n_ops + A.n_consts) * 8);
        __syncthreads();
    } else {
        slots = (double *)smem;
    }
    const uint32_t spill_stride = gridDim.x * BLOCK;
    double *spill_base = A.spill ? A.spill + (size_t)blockIdx.x * BLOCK + threadIdx.x : nullptr;

    // the first tile of a block is its own, the rest come from a queue: tiles of the board cost tens of times a tile
    // of sky, and any fixed deal leaves blocks idle (waves were busy 58 % of the launch with a stride of gridDim.x)
    __shared__ uint32_t drawn;
    const uint32_t n_work = A.tile_list ? A.tile_list[0] : A.n_tiles;      // wave-uniform
    for (uint32_t wi = blockIdx.x; wi < n_work;) {
        const uint32_t tile = A.tile_list ? A.tile_list[1 + wi] : wi;
        const uint32_t r = tile / A.tiles_per_row;                 // row within this launch (uniform)
        const uint32_t x = (tile - r * A.tiles_per_row) * BLOCK + threadIdx.x;
        const uint32_t y = A.y0 + (r / A.blk_rows) * A.blk_stride + r % A.blk_rows;
        double o0 = 0.0, o1 = 0.0, o2 = 0.0;
        uint32_t unused = 0;
        Item I{};
        I.yrow = A.yvals + (size_t)r * A.n_yvals;
        // the guard words of this wavefront's rectangle (a wavefront's 64 pixels lie in one: rectangles are 64 .. 256 wide)
        // (readfirstlane: the index is the same on every lane of a wavefront, and the words must come by scalar loads)
        I.gk = A.guard_w32 ? A.gbits + (((size_t)(r / A.guard_rows) * A.tiles_per_row + (tile - r * A.tiles_per_row)) * A.guard_sub +
                                        (uint32_t)__builtin_amdgcn_readfirstlane((int)((threadIdx.x * A.guard_sub) / BLOCK))) * A.guard_w32 : nullptr;
        I.X = (double)x; I.Y = (double)y;                                            // p = [x as f64, y as f64]
        if (A.xtape) {
            slots[A.x_slot * BLOCK + threadIdx.x] = I.X;
            slots[(A.x_slot + 1) * BLOCK + threadIdx.x] = I.Y;
            run_xtape<TAPE_LDS, MODE_PIXEL>(A, A.xtape, A.n_ops, tape_lds, consts_lds, slots, I, o0, o1, o2, unused);
        } else
            run_tape<TAPE_LDS, MODE_PIXEL>(A, A.tape, A.n_ops, tape_lds, consts_lds, slots, spill_base, spill_stride, I, o0, o1, o2, unused);
        if (x < A.w) {
            const size_t p = ((size_t)r * A.w + x) * 3;
            if (A.rgb64) { A.rgb64[p] = o0; A.rgb64[p + 1] = o1; A.rgb64[p + 2] = o2; }
            if (A.rgb8) {
                A.rgb8[p] = (unsigned char)mr_cast_u8(o0);
                A.rgb8[p + 1] = (unsigned char)mr_cast_u8(o1);
                A.rgb8[p + 2] = (unsigned char)mr_cast_u8(o2);
            }
        }
        __syncthreads();                                           // the previous draw has been read by every wave
        if (threadIdx.x == 0) drawn = gridDim.x + atomicAdd(A.queue, 1u);
        __syncthreads();
        wi = drawn;
    }
}

// ROW kernel: one work-item per image row evaluates the ROW section (in guard-bit mode: the part of it that the
// operand y values depend on) and writes the row's y values.  Guards it evaluates are bounded over the whole row.
// Tiny (rows x n_row_ops); slots live in LDS or spill.
__global__ void __launch_bounds__(BLOCK) maray_tape_rows(const KArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *slots = (double *)smem;
    const uint32_t spill_stride = gridDim.x * BLOCK;
    double *spill_base = A.spill ? A.spill + (size_t)blockIdx.x * BLOCK + threadIdx.x : nullptr;
    const uint32_t row_blocks = (A.rows + BLOCK - 1) / BLOCK;
    // the section is cut into jobs (the cone of 8 y values each: dozens of ops, not the section's thousands in one
    // dependent chain); blocks stride over (job, block of rows) pairs, so the grid and the spill area stay bounded
    for (uint32_t u = blockIdx.x; u < row_blocks * A.n_tiles /* = jobs */; u += gridDim.x) {
        const uint32_t job = u / row_blocks;
        const uint32_t r = (u - job * row_blocks) * BLOCK + threadIdx.x;
        const uint32_t rr = r < A.rows ? r : A.rows - 1;           // keep the wave uniform; surplus lanes recompute the last row
        double o0, o1, o2;
        uint32_t unused = 0;
        Item I{};
        I.Y = (double)(A.y0 + (rr / A.blk_rows) * A.blk_stride + rr % A.blk_rows);
        I.xmin = 0.0; I.xmax = (double)(A.w - 1u); I.ymin = I.Y; I.ymax = I.Y;
        I.yout = A.yout + (size_t)rr * A.n_yvals;
        if (A.xtape) {
            I.yrow = A.consts;                                         // no y values in a ROW section: any readable address
            park_specials(A, slots, I);
            run_xtape<false, MODE_ROW>(A, A.xtape + A.job_off[job], A.job_len[job], nullptr, nullptr, slots, I, o0, o1, o2, unused);
        } else
            run_tape<false, MODE_ROW>(A, A.tape + A.job_off[job], A.job_len[job], nullptr, nullptr, slots, spill_base, spill_stride, I, o0, o1, o2, unused);
    }
}

// GUARDS kernel: one work-item per rectangle of guard_rows rows x 256 / guard_sub pixels, blockIdx.y = job (8 guards = one byte
// of the rectangle's guard bits; its tape is the cone of those guards: short jobs, many wavefronts -- each is one
// dependent chain).  XMIN..YMAX = the rectangle's ends.
__global__ void __launch_bounds__(BLOCK) maray_tape_guards(const KArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *slots = (double *)smem;
    const uint32_t spill_stride = gridDim.x * BLOCK;
    double *spill_base = A.spill ? A.spill + (size_t)blockIdx.x * BLOCK + threadIdx.x : nullptr;
    const uint32_t n_groups = (A.rows + A.guard_rows - 1) / A.guard_rows;
    const uint32_t per_row = A.tiles_per_row * A.guard_sub, gw = BLOCK / A.guard_sub;       // rectangles per row of them; their width
    const uint32_t n_items = n_groups * per_row;
    const uint32_t item_blocks = (n_items + BLOCK - 1) / BLOCK;
    // resident blocks draw (job, block of items) units from a queue, longest jobs first: the grid, and with it the spill
    // area, stays bounded for any image, and a block that drew a short job comes back for more while a long one runs.
    // Every block ends with a draw past the last unit.
    __shared__ uint32_t drawn;
    for (;;) {
        __syncthreads();                                               // the previous unit's reads of `drawn` are done
        if (threadIdx.x == 0) drawn = atomicAdd(A.queue, 1u);
        __syncthreads();
        const uint32_t u = drawn;
        if (u >= item_blocks * A.n_tiles /* = jobs */) break;
        const uint32_t job = u / item_blocks;
        const uint32_t it = (u - job * item_blocks) * BLOCK + threadIdx.x;
        const uint32_t item = it < n_items ? it : n_items - 1;         // keep the wave uniform
        const uint32_t grp = item / per_row, tx = item - grp * per_row;
        const uint32_t r = grp * A.guard_rows, r_last = r + A.guard_rows - 1 < A.rows - 1 ? r + A.guard_rows - 1 : A.rows - 1;
        // (the last tile of a ragged row may own rectangles past the edge: they bound the last pixel)
        const uint32_t xlo_ = tx * gw, xlo = xlo_ < A.w - 1 ? xlo_ : A.w - 1, xhi = xlo_ + gw - 1 < A.w - 1 ? xlo_ + gw - 1 : A.w - 1;
        double o0, o1, o2;
        uint32_t bits = 0;
        Item I{};
        // a group never straddles two row blocks (the host picks guard_rows | blk_rows): its image rows are consecutive
        I.Y = (double)(A.y0 + (r / A.blk_rows) * A.blk_stride + r % A.blk_rows);
        I.xmin = (double)xlo; I.xmax = (double)xhi; I.ymin = I.Y; I.ymax = I.Y + (double)(r_last - r);
        if (A.xtape) {
            I.yrow = A.consts;
            park_specials(A, slots, I);
            run_xtape<false, MODE_GUARDS>(A, A.xtape + A.job_off[job], A.job_len[job], nullptr, nullptr, slots, I, o0, o1, o2, bits);
        } else
            run_tape<false, MODE_GUARDS>(A, A.tape + A.job_off[job], A.job_len[job], nullptr, nullptr, slots, spill_base, spill_stride, I, o0, o1, o2, bits);
        if (it < n_items) ((unsigned char *)A.gbits)[(size_t)item * (A.guard_w32 * 4u) + A.job_id[job]] = (unsigned char)bits;
    }
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            throw Error{MARAY_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)};           \
    } while (0)

// PIXEL section -> xtape (see run_xtape).  guard_first: y values from this index on are read as guard bits by SKIP ops
// (0xFFFFFFFF: none are).
std::vector<uint64_t> predecode(const maray_program &P, const uint64_t *ops, uint32_t n_ops, uint32_t x_slot, uint32_t guard_first)
{
    std::vector<uint64_t> out(n_ops, 0);
    auto operand = [&](uint32_t ref, uint32_t &k, uint32_t &idx) {
        const uint32_t kind = MARAY_REF_KIND(ref), i = MARAY_REF_INDEX(ref);
        if (kind == MARAY_K_SLOT) { k = XS; idx = i; }
        else if (kind == MARAY_K_CONST) { k = XU; idx = i; }
        else if (kind == MARAY_K_YVAL) { k = XU; idx = 0x8000u | i; }
        else if (i == MARAY_SPEC_ACC) { k = XA; idx = 0; }
        else {
            static const uint32_t park[] = {0, 1, 0, 4, 3, 6, 5};          // X, Y, -, XMAX, XMIN, YMAX, YMIN -> reserved slot
            k = XS; idx = x_slot + park[i];
        }
    };
    for (uint32_t j = 0; j < n_ops; j++) {
        const uint64_t ins = ops[j];
        const uint32_t op = MARAY_INS_OP(ins), aux = MARAY_INS_AUX(ins);
        uint32_t dst = MARAY_INS_DST(ins);
        if (dst == MARAY_DST_NONE) dst = x_slot + 2;
        uint32_t ka = 0, ia = 0, kb = 0, ib = 0, x = X_NOP;
        if (op != MARAY_OP_NOP && op != MARAY_OP_TEXDIM) operand(MARAY_INS_A(ins), ka, ia);
        if (op >= MARAY_OP_ADD && op <= MARAY_OP_APP) operand(MARAY_INS_B(ins), kb, ib);
        switch (op) {
        case MARAY_OP_NOP: break;
        case MARAY_OP_ADD: x = X_BIN + 0 + 3 * ka + kb; break;
        case MARAY_OP_MUL: x = X_BIN + 9 + 3 * ka + kb; break;
        case MARAY_OP_MAX: x = X_BIN + 18 + 3 * ka + kb; break;
        case MARAY_OP_MIN: x = X_BIN + 27 + 3 * ka + kb; break;
        case MARAY_OP_NEG: x = X_UN + 0 + ka; break;
        case MARAY_OP_ABS: x = X_UN + 3 + ka; break;
        case MARAY_OP_STEP: x = X_UN + 6 + ka; break;
        case MARAY_OP_MOV: x = X_UN + 9 + ka; break;
        case MARAY_OP_RECIP: x = X_HEAVY + 0; ib = ka; break;
        case MARAY_OP_SQRT: x = X_HEAVY + 1; ib = ka; break;
        case MARAY_OP_SIN: x = X_HEAVY + 2; ib = ka; break;
        case MARAY_OP_STEPSIN: x = X_HEAVY + 3; ib = ka; break;
        case MARAY_OP_EXP: x = X_HEAVY + 4; ib = ka; break;
        case MARAY_OP_LN: x = X_HEAVY + 5; ib = ka; break;
        case MARAY_OP_APP: x = X_APP + 3 * ka + kb; break;
        case MARAY_OP_TEXDIM: x = X_TEXDIM; break;
        case MARAY_OP_OUT: x = X_OUT + ka; dst = 0; break;
        case MARAY_OP_SKIPZ: case MARAY_OP_SKIPNZ: {
            const uint32_t ref = MARAY_INS_A(ins);
            if (MARAY_REF_KIND(ref) == MARAY_K_YVAL && MARAY_REF_INDEX(ref) >= guard_first) { ka = XG; ia = MARAY_REF_INDEX(ref) - guard_first; }
            x = (op == MARAY_OP_SKIPZ ? X_SKIPZ : X_SKIPNZ) + ka;
            break;
        }
        default: break;
        }
        out[j] = MARAY_INS(x, aux, dst, ia, ib);
    }
    return out;
}

struct TapeBackend final : Backend {
    int device = 0;
    bool tape_lds = true;
    hipDeviceProp_t prop{};
    maray_program P{};
    uint64_t *d_row_ops = nullptr, *d_pix_ops = nullptr;
    double *d_consts = nullptr;
    MarayTex *d_tex = nullptr;
    std::vector<unsigned char *> d_tex_rgb;
    double *d_yvals = nullptr; size_t yvals_cap = 0;
    double *d_spill = nullptr; size_t spill_cap = 0;
    // Guards per rectangle of pixels (include/maray_tape.h, SPEC XMIN..YMIN): when no guard reads Y, the ROW tape is
    // cut into the cone of the operand y values (run per row) and the cones of the guards, 32 to a job (run per
    // rectangle of 32 rows x 64 pixels by maray_tape_guards); the pixel kernel then reads guard bits.
    bool tile_guards = false;
    uint32_t n_ynum = 0, n_guard_jobs = 0, n_guard_w32 = 0;
    uint32_t rows_slots = 0, guard_slots = 0;          // value slots of the cut tapes (renumbered by liveness)
    uint32_t guard_lds_slots = 0, guard_lds_bytes = 0;
    uint64_t *d_guard_ops = nullptr;
    uint32_t *d_job_off = nullptr, *d_job_len = nullptr;
    uint32_t *d_gbits = nullptr; size_t gbits_cap = 0;
    // PIXEL section pre-decoded for run_xtape: one variant reads guards as bits, one as y values (the drain)
    uint64_t *d_xtape_bits = nullptr, *d_xtape_rows = nullptr;
    uint32_t x_slot = 0;
    // ... and the ROW tape and the guard jobs (specials parked in 7 reserved slots after the tape's own)
    uint64_t *d_xrows = nullptr, *d_xguards = nullptr;
    uint32_t *d_row_job_off = nullptr, *d_row_job_len = nullptr, *d_job_id = nullptr, *d_queue = nullptr;
    uint32_t n_row_jobs = 0;
    uint32_t xrows_slot = 0, xguards_slot = 0;
    unsigned char *d_rgb8 = nullptr; size_t rgb8_cap = 0;      // time_rows without a caller's buffer
    std::unique_ptr<HostPipe> pipe;     // streams + staging of the host-raster entry points (from the device's pool: host_pipe.hpp)
    hipStream_t own_stream = nullptr;   // = pipe's compute stream
    hipStream_t last_stream = nullptr; bool have_last = false;      // the stream of the last launch (see launch())
    hipEvent_t handover = nullptr;
    // launch geometry of the pixel kernel
    uint32_t n_lds_slots = 0, lds_bytes = 0, blocks_per_cu = 1;
    uint32_t k_guard_h = 32, k_guard_sub = 4;           // MARAY_TAPE_GUARD_H / _W, read once in init()
    uint32_t row_lds_slots = 0, row_lds_bytes = 0;
    std::string kname;

    ~TapeBackend() override {
        (void)hipSetDevice(device);
        (void)hipFree(d_row_ops); (void)hipFree(d_pix_ops); (void)hipFree(d_consts); (void)hipFree(d_tex);
        for (auto p : d_tex_rgb) (void)hipFree(p);
        (void)hipFree(d_yvals); (void)hipFree(d_spill); (void)hipFree(d_rgb8);
        (void)hipFree(d_guard_ops); (void)hipFree(d_job_off); (void)hipFree(d_job_len); (void)hipFree(d_gbits);
        (void)hipFree(d_xtape_bits); (void)hipFree(d_xtape_rows); (void)hipFree(d_xrows); (void)hipFree(d_xguards);
        (void)hipFree(d_row_job_off); (void)hipFree(d_row_job_len); (void)hipFree(d_job_id); (void)hipFree(d_queue);
        if (handover) (void)hipEventDestroy(handover);
        if (pipe) { (void)hipStreamSynchronize(pipe->compute_stream()); host_pipe_release(std::move(pipe)); }
    }

    void init(int dev, const maray_program &prog, const maray_texture *tex, uint32_t n_tex, bool lds_variant) {
        device = dev;
        HIP_TRY(hipSetDevice(dev));
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            throw Error{MARAY_E_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only"};
        P = prog;
        P.consts = nullptr; P.row_ops = nullptr; P.pix_ops = nullptr;   // host pointers are borrowed for this call only
        pipe = host_pipe_acquire(dev);
        own_stream = pipe->compute_stream();
        HIP_TRY(hipEventCreateWithFlags(&handover, hipEventDisableTiming));
        auto up = [&](const void *src, size_t bytes, void **dst) {
            HIP_TRY(hipMalloc(dst, bytes ? bytes : 8));
            if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        };
        HIP_TRY(hipMalloc((void **)&d_queue, 8));               // work queues: [0] guards kernel, [1] pixel kernel
        n_ynum = numeric_yvals(prog);
        const uint32_t n_guards = prog.n_yvals - n_ynum;
        tile_guards = n_guards > 0 && prog.n_row_ops > 0 && !any_guard_reads_y(prog) && !getenv("MARAY_TAPE_ROW_GUARDS");
        if (const char *e_ = getenv("MARAY_TAPE_GUARD_H")) { const int v = atoi(e_); if (v == 8 || v == 16 || v == 32 || v == 64) k_guard_h = (uint32_t)v; }
        if (const char *e_ = getenv("MARAY_TAPE_GUARD_W")) { const int v = atoi(e_); if (v == 64 || v == 128 || v == 256) k_guard_sub = 256u / (uint32_t)v; }
        std::vector<uint64_t> rows_host, guards_host;
        const bool keep_order = getenv("MARAY_TAPE_KEEP_ORDER") != nullptr;
        const RowTapeDeps deps = row_tape_deps(prog);
        {
            // ROW kernel jobs: 8 y values each -- the ones PIXEL ops read as operands, and (when guards are evaluated
            // per row, as y values) the guards as well
            const uint32_t n_row_outs = tile_guards ? n_ynum : prog.n_yvals;
            n_row_jobs = prog.n_row_ops ? (n_row_outs + 7) / 8 : 0;
            std::vector<uint32_t> off, len;
            for (uint32_t j = 0; j < n_row_jobs; j++) {
                std::vector<uint32_t> outs;
                for (uint32_t o : deps.outs) {
                    const uint32_t aux = MARAY_INS_AUX(prog.row_ops[o]);
                    if (aux >= 8 * j && aux < 8 * (j + 1) && aux < n_row_outs) outs.push_back(o);
                }
                std::vector<uint64_t> t = compact_tape(row_tape_cone(prog, deps, outs, nullptr));
                rows_slots = std::max(rows_slots, keep_order ? renumber_slots(t) : reschedule_tape(t));
                if (t.empty()) continue;
                off.push_back((uint32_t)rows_host.size()); len.push_back((uint32_t)t.size());
                rows_host.insert(rows_host.end(), t.begin(), t.end());
            }
            n_row_jobs = (uint32_t)off.size();
            up(rows_host.data(), rows_host.size() * 8, (void **)&d_row_ops);
            up(off.data(), off.size() * 4, (void **)&d_row_job_off);
            up(len.data(), len.size() * 4, (void **)&d_row_job_len);
        }
        if (tile_guards) {
            n_guard_jobs = (n_guards + 7) / 8;
            n_guard_w32 = (n_guards + 31) / 32;
            std::vector<std::vector<uint64_t>> cones(n_guard_jobs);
            for (uint32_t j = 0; j < n_guard_jobs; j++) {
                std::vector<uint32_t> outs;
                for (uint32_t o : deps.outs) {
                    const uint32_t aux = MARAY_INS_AUX(prog.row_ops[o]);
                    if (aux >= n_ynum + 8 * j && aux < n_ynum + 8 * (j + 1)) outs.push_back(o);
                }
                std::vector<uint64_t> t = compact_tape(row_tape_cone(prog, deps, outs, nullptr));
                guard_slots = std::max(guard_slots, keep_order ? renumber_slots(t) : reschedule_tape(t));
                cones[j].swap(t);
            }
            std::vector<uint32_t> ids(n_guard_jobs), off(n_guard_jobs), len(n_guard_jobs);
            for (uint32_t j = 0; j < n_guard_jobs; j++) ids[j] = j;
            std::stable_sort(ids.begin(), ids.end(), [&](uint32_t x, uint32_t y) { return cones[x].size() > cones[y].size(); });
            std::vector<uint64_t> all;
            for (uint32_t k = 0; k < n_guard_jobs; k++) {
                off[k] = (uint32_t)all.size(); len[k] = (uint32_t)cones[ids[k]].size();
                all.insert(all.end(), cones[ids[k]].begin(), cones[ids[k]].end());
            }
            up(ids.data(), ids.size() * 4, (void **)&d_job_id);

            guards_host = all;
            up(all.data(), all.size() * 8, (void **)&d_guard_ops);
            up(off.data(), off.size() * 4, (void **)&d_job_off);
            up(len.data(), len.size() * 4, (void **)&d_job_len);
        }
        up(prog.pix_ops, (size_t)prog.n_pix_ops * 8, (void **)&d_pix_ops);
        up(prog.consts, (size_t)prog.n_consts * 8, (void **)&d_consts);
        std::vector<MarayTex> descs(n_tex ? n_tex : 1);
        for (uint32_t i = 0; i < n_tex; i++) {
            unsigned char *d = nullptr;
            up(tex[i].rgb, (size_t)tex[i].w * tex[i].h * 3, (void **)&d);
            d_tex_rgb.push_back(d);
            descs[i] = MarayTex{d, tex[i].w, tex[i].h};
        }
        up(descs.data(), descs.size() * sizeof(MarayTex), (void **)&d_tex);

        // LDS budget: 160 KiB per workgroup on gfx950
        const size_t lds_cap = 163840 - 64;                      // less the kernels' static LDS (the queue draw)
        size_t base = lds_variant ? ((size_t)prog.n_pix_ops + prog.n_consts) * 8 : 0;
        base = (base + 15) & ~(size_t)15;
        tape_lds = lds_variant;
        if (lds_variant && base + (size_t)BLOCK * 8 > lds_cap)
            throw Error{MARAY_E_LIMIT, "tape does not fit in LDS (" + std::to_string(base) + " bytes); use MARAY_BACKEND_TAPE_SMEM or MARAY_BACKEND_JIT"};
        const size_t slot_bytes = (size_t)BLOCK * 8;
        n_lds_slots = (uint32_t)std::min<size_t>(prog.n_pix_slots, (lds_cap - base) / slot_bytes);
        lds_bytes = (uint32_t)(base + (size_t)n_lds_slots * slot_bytes);
        blocks_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(8, lds_cap / std::max<uint32_t>(lds_bytes, 1)));
        row_lds_slots = (uint32_t)std::min<size_t>(rows_slots, 65536 / slot_bytes);
        row_lds_bytes = (uint32_t)(row_lds_slots * slot_bytes);
        guard_lds_slots = (uint32_t)std::min<size_t>(guard_slots, 40);       // 80 KB: two blocks per CU; the rest spills
        guard_lds_bytes = (uint32_t)(guard_lds_slots * slot_bytes);
        if (tile_guards) HIP_TRY(hipFuncSetAttribute((const void *)maray_tape_guards, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
        HIP_TRY(hipFuncSetAttribute((const void *)maray_tape_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
        if (!getenv("MARAY_TAPE_GENERIC")) {
            // ROW tape and guard jobs through the pre-decoded loop when all their slots (+ 7 parked specials) fit LDS
            if (!rows_host.empty() && ((size_t)rows_slots + 7) * slot_bytes <= lds_cap) {
                xrows_slot = rows_slots;
                row_lds_slots = rows_slots + 7;
                row_lds_bytes = (uint32_t)(row_lds_slots * slot_bytes);
                const std::vector<uint64_t> x = predecode(prog, rows_host.data(), (uint32_t)rows_host.size(), xrows_slot, 0xFFFFFFFFu);
                up(x.data(), x.size() * 8, (void **)&d_xrows);
            }
            if (tile_guards && ((size_t)guard_slots + 7) * slot_bytes <= lds_cap) {
                xguards_slot = guard_slots;
                guard_lds_slots = guard_slots + 7;
                guard_lds_bytes = (uint32_t)(guard_lds_slots * slot_bytes);
                const std::vector<uint64_t> x = predecode(prog, guards_host.data(), (uint32_t)guards_host.size(), xguards_slot, 0xFFFFFFFFu);
                up(x.data(), x.size() * 8, (void **)&d_xguards);
            }
        }
        // the pre-decoded loop needs every slot in LDS plus three (X, Y, trash)
        if (base + ((size_t)prog.n_pix_slots + 3) * slot_bytes <= lds_cap && prog.n_pix_slots + 3 < MARAY_DST_NONE && !getenv("MARAY_TAPE_GENERIC")) {
            x_slot = prog.n_pix_slots;
            n_lds_slots = prog.n_pix_slots + 3;
            lds_bytes = (uint32_t)(base + (size_t)n_lds_slots * slot_bytes);
            blocks_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(8, lds_cap / std::max<uint32_t>(lds_bytes, 1)));
            const std::vector<uint64_t> xr = predecode(prog, prog.pix_ops, prog.n_pix_ops, x_slot, 0xFFFFFFFFu);
            up(xr.data(), xr.size() * 8, (void **)&d_xtape_rows);
            if (tile_guards) {
                const std::vector<uint64_t> xb = predecode(prog, prog.pix_ops, prog.n_pix_ops, x_slot, n_ynum);
                up(xb.data(), xb.size() * 8, (void **)&d_xtape_bits);
            }
        }
        if (lds_variant) {
            HIP_TRY(hipFuncSetAttribute((const void *)maray_tape_pixels<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
            kname = "maray_tape_pixels<true>";
        } else {
            HIP_TRY(hipFuncSetAttribute((const void *)maray_tape_pixels<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
            kname = "maray_tape_pixels<false>";
        }
    }

    template <typename T>
    void ensure(T *&p, size_t &cap, size_t n) {
        if (n <= cap) return;
        if (p) HIP_TRY(hipFree(p));
        p = nullptr; cap = 0;
        HIP_TRY(hipMalloc((void **)&p, n * sizeof(T)));
        cap = n;
    }

    void launch(uint32_t w, const RowBlocks &rb, unsigned char *d8, double *d64, hipStream_t st, bool rows_pass,
                const unsigned *tile_list = nullptr, const double *ext_yvals = nullptr) {
        const uint32_t rows = rb.n_rows, y0 = rb.y0;
        if (!rows || !w) return;
        // scratch tables and work queues belong to the context: launches are ordered by their stream, and a launch on
        // another stream than the last one first waits for what that one holds (one event, only at the hand-over)
        if (have_last && st != last_stream) {
            HIP_TRY(hipEventRecord(handover, last_stream));
            HIP_TRY(hipStreamWaitEvent(st, handover, 0));
        }
        last_stream = st; have_last = true;
        HIP_TRY(hipMemsetAsync(d_queue, 0, 8, st));              // both kernels' work queues start at 0
        (void)hipGetLastError();        // the launches below are checked with hipGetLastError(): drop what an earlier, unrelated call left
        if (!ext_yvals) ensure(d_yvals, yvals_cap, (size_t)rows * std::max<uint32_t>(P.n_yvals, 1));
        if (ext_yvals) rows_pass = false;
        // guard bits per rectangle: only with this back-end's own ROW pass (a caller's y-value table carries the guards
        // as y values bounded over the row), and only if a group of rows never straddles two row blocks
        const bool bits = tile_guards && !ext_yvals;
        // The rectangle a guard is bounded over: 64 pixels x 32 rows, like the specialised path's (jit_source.cpp,
        // jit_guard_geom: as many rectangles as 256 x 8, closer to a shape's outline for a wavefront of 64 pixels);
        // fewer rows when a group would straddle two row blocks.  MARAY_TAPE_GUARD_W / _H (read at context creation): measurement knobs.
        uint32_t want_rows = k_guard_h, guard_sub = k_guard_sub;
        uint32_t guard_rows = 1;
        if (bits)
            for (uint32_t g = want_rows; g >= 8; g /= 2)
                if (rb.block_rows >= rows || rb.block_rows % g == 0) { guard_rows = g; break; }
        if (!bits) guard_sub = 1;
        const uint32_t tiles_per_row = (w + BLOCK - 1) / BLOCK;
        const uint32_t n_groups = (rows + guard_rows - 1) / guard_rows;
        if (bits) {
            const size_t had = gbits_cap;
            ensure(d_gbits, gbits_cap, (size_t)n_groups * tiles_per_row * guard_sub * n_guard_w32);
            if (gbits_cap != had) HIP_TRY(hipMemsetAsync(d_gbits, 0, gbits_cap * 4, st));     // bytes past the last job are never written
        }
        if (rows_pass && (n_row_jobs || bits)) {
            KArgs R{};
            R.tape = d_row_ops; R.consts = d_consts; R.yout = d_yvals; R.tex = d_tex;
            R.n_ops = 0; R.n_tiles = n_row_jobs; R.job_off = d_row_job_off; R.job_len = d_row_job_len;
            R.n_consts = P.n_consts; R.n_yvals = P.n_yvals;
            R.n_slots = rows_slots; R.n_lds_slots = row_lds_slots;
            R.xtape = d_xrows; R.x_slot = xrows_slot;
            R.w = w; R.y0 = y0; R.rows = rows; R.blk_rows = rb.block_rows; R.blk_stride = rb.block_stride;
            const uint64_t row_units = (uint64_t)((rows + BLOCK - 1) / BLOCK) * n_row_jobs;
            const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(row_units, 1), (uint64_t)prop.multiProcessorCount * 4);
            if (n_row_jobs && rows_slots > row_lds_slots) {
                ensure(d_spill, spill_cap, std::max(spill_cap, (size_t)(rows_slots - row_lds_slots) * grid * BLOCK));
                R.spill = d_spill;
            }
            if (n_row_jobs) {
                hipLaunchKernelGGL(maray_tape_rows, dim3(grid), dim3(BLOCK), row_lds_bytes, st, R);
                HIP_TRY(hipGetLastError());
            }
            if (bits) {
                KArgs G = R;
                G.tape = d_guard_ops; G.yout = nullptr;
                G.gbits = d_gbits; G.job_off = d_job_off; G.job_len = d_job_len;
                G.guard_first = n_ynum; G.guard_w32 = n_guard_w32; G.guard_rows = guard_rows; G.tiles_per_row = tiles_per_row; G.guard_sub = guard_sub;
                const uint64_t items = (uint64_t)n_groups * tiles_per_row * guard_sub;
                if (items > 0x7FFFFFFFull) throw Error{MARAY_E_ARG, "too many tiles in one launch; render fewer rows per call"};
                const uint64_t units = ((items + BLOCK - 1) / BLOCK) * n_guard_jobs;        // (job, block of items) pairs
                const uint32_t resident = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, 163840 / std::max<uint32_t>(guard_lds_bytes, 1)));
                const uint32_t ggrid = (uint32_t)std::min<uint64_t>(units, (uint64_t)prop.multiProcessorCount * resident);
                G.queue = d_queue; G.job_id = d_job_id;
                G.spill = nullptr;
                G.n_slots = guard_slots; G.n_lds_slots = guard_lds_slots;
                G.xtape = d_xguards; G.x_slot = xguards_slot;
                G.n_tiles = n_guard_jobs;                                                       // the GUARDS kernel reads its job count here
                if (guard_slots > guard_lds_slots) {
                    ensure(d_spill, spill_cap, std::max(spill_cap, (size_t)(guard_slots - guard_lds_slots) * ggrid * BLOCK));
                    G.spill = d_spill;
                }
                hipLaunchKernelGGL(maray_tape_guards, dim3(ggrid), dim3(BLOCK), guard_lds_bytes, st, G);
                HIP_TRY(hipGetLastError());
            }
        }
        KArgs A{};
        A.tape = d_pix_ops; A.consts = d_consts; A.yvals = ext_yvals ? ext_yvals : d_yvals; A.tex = d_tex;
        A.tile_list = tile_list;
        A.rgb8 = d8; A.rgb64 = d64;
        A.n_ops = P.n_pix_ops; A.n_consts = P.n_consts; A.n_yvals = P.n_yvals;
        A.n_slots = P.n_pix_slots; A.n_lds_slots = n_lds_slots;
        A.w = w; A.y0 = y0; A.rows = rows; A.blk_rows = rb.block_rows; A.blk_stride = rb.block_stride;
        A.tiles_per_row = tiles_per_row;
        if (bits) { A.gbits = d_gbits; A.guard_first = n_ynum; A.guard_w32 = n_guard_w32; A.guard_rows = guard_rows; A.guard_sub = guard_sub; }
        A.xtape = bits ? d_xtape_bits : d_xtape_rows;
        A.queue = d_queue + 1;
        A.x_slot = x_slot;
        const uint64_t tiles = (uint64_t)A.tiles_per_row * rows;
        if (tiles > 0xFFFFFFFFull) throw Error{MARAY_E_ARG, "too many tiles in one launch; render fewer rows per call"};
        A.n_tiles = (uint32_t)tiles;
        const uint32_t per_cu = blocks_per_cu;
        const uint32_t grid = (uint32_t)std::min<uint64_t>(tiles, (uint64_t)prop.multiProcessorCount * per_cu);
        if (P.n_pix_slots > n_lds_slots) {
            ensure(d_spill, spill_cap, std::max(spill_cap, (size_t)(P.n_pix_slots - n_lds_slots) * grid * BLOCK));
            A.spill = d_spill;
        }
        if (tape_lds) hipLaunchKernelGGL(maray_tape_pixels<true>, dim3(grid), dim3(BLOCK), lds_bytes, st, A);
        else hipLaunchKernelGGL(maray_tape_pixels<false>, dim3(grid), dim3(BLOCK), lds_bytes, st, A);
        HIP_TRY(hipGetLastError());
    }

    void render_device(uint32_t w, uint32_t, const RowBlocks &rb, void *d8, void *d64, void *stream) override {
        HIP_TRY(hipSetDevice(device));
        launch(w, rb, (unsigned char *)d8, (double *)d64, (hipStream_t)stream, true);
    }

    void render_flagged(uint32_t w, const RowBlocks &rb, void *d8, void *d64, void *stream, const unsigned *flags,
                        const double *yvals) override {
        HIP_TRY(hipSetDevice(device));
        launch(w, rb, (unsigned char *)d8, (double *)d64, (hipStream_t)stream, false, flags, yvals);
    }

    void render_host_tiles(uint32_t w, uint32_t, const std::vector<RowTile> &tiles, uint32_t row0, uint8_t *rgb8, double *rgb64,
                           const std::function<void(uint32_t, uint32_t)> &done) override {
        HIP_TRY(hipSetDevice(device));
        pipe->run(w, tiles, row0, rgb8, rgb64,
                 [&](const RowBlocks &rb, unsigned char *d8, double *d64, hipStream_t st) { launch(w, rb, d8, d64, st, true); }, done);
    }

    float time_rows(uint32_t w, uint32_t h, const RowBlocks &rb, void *d8, void *d64, int reps) override {
        HIP_TRY(hipSetDevice(device));
        (void)h;
        const size_t n = (size_t)rb.n_rows * w * 3;
        unsigned char *p8 = (unsigned char *)d8;
        double *p64 = (double *)d64;
        if (!p8 && !p64) { ensure(d_rgb8, rgb8_cap, n); p8 = d_rgb8; }
        launch(w, rb, p8, p64, own_stream, true);   // warm-up + y values
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, own_stream));
        for (int i = 0; i < reps; i++) launch(w, rb, p8, p64, own_stream, false);   // pixel kernel only
        HIP_TRY(hipEventRecord(e1, own_stream));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        return ms / (float)(reps > 0 ? reps : 1);
    }

    const char *kernel_name() const override { return kname.c_str(); }
};

}   // namespace

int hip_device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

Backend *make_tape_backend(int device, const maray_program &prog, const maray_texture *tex, uint32_t n_tex, bool lds_variant)
{
    if (hip_device_count() <= 0) throw Error{MARAY_E_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)"};
    auto *b = new TapeBackend();
    try {
        b->init(device, prog, tex, n_tex, lds_variant);
    } catch (...) {
        delete b;
        throw;
    }
    return b;
}

}   // namespace maray
