// jit_source.cpp — tape -> HIP source of the two specialised kernels (product code).  The GPU analogue of the
// reference's wasmer JIT front half:
//
//   jit_source / jit_source_rows   <- wasm::gen_expr / gen_vars / Wasm::from_expr   src/wasm.rs:77-158
//                                     (Expr -> WAT text -> compiled module; one `(local $id f64)` per Let variable)
//
// Differences by design: the input is the lowered tape (one DAG shared by R, G and B, constants folded, Y-only work
// hoisted into a per-row kernel), every op is an inline f64 instruction (the reference's JIT calls host imports for
// abs/recip/step/sin/exp/ln, src/wasm.rs:40-47), max/min follow the interpreter (f64::max/min), not wasm's
// NaN-propagating f64.max/min (src/wasm.rs:58-59).
//
// Generated code: every tape op becomes `const double vN = op(...)`; value slots and ACC disappear (the compiler
// allocates registers), constants become exact hex-float literals or entries of a constant table, y values are scalar
// loads from the row table.  Compiled with -ffp-contract=off so no a*b+c is fused behind the tape's back.
#include "jit_emit.hpp"

namespace maray {

// Can the pixel kernel defer tiles to the interpreter?  Only a Sin / Step(Sin) whose argument is not proven bounded can.
bool may_defer_tiles(const maray_program &P)
{
    for (uint32_t i = 0; i < P.n_pix_ops; i++) {
        const uint32_t op = MARAY_INS_OP(P.pix_ops[i]);
        if ((op == MARAY_OP_SIN || op == MARAY_OP_STEPSIN) && !(MARAY_INS_AUX(P.pix_ops[i]) & MARAY_AUX_SIN_BOUNDED)) return true;
    }
    return false;
}

// Which y values are booleans: a dry run of the emitter over the ROW section (its typing is the one the PIXEL section
// will rely on).
std::vector<uint8_t> jit_bool_yvals(const maray_program &P)
{
    std::vector<uint8_t> r(P.n_yvals, 0);
    if (!P.n_row_ops) return r;
    Emitter D(P);
    D.section(P.row_ops, P.n_row_ops, P.n_row_slots, false, "r");
    const uint32_t n_ynum = numeric_yvals(P);
    for (uint32_t k = 0; k < P.n_yvals && k < D.out_is_bool.size() && k < n_ynum; k++) r[k] = D.out_is_bool[k];
    return r;
}

// The guard plan of a program (GuardPlan).  A guard is derived when its source is a MAX tree, boolean-typed all the way
// (on {+0.0, 1.0} max is OR, so "value != 0" distributes over it exactly), whose leaves are sources of other guards.
// Bits are handed out in the order the members are met, so that a group's bits are neighbours (one word, one s_and).
GuardPlan jit_guard_plan(const maray_program &P)
{
    GuardPlan gp;
    const uint32_t n_ynum = numeric_yvals(P), n_guards = P.n_yvals - n_ynum;
    gp.pos.assign(n_guards, -1);
    gp.members.assign(n_guards, {});
    const bool derive = true;
    const RowTapeDeps d = row_tape_deps(P);
    std::vector<int32_t> src(n_guards, -1);                    // op that produces a guard's value
    std::unordered_map<int32_t, uint32_t> guard_of;            // op -> (first) guard it is the source of
    for (uint32_t o : d.outs) {
        const uint32_t aux = MARAY_INS_AUX(P.row_ops[o]);
        if (aux < n_ynum) continue;
        src[aux - n_ynum] = d.deps[o][0];
        if (d.deps[o][0] >= 0) guard_of.emplace(d.deps[o][0], aux - n_ynum);
    }
    std::vector<uint8_t> isb;
    if (derive && P.n_row_ops) {
        Emitter D(P);
        D.section(P.row_ops, P.n_row_ops, P.n_row_slots, false, "r");
        isb = D.is_bool_op;
    }
    // leaves of guard g's MAX tree; false if it is not one
    std::vector<std::vector<uint32_t>> kids(n_guards);
    std::vector<uint8_t> derived(n_guards, 0);
    for (uint32_t g = 0; g < n_guards && derive; g++) {
        const int32_t s0 = src[g];
        if (s0 < 0 || MARAY_INS_OP(P.row_ops[s0]) != MARAY_OP_MAX || !isb[s0]) continue;
        std::vector<int32_t> st = {d.deps[s0][0], d.deps[s0][1]};
        std::vector<uint32_t> leaves;
        bool ok = true;
        while (ok && !st.empty()) {
            const int32_t o = st.back(); st.pop_back();
            if (o < 0) { ok = false; break; }
            auto it = guard_of.find(o);
            if (it != guard_of.end() && it->second != g) { leaves.push_back(it->second); continue; }
            if (MARAY_INS_OP(P.row_ops[o]) == MARAY_OP_MAX && isb[o]) { st.push_back(d.deps[o][1]); st.push_back(d.deps[o][0]); continue; }
            ok = false;
        }
        if (ok && !leaves.empty() && leaves.size() <= 64) { derived[g] = 1; kids[g] = leaves; }
    }
    // bits: members of a derived guard first, in tree order (recursively: a member may be derived itself), then the rest
    std::function<void(uint32_t, std::vector<uint32_t> &)> place = [&](uint32_t g, std::vector<uint32_t> &into) {
        if (derived[g]) {
            if (gp.members[g].empty()) for (uint32_t k : kids[g]) place(k, gp.members[g]);
            into.insert(into.end(), gp.members[g].begin(), gp.members[g].end());
            return;
        }
        if (gp.pos[g] < 0) gp.pos[g] = (int32_t)gp.n_pos++;
        into.push_back((uint32_t)gp.pos[g]);
    };
    std::vector<uint32_t> sink;
    // dearest groups first: their members end up contiguous
    std::vector<uint32_t> order(n_guards);
    for (uint32_t g = 0; g < n_guards; g++) order[g] = g;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return kids[a].size() > kids[b].size(); });
    for (uint32_t g : order) { sink.clear(); place(g, sink); }
    return gp;
}

// The ROW section split into chunks that different wavefronts evaluate side by side.  One
// work-item per row is all the parallelism a straight-line ROW kernel has (4096 rows = 64 waves,
// each walking thousands of dependent f64 ops: ~45 us for chess, an eighth of the frame).  The y
// values are independent outputs, so the tape is cut by outputs: chunk k keeps the ops its outputs
// depend on (ops two chunks share are computed in both) and the rest become NOPs.
// Chunk k writes the y values [first[k], first[k] + count[k]): outputs are taken in index order, so that a chunk's
// values are neighbours in a row of the table and leave the kernel as full cache lines (jit_source_rows).
struct RowChunks {
    std::vector<std::vector<uint64_t>> tapes;
    std::vector<uint32_t> first, count;
};
static const uint32_t ROW_CHUNK_MAX_OUTS = 16;      // x 68 x 8 B of LDS per wavefront

RowChunks split_row_tape(const maray_program &P, const RowTapeDeps &d, uint32_t out_limit)
{
    const uint32_t n = P.n_row_ops;
    std::vector<uint32_t> outs;
    for (uint32_t j : d.outs) if (MARAY_INS_AUX(P.row_ops[j]) < out_limit) outs.push_back(j);
    std::sort(outs.begin(), outs.end(), [&](uint32_t x, uint32_t y) { return MARAY_INS_AUX(P.row_ops[x]) < MARAY_INS_AUX(P.row_ops[y]); });
    size_t total = 0;
    (void)row_tape_cone(P, d, outs, &total);
    const size_t per_chunk = 64;
    const uint32_t n_chunks = (uint32_t)std::min<size_t>(64, std::max<size_t>(std::max<size_t>(1, total / per_chunk), (outs.size() + ROW_CHUNK_MAX_OUTS - 1) / ROW_CHUNK_MAX_OUTS));
    const size_t budget = (total + n_chunks - 1) / n_chunks;
    RowChunks rc;
    size_t next = 0;
    while (next < outs.size()) {
        std::vector<uint32_t> mine;
        size_t cost = 0;
        while (next < outs.size() && mine.size() < ROW_CHUNK_MAX_OUTS && cost < budget) {
            // a y value with no OUT of its own between two others would break the chunk's index range: a new chunk starts there
            if (!mine.empty() && MARAY_INS_AUX(P.row_ops[outs[next]]) != MARAY_INS_AUX(P.row_ops[mine.back()]) + 1) break;
            mine.push_back(outs[next++]);
            (void)row_tape_cone(P, d, mine, &cost);
        }
        rc.first.push_back(MARAY_INS_AUX(P.row_ops[mine.front()]));
        rc.count.push_back((uint32_t)mine.size());
        rc.tapes.push_back(row_tape_cone(P, d, mine, &cost));
    }
    if (rc.tapes.empty()) { rc.tapes.emplace_back(n, 0); rc.first.push_back(0); rc.count.push_back(0); }
    return rc;
}

// How the specialised kernels use the row guards of a program: as bits, 64 per word, one set per
// 256-pixel tile of a row (guard_words = 0: not at all -- none, far too many, or switched off).  Up to 12 words a tile's
// words sit in SGPRs; beyond, a guard test reads its word from LDS.
uint32_t jit_guard_words(const maray_program &P)
{
    if (!jit_row_guards_enabled() || P.n_yvals == numeric_yvals(P)) return 0;
    const uint32_t nw = (jit_guard_plan(P).n_pos + 63) / 64;
    return nw <= 1024 ? std::max(nw, 1u) : 0;       // 1024 words x 8 tiles = 64 KB of LDS
}


// The rectangle a guard is bounded over: `gh` rows x `gw` pixels.  gh = 1 (and gw = 256) when some guard's cone reads Y;
// else every guard bounds its boolean over the rows [YMIN, YMAX] too (include/maray_tape.h) and the rectangle is the
// back-end's choice.  The number of rectangles is what the ROW kernel pays for, their shape is what the PIXEL kernel
// gains from: a shape's edge is met by ~(extent / side + 1) rectangles each way, a wavefront enters regions per 64
// pixels of ONE row, and shapes are tall against 8 rows -- so a rectangle that is narrower and taller by the same factor
// costs the ROW kernel nothing and spares the PIXEL kernel region entries.  Default 64 x 32 (chess @4096^2, frame / board
// crop in us, 256 x 8: 49.3 / 104; 256 x 16: 48.9 / 104; 128 x 16: 45.1 / 92; 64 x 8: 49.0 / 83 -- four times the guard work;
// 64 x 16: 45.5 / 84; 64 x 32: 43.8 / 84; 64 x 64: 45.1 / 86; 64 x 128: 48.1 / 88).  MARAY_JIT_GUARD_W = 64 / 128 / 256,
// MARAY_JIT_GUARD_H = 8 ... 128: measurement knobs.  A strip's words are held one per lane, so a tile's rectangles together
// have to fit a wavefront's 64 lanes: a program with many guard words gets wider rectangles.
GuardGeom jit_guard_geom(const maray_program &P)
{
    GuardGeom g{256u, 1u};
    const uint32_t nw = jit_guard_words(P);
    if (!nw || any_guard_reads_y(P)) return g;
    g.gh = 32u;
    const char *env_h = getenv("MARAY_JIT_GUARD_H");
    if (env_h) { const int v = atoi(env_h); if (v == 8 || v == 16 || v == 32 || v == 64 || v == 128) g.gh = (uint32_t)v; }
    uint32_t want = 64u;
    if (const char *e_ = getenv("MARAY_JIT_GUARD_W")) { const int v = atoi(e_); if (v == 64 || v == 128 || v == 256) want = (uint32_t)v; }
    while (want < 256u && nw * (256u / want) > 64u) want *= 2u;
    if (want == 256u && !env_h) g.gh = 8u;      // wide rectangles gain nothing from height (chess, 256 x 8 / 256 x 32: 48.6 / 50.2 us per frame)
    g.gw = want;
    return g;
}


// Source of the ROW kernel, maray_jit_rows: one wavefront per block, blockIdx.y picks the job.
//  y < n_chunks: chunk y of the ROW section, one work-item per row; writes the y values the pixel
//    kernel reads as operands (and, for a program that may defer tiles to the interpreter, the
//    guards too, bounded over the whole row as the interpreter expects).
//  y >= n_chunks: guards 8 (y - n_chunks) .. +7, one work-item per rectangle (group of `yrows` rows,
//    run of jit_guard_geom().gw pixels), evaluated with XMIN / XMAX = the run's ends and YMIN / YMAX = the group's
//    (a bound over a rectangle skips far more than one over the row); writes its byte of
//    the rectangle's guard words (64 guards per word).  yrows = 1 when some guard reads Y.  Small
//    jobs on purpose: each is one long dependent chain, and only more wavefronts hide that.
// One launch for both: the few y-value wavefronts run in the shadow of the guard ones.
// Plain device_math.h: the rare huge-argument tail of sin is a real (out-of-line) call here.
std::string jit_source_rows(const maray_program &P, uint32_t *n_chunks_out, uint32_t *n_gjobs_out)
{
    validate_program(P);
    Emitter E(P);
    const RowTapeDeps deps = row_tape_deps(P);
    const uint32_t n_ynum = numeric_yvals(P), n_gwords = jit_guard_words(P);
    const GuardPlan plan = jit_guard_plan(P);
    const GuardGeom geom = jit_guard_geom(P);
    // wave-level SKIP ops of the ROW section: a wavefront's lanes are 64 rows, or the 64 rectangles of a band of rows, and
    // agree on the sky only; a job is one wavefront's chain, and every short region it has to test and branch around
    // lengthens it (chess, step minus pixel kernel in us, regions kept from 0 / 12 / 24 / 60 / 200 instructions / none:
    // 7.2 / 6.9 / 6.3 / 6.8 / 8.3 / 8.3)
    E.min_region_row = 24;
    const uint32_t n_gjobs = n_gwords ? (plan.n_pos + 7) / 8 : 0;                // 8 bits = one byte of a word per job
    if (n_gjobs_out) *n_gjobs_out = n_gjobs;
    // the interpreter (which drains deferred tiles from the same y-value table) does read the guard values
    const uint32_t out_limit = may_defer_tiles(P) ? 0xFFFFFFFFu : n_ynum;
    const RowChunks rc = split_row_tape(P, deps, out_limit);
    const std::vector<std::vector<uint64_t>> &chunks = rc.tapes;
    if (n_chunks_out) *n_chunks_out = (uint32_t)chunks.size();
    std::string &s = E.out;
    s += "// generated by libmaray_hip (jit_source.cpp): ROW section, " + std::to_string(P.n_row_ops) + " ops; y values in " +
         std::to_string(chunks.size()) + " chunks, " + std::to_string(P.n_yvals - n_ynum) + " guards in " + std::to_string(n_gwords) + " words,\n"
         "// each bounded over rectangles of " + std::to_string(geom.gw) + " pixels x " + std::to_string(geom.gh) + " rows (the height is a launch parameter, and part of the code key through this line:\n"
         "// a cached code object carries its geometry)\n";
    s += "#include \"device_math.h\"\n\n";
    // (constants stay literals here: from a table in constant memory like the PIXEL kernel's, the code is a tenth shorter
    // and the kernel 0.8 us slower -- the loads' waits sit in the one chain a job is -- and spills to scratch)
    const unsigned row_block = ROW_BLOCK;
    s += "extern \"C\" __global__ void __launch_bounds__(" + std::to_string(row_block) + ") maray_jit_rows(double *__restrict__ yvals, unsigned long long *__restrict__ gbits,\n"
         "                                                                 const MarayTex *__restrict__ tex,\n"
         "                                                                 unsigned y0, unsigned rows, unsigned n_yvals, unsigned w, unsigned n_tx,\n"
         "                                                                 unsigned blk_rows, unsigned blk_stride, unsigned yrows)\n{\n"
         "    const unsigned item = blockIdx.x * blockDim.x + threadIdx.x;         // (the host keeps the items of a launch below 2^32)\n"
         "    (void)tex; (void)gbits; (void)n_tx;\n"
         "    // guard jobs first in the grid (they are the long ones: the y-value jobs fill in behind them): job = the switch index\n"
         "    const unsigned mr_job = blockIdx.y < " + std::to_string(n_gjobs) + "u ? " + std::to_string(chunks.size()) + "u + blockIdx.y : blockIdx.y - " + std::to_string(n_gjobs) + "u;\n"
         "    if (mr_job < " + std::to_string(chunks.size()) + "u) {\n"
         +
         "    // y values: a work-item per row.  A lane's values go to LDS ([value][row], values 68 apart) and leave as rows of the\n"
         "    // table, a chunk's values side by side: full cache lines.  Stored from the registers, a\n"
         "    // wavefront's store touches 64 lines for 8 bytes each -- 1.2 M partial writes per frame, which is what the kernel\n"
         "    // then waits for (11.8 us; the arithmetic needs 2).\n"
         "    __shared__ double mr_ys[" + std::to_string(row_block / 64) + " * " + std::to_string(ROW_CHUNK_MAX_OUTS * 68) + "];\n"
         "    const unsigned mr_lane = threadIdx.x & 63u;\n"
         "    double *ys = mr_ys + (threadIdx.x >> 6) * " + std::to_string(ROW_CHUNK_MAX_OUTS * 68) + "u;\n"
         "    const unsigned row0 = item - mr_lane;                                 // first row of this wavefront\n"
         "    if (row0 >= rows) return;                                            // whole wavefronts only: every lane helps to store\n"
         "    const unsigned r = item;\n"
         "    const double Y = (double)(blk_stride == 0u ? y0 + r : y0 + (r / blk_rows) * blk_stride + r % blk_rows), XMIN = 0.0, XMAX = (double)(w - 1u);\n"
         "    const double YMIN = Y, YMAX = Y;\n"
         "    (void)Y; (void)XMIN; (void)XMAX; (void)YMIN; (void)YMAX; (void)yrows;\n"
         "    unsigned mr_k0 = 0u, mr_kn = 0u;\n"
         "    switch (mr_job) {\n";
    for (size_t k = 0; k < chunks.size(); k++) {
        s += "    case " + std::to_string(k) + ": {\n";
        E.stage_first = (int)rc.first[k];
        E.section(chunks[k].data(), P.n_row_ops, P.n_row_slots, false, "r");
        E.stage_first = -1;
        s += "    mr_k0 = " + std::to_string(rc.first[k]) + "u; mr_kn = " + std::to_string(rc.count[k]) + "u;\n    } break;\n";
    }
    s += "    }\n"
         "    __builtin_amdgcn_wave_barrier();                                       // same wavefront: LDS keeps its order\n"
         "    // lane 16 q + j stores value j of the rows 4 i + q, i = 0 .. 15: an instruction writes four rows of the chunk\n"
         "    const unsigned mr_j = mr_lane & 15u, mr_q = mr_lane >> 4;\n"
         "    double *mr_dst = yvals + (size_t)(row0 + mr_q) * n_yvals + mr_k0 + mr_j;\n"
         "    const size_t mr_step = (size_t)4u * n_yvals;\n"
         "    if (mr_j < mr_kn) {\n"
         "        _Pragma(\"unroll\") for (unsigned i = 0; i < 16u; i++)\n"
         "            if (row0 + 4u * i + mr_q < rows) mr_dst[i * mr_step] = ys[mr_j * 68u + 4u * i + mr_q];\n"
         "    }\n"
         "    return;\n    }\n";
    if (n_gwords) {
        s += "    // guards: (row group, tile), the tiles of a group adjacent\n"
             "    const unsigned n_groups = (rows + yrows - 1u) / yrows;\n"
             "    if (item >= n_groups * n_tx) return;\n"
             "    const unsigned grp = item / n_tx, tile = item - grp * n_tx;\n"
             "    const unsigned r = grp * yrows, r_last = r + yrows - 1u < rows - 1u ? r + yrows - 1u : rows - 1u;      // launch rows of the group\n"
             "    // (n_tx counts rectangles here; the last 256-pixel tile of a ragged row may own rectangles past the edge: they bound the last pixel)\n"
             "    const unsigned xlo_ = tile * " + std::to_string(geom.gw) + "u, xlo = xlo_ < w - 1u ? xlo_ : w - 1u, xhi = xlo_ + " + std::to_string(geom.gw - 1) + "u < w - 1u ? xlo_ + " + std::to_string(geom.gw - 1) + "u : w - 1u;\n"
             "    // a group never straddles two row blocks (the host picks yrows | blk_rows), so its image rows are consecutive\n"
             "    const double Y = (double)(blk_stride == 0u ? y0 + r : y0 + (r / blk_rows) * blk_stride + r % blk_rows), XMIN = (double)xlo, XMAX = (double)xhi;\n"
             "    const double YMIN = Y, YMAX = Y + (double)(r_last - r);\n"
             "    unsigned long long gacc = 0ull;\n"
             "    double *yout = nullptr;\n"
             "    (void)Y; (void)XMIN; (void)XMAX; (void)YMIN; (void)YMAX; (void)yout;\n"
             "    switch (mr_job - " + std::to_string(chunks.size()) + "u) {\n";
        E.out_guard_bits = true;
        E.guard_first = n_ynum;
        E.plan = &plan;
        for (uint32_t j = 0; j < n_gjobs; j++) {
            std::vector<uint32_t> outs;          // the guards whose bits are 8 j .. 8 j + 7 (derived guards have none: no job computes them)
            for (uint32_t o : deps.outs) {
                const uint32_t aux = MARAY_INS_AUX(P.row_ops[o]);
                if (aux < n_ynum) continue;
                const int32_t k = plan.pos[aux - n_ynum];
                if (k >= (int32_t)(8 * j) && k < (int32_t)(8 * (j + 1))) outs.push_back(o);
            }
            const std::vector<uint64_t> tape = row_tape_cone(P, deps, outs, nullptr);
            s += "    case " + std::to_string(j) + ": {\n";
            E.section(tape.data(), P.n_row_ops, P.n_row_slots, false, "r");
            s += "    } break;\n";
        }
        s += "    }\n"
             "    ((unsigned char *)gbits)[(size_t)item * " + std::to_string(8 * n_gwords) + "u + (mr_job - " + std::to_string(chunks.size()) + "u)] = (unsigned char)gacc;\n";
    }
    s += "}\n";
    // Launch order of the PIXEL kernel: groups of rows by what they cost, dearest first, so that the tail of the launch is
    // made of cheap blocks.  Cost of a group = set bits in the guard words of its rectangles (shapes that may show there).
    // The bits are a function of the program and of the launch's geometry only: the order is computed once per geometry
    // (one block; rank by counting) and reused.  Rows of a group stay neighbours (they share guard words and cache lines).
    if (n_gwords)
        s += "extern \"C\" __global__ void __launch_bounds__(256) maray_jit_order(const unsigned long long *__restrict__ gbits, unsigned *__restrict__ order,\n"
             "                                                                  unsigned rows, unsigned n_tx, unsigned yrows)\n{\n"
             "    extern __shared__ unsigned mr_cost[];\n"
             "    const unsigned n_groups = (rows + yrows - 1u) / yrows, n_full = rows / yrows, per = n_tx * " + std::to_string(n_gwords) + "u;\n"
             "    for (unsigned g = threadIdx.x; g < n_groups; g += 256u) {\n"
             "        unsigned c = 0;\n"
             "        for (unsigned i = 0; i < per; i++) c += (unsigned)__builtin_popcountll(gbits[(size_t)g * per + i]);\n"
             "        mr_cost[g] = c;\n"
             "    }\n"
             "    __syncthreads();\n"
             "    for (unsigned g = threadIdx.x; g < n_full; g += 256u) {\n"
             "        const unsigned c = mr_cost[g];\n"
             "        unsigned rank = 0;\n"
             "        for (unsigned h = 0; h < n_full; h++) rank += (mr_cost[h] > c || (mr_cost[h] == c && h < g)) ? 1u : 0u;\n"
             "        for (unsigned i = 0; i < yrows; i++) order[rank * yrows + i] = g * yrows + i;\n"
             "    }\n"
             "    for (unsigned r = n_full * yrows + threadIdx.x; r < rows; r += 256u) order[r] = r;      // a partial last group stays last\n"
             "}\n";
    return s;
}

// The general section four pixels per lane: only a short program without guards whose ops are single instructions (no libm
// bodies, no gathers).
bool jit_wide_general(const maray_program &P, uint32_t n_gwords)        // n_gwords = jit_guard_words(P) (a walk over the ROW tape: the caller has it)
{
    bool heavy = false;
    // texture lookups do not keep a small program from the four-wide form (round 4: four gathers per lane in flight, the scalar
    // unit's share paid once per 256 pixels: config 5 41.5 -> 39.3 us); MARAY_JIT_WIDE_APP=0: one pixel per lane (ablation)
    const char *e_ = getenv("MARAY_JIT_WIDE_APP");
    const bool wide_app = !(e_ && e_[0] == '0');
    for (uint32_t i = 0; i < P.n_pix_ops; i++) {
        const uint32_t op = MARAY_INS_OP(P.pix_ops[i]);
        heavy |= op == MARAY_OP_SIN || op == MARAY_OP_EXP || op == MARAY_OP_LN || op == MARAY_OP_STEPSIN || (op == MARAY_OP_APP && !wide_app);
    }
    return n_gwords == 0 && !heavy && P.n_pix_slots <= (wide_app ? 12u : 6u) && P.n_pix_ops <= 256;
}

// Two rows per wavefront in the busy tiles (jit_source): a program whose shapes are guarded per rectangle of an even number
// of rows, evaluated one pixel per lane.  MARAY_JIT_ROWS2=0 / 1: off / on (ablation; part of the code key like every MARAY_JIT_*).
bool jit_rows2(const maray_program &P)
{
    const char *e_ = getenv("MARAY_JIT_ROWS2");
    if (!(e_ && e_[0] == '1')) return false;
    const uint32_t nw = jit_guard_words(P);
    if (!nw || nw > GW_INLINE_MAX || jit_wide_general(P, nw)) return false;
    const GuardGeom g = jit_guard_geom(P);
    return g.gh >= 2 && g.gh % 2 == 0;
}

// Source of the PIXEL kernel, maray_jit_pixels.  A wavefront owns a strip of `tiles` consecutive 256-pixel tiles of one
// row (blockIdx.y); a block is four wavefronts = four neighbouring strips that share nothing but the instruction cache:
// no staging, no barrier.  The strip's guard words arrive with one vector load (lane i = word i); per tile one scalar
// test of a ballot picks the variant:
//
//  * WIDE, four pixels per lane (device_math.h, MR_VEC4: every value four f64, every boolean four lane masks).  The
//    variant of a tile none of whose guard bits is set (every guarded region is the literal 0: for chess the background,
//    one multiply), and the whole section of a small program without guards (config 2: six ops).  The scalar unit's
//    share of a tile and the store's address arithmetic are paid once per 256 pixels, and a lane's four RGB8 pixels are
//    12 contiguous bytes: one global_store_dwordx3, no cross-lane packing.  This is the path that is bound by the store
//    (3 B per pixel) and little else.
//  * NARROW, one pixel per lane, four passes of 64 pixels (a loop: the section's code exists once).  The variant of a
//    tile where shapes may show.  Regions are entered per 64 pixels, where a wave-level SKIP op still finds all lanes
//    agreeing; values are single f64.  The passes leave their packed pixels in LDS (same-wave traffic: no barrier) and
//    the tile is stored like a wide one.
//
// When f64 planes are wanted too, element e of a wide lane l is pixel x0 + 64 e + l and every 64-pixel run is stored on
// its own (24 B per lane, the coalesced pattern of the f64 planes).  A Sin whose argument is huge (|x| >= 105414350), inf
// or NaN does not call the slow reduction here (a call site per Sin op would force every live value through scratch): the
// tile is flagged instead and re-evaluated by the tape interpreter kernel afterwards, so the final raster is identical.
// Layouts that were measured and lost (a wavefront per 64 pixels with guard words staged in LDS, a busy tile on the
// block's four wavefronts side by side, persistent wavefronts, two pixels per lane, guard words by scalar loads, a
// sky loop of its own ...) are history: DESIGN.md section 7.1, profiles/r2_ablations.jsonl.
std::string jit_source(const maray_program &P, int min_waves)
{
    validate_program(P);
    // 6 waves per SIMD, i.e. up to 102 SGPRs (at 8 the compiler gets 76 and spills ~400 of them to VGPR lanes, in the skeleton
    // of bit tests and branches every pass walks; chess needs 38 VGPRs either way and runs 7 waves per SIMD)
    const int min_waves_arg = min_waves;
    Emitter E(P);
    // Wave-level SKIP ops over fewer than 12 instructions' worth of ops are ignored: a busy tile is bound by the scalar unit
    // (branches, bit tests, mask algebra: 0.59 SALU instructions per cycle and CU against 35 % VALU issue), and a short
    // region's test and branch cost that unit more than its ops cost the vector one (chess board, us per 16.7 Mpx, with
    // guards per 256 x 8 pixels: none ignored 111, 24: 104; with guards per 64 x 32: 8 / 12 / 16 ... 32 / 64 / 200:
    // 83.3 / 82.4 / 84.7 / 85.4 / 128).
    E.min_region = 24;          // (round 4: from 12; pixel kernel / step / board 27.34 / 34.02 / 60.9 against 27.6 / 34.2-34.4 / 61.2 us, the same sign in round 3's table)
    if (const char *e_ = getenv("MARAY_JIT_MIN_REGION")) E.min_region = (uint32_t)atoi(e_);
    E.ybool = jit_bool_yvals(P);
    if (const char *e_ = getenv("MARAY_JIT_FUSE_CMP")) E.fuse_cmp = e_[0] != '0';
    E.ktab = true;
    for (uint32_t i = 0; i < P.n_pix_ops && E.sin_k < 0; i++)
        if (MARAY_INS_OP(P.pix_ops[i]) == MARAY_OP_STEPSIN) {
            // the first cache line of the table: what every leaf with a texture reads
            static const double sin_k[8] = {0x1.45f306dc9c883p-1, 0x1.8p52, 0x1.921fb58000000p+0, -0x1.dde973c000000p-27, -0x1.cb3b398000000p-55, -0x1.d747f23e32ed7p-83, 0x1p-70, 0.0};
            E.sin_k = 0;
            E.ktab_vals.assign(sin_k, sin_k + 8);
        }
    std::string &s = E.out;
    const uint32_t n_ynum = numeric_yvals(P);
    const uint32_t n_gwords = jit_guard_words(P);
    E.ignore_row_guards = n_gwords == 0;
    const GuardPlan plan = jit_guard_plan(P);
    const GuardGeom geom = jit_guard_geom(P);
    const uint32_t sub = 256u / geom.gw;                   // guard rectangles per 256-pixel tile (> 1: their words are taken per pass)
    const std::string tw = std::to_string(sub * n_gwords);  // guard words per tile
    if (n_gwords) { E.guard_first = n_ynum; E.guard_words = n_gwords; E.plan = &plan; E.gw_inline_max = GW_INLINE_MAX; }
    RedPlan reductions;
    if (!E.ignore_row_guards) {         // dry run: which ops yield lane masks
        Emitter D(P);
        D.ignore_row_guards = true;
        D.min_region = E.min_region;
        D.ybool = E.ybool;
        D.section(P.pix_ops, P.n_pix_ops, P.n_pix_slots, true, "v");
        E.bool_hint = D.is_bool_op;
        // OR trees of guarded shapes: evaluated from their set guard bits (RedPlan).  MARAY_JIT_REDUCE=0: walked as written (ablation)
        const char *e_ = getenv("MARAY_JIT_REDUCE");
        if (!(e_ && e_[0] == '0')) reductions = plan_reductions(P.pix_ops, P.n_pix_ops, P.n_pix_slots, D.is_bool_op, n_ynum, plan, E.ybool);
    }
    // Occupancy asked of the compiler.  Walking a tree of bit tests needs the SGPRs of 6 waves per SIMD (up to 102; at 8 the
    // compiler gets 80 and spilled ~400 of them to VGPR lanes, in the skeleton every pass walked); with the tree evaluated as
    // a reduction chess fits 78 and runs 8 (frame 29.9 -> 29.5 us, sky 12.3 -> 11.5, board 67.1 -> 65.1)
    if (min_waves_arg == 0) min_waves = reductions.empty() ? 6 : 8;
    // two rows per wavefront: twice the lane masks alive (chess at 8 waves: 240 SGPRs spilled to VGPR lanes); MARAY_JIT_ROWS2_WAVES: ablation
    if (min_waves_arg == 0 && jit_rows2(P)) { min_waves = 6; if (const char *e_ = getenv("MARAY_JIT_ROWS2_WAVES")) if (atoi(e_) > 0) min_waves = atoi(e_); }
    const bool defer = may_defer_tiles(P);
    const std::string nw = std::to_string(n_gwords);
    // a strip's guard words: one vector load per wavefront (lane i holds word i), then v_readlane per tile or pass -- one
    // memory latency per strip instead of one per tile
    const bool gw_vgpr = n_gwords && n_gwords <= GW_INLINE_MAX;
    const bool wide_general = jit_wide_general(P, n_gwords);
    // Two rows per wavefront (jit_rows2): the busy tiles' passes take the same 64 pixels of two neighbouring rows of one guard
    // rectangle -- the same shapes are entered for both, so a shape's dispatch, its constants and its x-only arithmetic are
    // paid once per 128 pixels (device_math.h, mr_p).  Needs guards bounded over rectangles of an even number of rows.
    const bool pair = jit_rows2(P);
    const std::string esub = "(e >> " + std::to_string(sub == 4 ? 0 : 1) + "u)";       // rectangle of pass e inside its tile
    s += "// generated by libmaray_hip (jit_source.cpp) from a v" + std::to_string(P.version) + " tape: PIXEL section, " +
         std::to_string(P.n_pix_ops) + " ops; general variant " + (wide_general ? "four pixels per lane" : "one pixel per lane, four passes per tile") + "\n"
         "#define MR_VEC4 1\n"
         "__shared__ unsigned mr_slow[4];           // per wavefront: some Sin of the tile at hand needs the slow path\n"
         "__shared__ unsigned mr_tp[4 * " + std::string(pair ? "512" : "256") + "];       // per wavefront: the packed pixels of a tile's four passes (of both rows)\n"
         "__device__ inline double mr_defer_sin(double) { ((volatile unsigned *)mr_slow)[threadIdx.x >> 6] = 1u; return 0.0; }\n"
         "#define MR_SIN_HUGE(x) mr_defer_sin(x)   // plain Sin ops: flag the tile from the (rare) branch\n"
         "#include \"device_math.h\"\n"
         "typedef const __attribute__((address_space(4))) double *mr_kptr;\n"
         "struct __attribute__((aligned(4))) mr_u3 { unsigned a, b, c; };\n"
         "struct __attribute__((aligned(16))) mr_u4 { unsigned a, b, c, d; };\n"
         "__device__ inline mr_mask mr_lane64(unsigned long long v, unsigned lane)      // lane `lane` (wave-uniform) of a per-lane 64-bit value -> SGPR pair\n"
         "{\n"
         "    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, (int)lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), (int)lane);\n"
         "    return ((mr_mask)hi << 32) | lo;\n"
         "}\n/*MR_KTAB*/\n";
    s += "extern \"C\" __global__ void __launch_bounds__(256, " + std::to_string(min_waves) +
         ") maray_jit_pixels(unsigned char *__restrict__ rgb8, double *__restrict__ rgb64,\n"
         "                                                                    const double *__restrict__ yvals, const MarayTex *__restrict__ tex,\n"
         "                                                                    unsigned *__restrict__ tile_list, unsigned tile_base,\n"
         "                                                                    const unsigned long long *__restrict__ gbits, unsigned n_tx,\n"
         "                                                                    unsigned w, unsigned y0, unsigned n_yvals, unsigned tiles,\n"
         "                                                                    unsigned blk_rows, unsigned blk_stride, unsigned row_base, unsigned yrows,\n"
         "                                                                    const unsigned *__restrict__ row_order, unsigned rows, unsigned rpw)\n{\n"
         "    const unsigned mr_wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), mr_lane = threadIdx.x & 63u;\n"
         "    // (workgroups go to the 8 XCDs round robin by their linear id: with 2, 4 or 8 blocks per row a column of the image\n"
         "    // always meets the same XCDs; rotating a row's strips by the row was measured and is not worth it, DESIGN.md 7.1)\n"
         "    const unsigned tile0 = (blockIdx.x * 4u + mr_wv) * tiles;               // this wavefront's strip of the row\n"
         "    if (tile0 >= n_tx) return;\n"
         "    // rows of this launch (dearest groups of rows first); row_base + r = row of the whole call.  rpw = 2 (two-row kernels, when the\n"
         "    // launch's guard groups have an even number of rows): this wavefront owns rows r and r + 1 of one group\n"
         "    const unsigned mr_rr = blockIdx.y * rpw;\n"
         "    const unsigned r = row_order ? row_order[mr_rr] : mr_rr;\n"
         "    (void)rows;\n" +
         (pair ? std::string(
         "    const bool mr_has1 = rpw == 2u && mr_rr + 1u < rows;                  // (else the second row of the pair repeats the first and is not stored)\n"
         "    const unsigned r1 = mr_has1 ? r + 1u : r;\n"
         "    const unsigned long long mr_ybase0 = (unsigned long long)(yvals + (size_t)r * n_yvals), mr_ybase1 = (unsigned long long)(yvals + (size_t)r1 * n_yvals);\n"
         "    const double Y0 = (double)(blk_stride == 0u ? y0 + row_base + r : y0 + ((row_base + r) / blk_rows) * blk_stride + (row_base + r) % blk_rows);\n"
         "    const double Y1 = (double)(blk_stride == 0u ? y0 + row_base + r1 : y0 + ((row_base + r1) / blk_rows) * blk_stride + (row_base + r1) % blk_rows);\n"
         "    (void)Y0; (void)Y1; (void)tex; (void)gbits; (void)yrows; (void)tile_list; (void)tile_base;\n")
               : std::string(
         "    const unsigned long long mr_ybase0 = (unsigned long long)(yvals + (size_t)r * n_yvals);\n"
         "    // -> image row (RowBlocks); one range of rows (blk_stride == 0) needs no division, and yrows is a power of two\n"
         "    const double Y = (double)(blk_stride == 0u ? y0 + row_base + r : y0 + ((row_base + r) / blk_rows) * blk_stride + (row_base + r) % blk_rows);\n"
         "    (void)Y; (void)tex; (void)gbits; (void)yrows; (void)tile_list; (void)tile_base;\n"));
    {   // the descriptors of the textures the section samples: scalar loads, once per wavefront
        std::vector<uint8_t> used;
        for (uint32_t i = 0; i < P.n_pix_ops; i++)
            if (MARAY_INS_OP(P.pix_ops[i]) == MARAY_OP_APP) { const uint32_t img = MARAY_INS_AUX(P.pix_ops[i]) / 5u; if (used.size() <= img) used.resize(img + 1, 0); used[img] = 1; }
        const char *e_ = getenv("MARAY_JIT_TEXEL_ONCE");           // "0": a call of mr_app per App op (= round 3; ablation)
        E.texel_once = !used.empty() && !(e_ && e_[0] == '0');
        if (E.texel_once)
            for (size_t img = 0; img < used.size(); img++)
                if (used[img]) s += "    const MarayTex mr_t" + std::to_string(img) + " = tex[" + std::to_string(img) + "];\n";
    }
    if (n_gwords)
        s += "    const unsigned long long mr_gbase0 = (unsigned long long)(gbits + ((size_t)((row_base + r) >> __builtin_ctz(yrows)) * n_tx + tile0) * " + tw + "u);\n";
    if (gw_vgpr)
        s += "    const unsigned mr_gn = (n_tx - tile0 < tiles ? n_tx - tile0 : tiles) * " + tw + "u;       // <= 64: the host bounds `tiles`\n"
             "    const unsigned long long mr_gv = mr_lane < mr_gn ? ((const unsigned long long *)mr_gbase0)[mr_lane] : 0ull;\n" +
             (sub > 1 ? "    const unsigned long long mr_gnz = mr_ballot(mr_gv != 0ull);            // which of the strip's words have a bit set\n" : "");
    s += "    const bool mr_wide = rgb64 == nullptr;                                   // wide variants: element e of lane l is pixel x0 + 4 l + e, else x0 + 64 e + l\n"
         "    const unsigned mr_xl = mr_wide ? 4u * mr_lane : mr_lane, mr_xs = mr_wide ? 1u : 64u;\n"
         "    const unsigned mr_src = (mr_lane * 4u) / 3u, mr_shift = ((mr_lane * 4u) % 3u) * 8u;   // RGB8 packing of one 64-pixel run\n"
         + std::string(pair ? "    const size_t row_px0 = (size_t)r * w, row_px1 = (size_t)r1 * w;\n" : "    const size_t row_px = (size_t)r * w;\n") +
         "    for (unsigned t = 0; t < tiles; t++) {\n"
         "    const unsigned x0 = (tile0 + t) * 256u;\n"
         "    if (x0 >= w) break;\n"
         "    // y values, constants, guard words: scalar loads where they are used, from addresses made opaque in every trip (fresh\n"
         "    // copies: an asm output carried around the loop counts as divergent once a lane-dependent branch sits in the loop)\n"
         "/*MR_KBASE*/";
    if (n_gwords && !gw_vgpr) {
        s += "    unsigned long long mr_gbase = mr_gbase0;\n"
             "    asm volatile(\"\" : \"+s\"(mr_gbase));\n";
        if (sub > 1)   // many words, narrow rectangles: lane i of mr_gt0 holds word i of the tile's rectangles (<= 64 together, jit_guard_geom)
            s += "    unsigned long long mr_gt0 = mr_lane < " + tw + "u ? ((const unsigned long long *)mr_gbase)[t * " + tw + "u + mr_lane] : 0ull;\n";
        else           // lane i of mr_gt<j> holds word 64 j + i of this tile (one vector load each); a test takes its word with v_readlane
            for (uint32_t j = 0; j < (n_gwords + 63) / 64; j++)
                s += "    unsigned long long mr_gt" + std::to_string(j) + " = " + std::to_string(64 * j) + "u + mr_lane < " + nw + "u ? ((const unsigned long long *)mr_gbase)[t * " + nw + "u + " +
                     std::to_string(64 * j) + "u + mr_lane] : 0ull;\n";
    } else if (gw_vgpr && sub == 1)
        for (uint32_t j = 0; j < n_gwords; j++)
            s += "    mr_mask gq" + std::to_string(j) + " = mr_lane64(mr_gv, t * " + nw + "u + " + std::to_string(j) + "u);\n";
    if (defer) s += "    ((volatile unsigned *)mr_slow)[mr_wv] = 0u;\n    bool mr_slow_tile = false;\n";
    // what opens a pass of either width: the tables made opaque (LICM would hoist every constant and y value out of the
    // loops and spill them), the pixel coordinates, the outputs
    const std::string opaque = std::string(
        pair ? "    unsigned long long mr_ybase = mr_q ? mr_ybase1 : mr_ybase0;      // (wide tiles of a two-row kernel: row after row)\n"
             : "    unsigned long long mr_ybase = mr_ybase0;\n") +
        "    asm volatile(\"\" : \"+s\"(mr_ybase));\n"
        "    mr_kptr yv = (mr_kptr)mr_ybase;\n"
        "    const __attribute__((address_space(4))) unsigned *yw = (const __attribute__((address_space(4))) unsigned *)yv;\n"
        "    (void)yv; (void)yw;\n/*MR_KC*/";
    // the tables of both rows of a pair
    const std::string opaque2 =
        "    unsigned long long mr_ybase = mr_ybase0, mr_ybaseb = mr_ybase1;\n"
        "    asm volatile(\"\" : \"+s\"(mr_ybase), \"+s\"(mr_ybaseb));\n"
        "    mr_kptr yv = (mr_kptr)mr_ybase, yv1 = (mr_kptr)mr_ybaseb;\n"
        "    const __attribute__((address_space(4))) unsigned *yw = (const __attribute__((address_space(4))) unsigned *)yv, *yw1 = (const __attribute__((address_space(4))) unsigned *)yv1;\n"
        "    (void)yv; (void)yw; (void)yv1; (void)yw1;\n/*MR_KC*/";
    // the guard words of the rectangle at hand, opaque anew in every pass: left visible, all their bit tests are loop
    // invariants too (168 booleans for chess, hoisted and spilled to VGPR lanes)
    std::string gq_pass;
    if (n_gwords && !gw_vgpr && sub > 1) {
        gq_pass = "    asm volatile(\"\" : \"+v\"(mr_gt0));\n"
                  "    const unsigned mr_gsub = " + esub + " * " + nw + "u;           // first word of this pass's rectangle\n"
                  "    const unsigned long long mr_gnzp = mr_ballot(mr_gt0 != 0ull) >> mr_gsub;      // which of its words have a bit set\n"
                  "    (void)mr_gnzp;\n";
        E.gw_lane_base = "mr_gsub";
    } else if (gw_vgpr && sub > 1)
        for (uint32_t j = 0; j < n_gwords; j++) {
            const std::string k = std::to_string(j);
            gq_pass += "    mr_mask gq" + k + " = mr_lane64(mr_gv, (t * " + std::to_string(sub) + "u + " + esub + ") * " + nw + "u + " + k + "u);\n"
                       "    asm volatile(\"\" : \"+s\"(gq" + k + "));\n";
        }
    else if (gw_vgpr)
        for (uint32_t j = 0; j < n_gwords; j++) gq_pass += "    asm volatile(\"\" : \"+s\"(gq" + std::to_string(j) + "));\n";
    else if (n_gwords)
        for (uint32_t j = 0; j < (n_gwords + 63) / 64; j++) {
            const std::string k = std::to_string(j);
            gq_pass += "    asm volatile(\"\" : \"+v\"(mr_gt" + k + "));\n"
                       "    const unsigned long long mr_gnz" + k + " = mr_ballot(mr_gt" + k + " != 0ull);      // which of the tile's words have a bit set\n"
                       "    (void)mr_gnz" + k + ";\n";
        }
    const std::string wide_open =
        std::string(pair ? "    for (unsigned mr_q = 0; mr_q <= (unsigned)mr_has1; mr_q++) {\n"
                           "    const double Y = mr_q ? Y1 : Y0;\n"
                           "    const size_t row_px = mr_q ? row_px1 : row_px0;\n"
                           "    (void)Y;\n"
                         : "    {\n") + opaque + (sub > 1 ? std::string() : gq_pass) +
        "    const unsigned xa = x0 + mr_xl;                                        // this lane's first pixel\n"
        "    const mr_d X((double)xa, (double)(xa + mr_xs), (double)(xa + 2u * mr_xs), (double)(xa + 3u * mr_xs));\n"
        "    mr_d o0 = 0.0, o1 = 0.0, o2 = 0.0;\n"
        "    float mr_defer = 0.0f;                     // fused Step(Sin) ops count their undecided cases in here\n"
        "    (void)X; (void)mr_defer;\n";
    const std::string defer_pass = defer ?
        "    mr_slow_tile |= mr_ballot(mr_defer != 0.0f) != 0ull || ((volatile unsigned *)mr_slow)[mr_wv] != 0u;      // wave-uniform\n" : "";
    const std::string wide_close = defer_pass +
        "    const unsigned p0 = mr_cast_u8(o0.a) | (mr_cast_u8(o1.a) << 8) | (mr_cast_u8(o2.a) << 16);\n"
        "    const unsigned p1 = mr_cast_u8(o0.b) | (mr_cast_u8(o1.b) << 8) | (mr_cast_u8(o2.b) << 16);\n"
        "    const unsigned p2 = mr_cast_u8(o0.c) | (mr_cast_u8(o1.c) << 8) | (mr_cast_u8(o2.c) << 16);\n"
        "    const unsigned p3 = mr_cast_u8(o0.d) | (mr_cast_u8(o1.d) << 8) | (mr_cast_u8(o2.d) << 16);\n"
        "    if (mr_wide) {\n"
        "        if (rgb8) {\n"
        "            unsigned char *q = rgb8 + (row_px + xa) * 3;                      // this lane's 12 bytes\n"
        "            if (x0 + 256u <= w && ((size_t)(rgb8 + (row_px + x0) * 3) & 3u) == 0u) {     // wave-uniform: a whole tile whose bytes start on a dword\n"
        "                mr_u3 d;\n"
        "                d.a = p0 | (p1 << 24); d.b = (p1 >> 8) | (p2 << 16); d.c = (p2 >> 16) | (p3 << 8);\n"
        "                *(mr_u3 *)q = d;\n"
        "            } else {\n"
        "                const unsigned pk[4] = {p0, p1, p2, p3};\n"
        "                for (unsigned e = 0; e < 4u; e++)\n"
        "                    if (xa + e < w) { q[3 * e] = (unsigned char)pk[e]; q[3 * e + 1] = (unsigned char)(pk[e] >> 8); q[3 * e + 2] = (unsigned char)(pk[e] >> 16); }\n"
        "            }\n"
        "        }\n"
        "    } else {\n"
        "        const unsigned pk[4] = {p0, p1, p2, p3};\n"
        "        const double c0[4] = {o0.a, o0.b, o0.c, o0.d}, c1[4] = {o1.a, o1.b, o1.c, o1.d}, c2[4] = {o2.a, o2.b, o2.c, o2.d};\n"
        "        _Pragma(\"unroll\") for (unsigned e = 0; e < 4u; e++)\n"
        "            mr_store_run(rgb8, rgb64, row_px, x0 + 64u * e, w, mr_lane, mr_src, mr_shift, pk[e], c0[e], c1[e], c2[e]);\n"
        "    }\n"
        "    }\n";
    // one 64-pixel run: f64 planes (24 B per lane) and / or RGB8 (48 lanes assemble a dword each from two neighbours'
    // packed colours; ragged ends and unaligned rows store bytes)
    const std::string store_run =
        "__device__ inline void mr_store_run(unsigned char *__restrict__ rgb8, double *__restrict__ rgb64, size_t row_px, unsigned xw, unsigned w,\n"
        "                                    unsigned lane, unsigned src, unsigned shift, unsigned pk, double c0, double c1, double c2)\n{\n"
        "    const unsigned x = xw + lane;\n"
        "    if (rgb64 && x < w) { const size_t p = (row_px + x) * 3; rgb64[p] = c0; rgb64[p + 1] = c1; rgb64[p + 2] = c2; }\n"
        "    if (rgb8) {\n"
        "        unsigned char *wave_out = rgb8 + (row_px + xw) * 3;\n"
        "        if (xw + 64u <= w && ((size_t)wave_out & 3u) == 0u) {                // wave-uniform\n"
        "            const unsigned pa = (unsigned)__builtin_amdgcn_ds_bpermute((int)(src * 4u), (int)pk);\n"
        "            const unsigned pb = (unsigned)__builtin_amdgcn_ds_bpermute((int)(src * 4u + 4u), (int)pk);\n"
        "            const unsigned dw = (unsigned)((((unsigned long long)pb << 24) | pa) >> shift);\n"
        "            if (lane < 48u) ((unsigned *)wave_out)[lane] = dw;\n"
        "        } else if (x < w) {\n"
        "            unsigned char *q = rgb8 + (row_px + x) * 3;\n"
        "            q[0] = (unsigned char)pk; q[1] = (unsigned char)(pk >> 8); q[2] = (unsigned char)(pk >> 16);\n"
        "        }\n"
        "    }\n"
        "}\n";
    std::string tile_end;         // closes a tile: the work list entry of a tile some Sin of which needs the slow path
    if (defer)
        tile_end = "    if (mr_slow_tile && mr_lane == 0u) tile_list[1u + atomicAdd(&tile_list[0], 1u)] = tile_base + r * n_tx + tile0 + t;\n" +
                   std::string(pair ? "    if (mr_slow_tile && mr_has1 && mr_lane == 0u) tile_list[1u + atomicAdd(&tile_list[0], 1u)] = tile_base + r1 * n_tx + tile0 + t;\n" : "");

    if (n_gwords) {
        // the variant of a tile with no guard bit set, four pixels per lane
        if (!gw_vgpr && sub > 1)
            s += "    if (mr_ballot(mr_gt0 != 0ull) == 0ull) {\n";
        else if (sub > 1)
            s += "    if (((mr_gnz >> (t * " + tw + "u)) & " + std::to_string((1ull << (sub * n_gwords)) - 1ull) + "ull) == 0ull) {\n";
        else if (gw_vgpr) {
            std::string any = "gq0";
            for (uint32_t j = 1; j < n_gwords; j++) any += " | gq" + std::to_string(j);
            s += "    if ((" + any + ") == 0ull) {\n";
        } else {
            std::string any = "mr_gt0";
            for (uint32_t j = 1; j < (n_gwords + 63) / 64; j++) any += " | mr_gt" + std::to_string(j);
            s += "    if (mr_ballot((" + any + ") != 0ull) == 0ull) {\n";
        }
        E.td = "mr_d"; E.tm = "mr_m";
        E.assume_guards_zero = true;
        s += wide_open;
        E.section(P.pix_ops, P.n_pix_ops, P.n_pix_slots, true, "v");
        s += wide_close;
        E.assume_guards_zero = false;
        s += tile_end + "    continue;\n    }\n";
    }
    if (wide_general) {
        E.td = "mr_d"; E.tm = "mr_m";
        s += wide_open;
        E.section(P.pix_ops, P.n_pix_ops, P.n_pix_slots, true, "v");
        s += wide_close + tile_end;
    } else if (pair) {
        // one wavefront, four passes of 64 pixels x 2 rows: a value that depends on the row is a pair (mr_p), the rest is
        // computed once; the packed pixels of both rows go to LDS and each row's tile is stored as a dwordx3 per lane
        E.td = "double"; E.tm = "mr_mask";
        E.pair = true;
        E.rplan = &reductions;
        {   // typing run: which ops yield pairs (the regions' variables are declared ahead of their last op)
            Emitter T(E);
            T.wide_hint.clear();
            T.section(P.pix_ops, P.n_pix_ops, P.n_pix_slots, true, "v");
            E.wide_hint = T.is_wide_op;
        }
        s += "    const bool mr_al0 = x0 + 256u <= w && ((size_t)(rgb8 + (row_px0 + x0) * 3) & 3u) == 0u, mr_al1 = x0 + 256u <= w && ((size_t)(rgb8 + (row_px1 + x0) * 3) & 3u) == 0u;\n"
             "    const bool mr_fast = rgb8 && mr_al0 && mr_al1;      // wave-uniform\n"
             "    _Pragma(\"unroll 1\") for (unsigned e = 0; e < 4u; e++) {\n" + opaque2 + gq_pass +
             "    const unsigned xw = x0 + 64u * e, x = xw + mr_lane;\n"
             "    const double X = (double)x;\n"
             "    const mr_p Y(Y0, Y1);\n"
             "    mr_p o0 = 0.0, o1 = 0.0, o2 = 0.0;\n"
             "    float mr_defer = 0.0f;\n"
             "    (void)X; (void)Y; (void)mr_defer;\n";
        E.section(P.pix_ops, P.n_pix_ops, P.n_pix_slots, true, "v");
        E.rplan = nullptr;
        E.pair = false;
        s += defer_pass +
             "    const unsigned pk0 = mr_cast_u8(o0.a) | (mr_cast_u8(o1.a) << 8) | (mr_cast_u8(o2.a) << 16);\n"
             "    const unsigned pk1 = mr_cast_u8(o0.b) | (mr_cast_u8(o1.b) << 8) | (mr_cast_u8(o2.b) << 16);\n"
             "    if (mr_fast) {\n"
             "        mr_tp[mr_wv * 512u + 64u * e + mr_lane] = pk0;\n"
             "        mr_tp[mr_wv * 512u + 256u + 64u * e + mr_lane] = pk1;\n"
             "        mr_store_run(nullptr, rgb64, row_px0, xw, w, mr_lane, mr_src, mr_shift, pk0, o0.a, o1.a, o2.a);\n"
             "        if (mr_has1) mr_store_run(nullptr, rgb64, row_px1, xw, w, mr_lane, mr_src, mr_shift, pk1, o0.b, o1.b, o2.b);\n"
             "    } else {\n"
             "        mr_store_run(rgb8, rgb64, row_px0, xw, w, mr_lane, mr_src, mr_shift, pk0, o0.a, o1.a, o2.a);\n"
             "        if (mr_has1) mr_store_run(rgb8, rgb64, row_px1, xw, w, mr_lane, mr_src, mr_shift, pk1, o0.b, o1.b, o2.b);\n"
             "    }\n"
             "    }\n"
             "    if (mr_fast) {\n"
             "        __builtin_amdgcn_wave_barrier();                                     // same wavefront wrote them: LDS keeps its order\n"
             "        for (unsigned mr_q = 0; mr_q <= (unsigned)mr_has1; mr_q++) {\n"
             "            const mr_u4 p = *(const mr_u4 *)&mr_tp[mr_wv * 512u + 256u * mr_q + 4u * mr_lane];\n"
             "            mr_u3 d;\n"
             "            d.a = p.a | (p.b << 24); d.b = (p.b >> 8) | (p.c << 16); d.c = (p.c >> 16) | (p.d << 8);\n"
             "            *(mr_u3 *)(rgb8 + ((mr_q ? row_px1 : row_px0) + x0 + 4u * mr_lane) * 3) = d;\n"
             "        }\n"
             "        __builtin_amdgcn_wave_barrier();\n"
             "    }\n" + tile_end;
    } else {
        // one wavefront, four passes of 64 pixels (a loop, not unrolled); the passes leave their packed pixels in LDS
        // (same-wave traffic: no barrier) and a whole aligned tile is stored as a dwordx3 per lane
        E.td = "double"; E.tm = "mr_mask";
        s += "    const bool mr_fast = rgb8 && x0 + 256u <= w && ((size_t)(rgb8 + (row_px + x0) * 3) & 3u) == 0u;      // wave-uniform\n"
             "    _Pragma(\"unroll 1\") for (unsigned e = 0; e < 4u; e++) {\n" + opaque + gq_pass +
             "    const unsigned xw = x0 + 64u * e, x = xw + mr_lane;\n"
             "    const double X = (double)x;\n"
             "    double o0 = 0.0, o1 = 0.0, o2 = 0.0;\n"
             "    float mr_defer = 0.0f;\n"
             "    (void)X; (void)mr_defer;\n";
        E.rplan = &reductions;
        E.section(P.pix_ops, P.n_pix_ops, P.n_pix_slots, true, "v");
        E.rplan = nullptr;
        s += defer_pass +
             "    const unsigned pk = mr_cast_u8(o0) | (mr_cast_u8(o1) << 8) | (mr_cast_u8(o2) << 16);\n"
             "    if (mr_fast) {\n"
             "        mr_tp[mr_wv * 256u + 64u * e + mr_lane] = pk;\n"
             "        mr_store_run(nullptr, rgb64, row_px, xw, w, mr_lane, mr_src, mr_shift, pk, o0, o1, o2);\n"
             "    } else mr_store_run(rgb8, rgb64, row_px, xw, w, mr_lane, mr_src, mr_shift, pk, o0, o1, o2);\n"
             "    }\n"
             "    if (mr_fast) {\n"
             "        __builtin_amdgcn_wave_barrier();                                     // same wavefront wrote them: LDS keeps its order\n"
             "        const mr_u4 p = *(const mr_u4 *)&mr_tp[mr_wv * 256u + 4u * mr_lane];\n"
             "        mr_u3 d;\n"
             "        d.a = p.a | (p.b << 24); d.b = (p.b >> 8) | (p.c << 16); d.c = (p.c >> 16) | (p.d << 8);\n"
             "        *(mr_u3 *)(rgb8 + (row_px + x0 + 4u * mr_lane) * 3) = d;\n"
             "        __builtin_amdgcn_wave_barrier();\n"
             "    }\n" + tile_end;
    }
    s += "    }\n}\n";
    {
        std::string tab = store_run;
        if (!E.ktab_vals.empty()) {
            tab += "__constant__ __attribute__((aligned(64))) double mr_kc_tab[" + std::to_string(E.ktab_vals.size()) + "] = {";
            for (size_t j = 0; j < E.ktab_vals.size(); j++) { tab += (j % 6 ? " " : "\n    "); tab += lit(E.ktab_vals[j]); tab += ","; }
            tab += "\n};\n";
        }
        s.replace(s.find("/*MR_KTAB*/"), 11, tab);
        // The table's address, made opaque once per tile and once per pass (see `opaque`)
        for (size_t at; (at = s.find("/*MR_KBASE*/")) != std::string::npos;)
            s.replace(at, 12, E.ktab_vals.empty() ? "" : "    unsigned long long mr_kbase = (unsigned long long)mr_kc_tab;\n");
        const std::string kc = E.ktab_vals.empty() ? "    asm volatile(\"\" ::: \"memory\");\n" :
                               "    asm volatile(\"\" : \"+s\"(mr_kbase) :: \"memory\");\n"
                               "    const mr_kptr mr_kc = (mr_kptr)mr_kbase;\n";
        for (size_t at; (at = s.find("/*MR_KC*/")) != std::string::npos;) s.replace(at, 9, kc);
    }
    return s;
}


}   // namespace maray
