#!/usr/bin/env python3
"""bench.py — Mpixels/s of the MI355X evaluator on chess.maray @ 4096x4096.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it
is launched under torch.distributed.run with one rank per GPU.  A "step" is one
pass of the hot path over the rank's share (4096 x 4096 pixels) of one image of
the chess scene, outputs resident in HBM.  Pixels are independent, so ranks own
disjoint rows (interleaved 64-row blocks, for balance) and no data-path
collective is issued (weak scaling: pixels per GPU are fixed); the only
collectives are the timing barrier and the max-over-ranks of the elapsed time.

Rank 0 prints ONE JSON line with the metric, the roofline of the dominant kernel
(live HIP-event timing on the launch stream) and, at N = 1, the CPU baseline
(the oracle = restatement of the reference's Rayon interpreter, timed on a
bounded sample of the same workload on this host's cores).

Roofline: SURVEY.md §8(d) gives two per-pixel figures for this path, 3 bytes of
mandatory HBM traffic (the RGB8 pixel written) and 10,241 f64 ops (the census of
the scene's DAG).  The kernel evaluates the scene exactly but, by proving whole
shapes absent from a 256-pixel tile, executes a small fraction of that census
(profiles/), so the roof that binds it is the output store: `roofline.bound` is
"hbm" with 3 B/pixel; the census figure is kept beside it as `valu_f64_census`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_OPS_PER_PIXEL = 10241          # 10,239 unique non-constant ops + the two rescale muls
PEAK_F64_TOPS = 39.3               # MI355X f64 VALU, non-FMA instr/s: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz
PEAK_HBM_GBS = 8000.0
W = 4096
H_TILE = 4096
BLOCK_ROWS = 64                    # rows are dealt to the ranks in blocks of this many (a multiple of the 8-row guard groups)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--backend', default=os.environ.get('MARAY_BENCH_BACKEND', 'auto'))
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='CPU baseline sample budget (0 = skip)')
    ap.add_argument('--cpu-jit', action='store_true', help='also time the JIT stand-in (scene compiled to native code by cc)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import maray_amd as M

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a MI355X: no HIP device is visible (there is no CPU fallback)')
    torch.cuda.set_device(local)
    n_gpus = world
    if world == 1 and os.environ.get('MARAY_BENCH_FAKE_WORLD'):      # rehearse rank r of N on one GPU, no process group
        n_gpus = int(os.environ['MARAY_BENCH_FAKE_WORLD'])
        rank = int(os.environ.get('MARAY_BENCH_FAKE_RANK', '0'))

    data = open(os.path.join(ROOT, 'tests', 'golden', 'chess.maray'), 'rb').read()
    scene = M.Scene(data)
    # N = 1: config 3, chess.maray regenerated at 4096 x 4096 (exact power-of-two rescale).  N > 1: the same scene
    # at N x 4096^2 pixels (8192x4096, 8192^2, 16384x8192 = config 4's width), rows dealt to the ranks in interleaved
    # 64-row blocks so that every rank sees sky and board alike: equal pixels per rank, no data-path collective.
    from maray_amd.sharding import interleaved_blocks, interleaved_layout, max_over_ranks, scene_scale
    sx, sy = scene_scale(n_gpus)
    scene.rescale(sx, sy)
    w_img, h_total = scene.size
    tape = scene.lower(row_guards=os.environ.get('MARAY_BENCH_ROW_GUARDS', '1') != '0')
    blocks = interleaved_blocks(rank, n_gpus, h_total, BLOCK_ROWS)
    layout = interleaved_layout(rank, n_gpus, h_total, BLOCK_ROWS)     # the same rows as one launch (None: ragged, block by block)
    rows_mine = sum(b - a for a, b in blocks)

    backends = {'tape': M.BACKEND_TAPE, 'tape-smem': M.BACKEND_TAPE_SMEM, 'jit': M.BACKEND_JIT}
    order = ['jit', 'tape-smem', 'tape'] if args.backend == 'auto' else [args.backend]
    ctx = None
    for name in order:
        try:
            ctx = M.Context(tape, device=local, backend=backends[name])
            backend_name = name
            break
        except M.MarayError as e:
            if args.backend != 'auto':
                raise
            print('bench.py: backend %s unavailable (%s); trying the next one' % (name, e), file=sys.stderr, flush=True)
            last = e
    if ctx is None:
        raise last

    out8 = torch.empty((rows_mine, w_img, 3), dtype=torch.uint8, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        if layout is not None:
            ctx.render_blocks_device(w_img, h_total, *layout, d_rgb8=out8.data_ptr(), stream=stream)
            return
        off = 0
        for a, b in blocks:
            ctx.render_rows_device(w_img, h_total, a, b, d_rgb8=out8.data_ptr() + off * w_img * 3, stream=stream)
            off += b - a

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dist if world > 1 else None, dt, device='cuda')

    px_per_step = w_img * h_total
    value = px_per_step * args.steps / dt / 1e6

    # roofline of the dominant (pixel) kernel: HIP events on the launch stream, this rank's first block
    a0, b0 = blocks[0]
    k_ms = ctx.time_rows(w_img, h_total, a0, b0, d_rgb8=out8.data_ptr(), reps=max(3, min(args.steps, 10)))
    px_launch = w_img * (b0 - a0)
    census_tops = ALG_OPS_PER_PIXEL * px_launch / (k_ms * 1e-3) / 1e12
    hbm_gbs = (px_launch * 3) / (k_ms * 1e-3) / 1e9

    # parity spot check of the timed output against the committed golden: pixel (sx*i, sy*j) of the rescaled scene
    # equals pixel (i, j) of the stored one (config 1) bit for bit
    parity = None
    if rank == 0:
        import hashlib
        g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'chess_1024.json')))
        if n_gpus == 1:
            sub = out8[::sy, ::sx].contiguous().cpu().numpy()
            parity = hashlib.sha256(sub.tobytes()).hexdigest() == g['rgb8_sha256']
        else:   # rank 0 holds rows [0,256) first: stored rows 0 and 100*... -> check stored row 0
            sub = out8[0, ::sx].contiguous().cpu().numpy()
            parity = hashlib.sha256(sub.tobytes()).hexdigest() == g['row_sha256']['0']

    # HBM traffic per launch from the committed PMC profile of this very command (FETCH_SIZE x2 per the gfx950
    # correction of MI355X_MICROARCH.md + WRITE_SIZE); bench.py cannot collect PMC counters itself.
    traffic = None
    executed = None
    prof = os.path.join(ROOT, 'profiles', 'r1_%s_chess4096_pmc.json' % backend_name)
    if os.path.exists(prof):
        d = json.load(open(prof))['derived']
        traffic = (d['hbm_fetch_bytes_x2_gfx950_correction'] + d['hbm_write_bytes']) * px_launch / (4096 * 4096)
        # what the kernel actually issues (committed PMC profile): wave-level short circuits skip most of the
        # boolean-gated work, so the executed VALU stream is far shorter than the algorithmic op count
        executed = {'source': os.path.relpath(prof, ROOT), 'valu_insts_per_wave': d['valu_insts_per_wave'],
                    'salu_insts_per_wave': d['salu_insts_per_wave'],
                    'cycles_per_valu_inst_per_simd': d['cycles_per_valu_inst_per_simd'],
                    # a wave64 f64 VALU op occupies its SIMD for 4 cycles, a 32-bit one for 2: issue-port occupancy bounds
                    'valu_port_busy_frac_bounds': [2.0 / d['cycles_per_valu_inst_per_simd'], 4.0 / d['cycles_per_valu_inst_per_simd']]}

    cpu = None
    if rank == 0 and n_gpus == 1 and world == 1 and args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from oracle_ffi import Scene as OScene
        o = OScene(scene.encode())
        W = w_img
        threads = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        batch = max(16, threads // 2)              # rows per call: keeps every thread busy
        rows, ct = 0, 0.0
        while ct < args.cpu_seconds and 2048 + rows + batch <= H_TILE:
            t = time.perf_counter()
            o.render_rows(W, H_TILE, 2048 + rows, 2048 + rows + batch, threads=threads, want_f64=False)
            ct += time.perf_counter() - t
            rows += batch
        cpu = {'value': W * rows / ct / 1e6, 'unit': 'Mpixels/s', 'cores': threads, 'kind': 'port',
               'sample': '%d rows x %d px of the same 4096x4096 chess scene (rows 2048..%d), oracle = restated '
                         'ParallelInterpreted (src/render.rs:35-99), %.1f s' % (rows, W, 2048 + rows, ct)}

    cpu_jit = None
    if cpu is not None and args.cpu_jit:
        from oracle_ffi import JitBaseline
        t = time.perf_counter()
        jb = JitBaseline(o)
        build_s = time.perf_counter() - t
        rows, ct = 0, 0.0
        while ct < args.cpu_seconds and 2048 + rows + batch <= H_TILE:
            t = time.perf_counter()
            jb.render_rows(W, 2048 + rows, 2048 + rows + batch, threads=threads)
            ct += time.perf_counter() - t
            rows += batch
        cpu_jit = {'value': W * rows / ct / 1e6, 'unit': 'Mpixels/s', 'cores': threads, 'kind': 'port',
                   'sample': '%d rows x %d px, scene emitted as straight-line C like src/wasm.rs gen_expr (un-shared, one '
                             'function per channel, out-of-line recip/step/sin), cc -O2, build %.1f s, run %.1f s'
                             % (rows, W, build_s, ct)}

    if rank == 0 or (world == 1 and n_gpus > 1):
        line = {
            'metric': 'Mpixels/s on chess.maray @4096x4096',
            'value': value, 'unit': 'Mpixels/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'data/chess.maray rescaled to %d x %d (SURVEY.md §8(d) config 3 at N=1; N x 4096^2 pixels '
                                   'of the same scene at N>1), rows dealt to the ranks in interleaved 64-row blocks'
                                   % (w_img, h_total),
                       'backend': backend_name, 'kernel': ctx.kernel_name, 'pixels_per_step': px_per_step,
                       'tape_ops_per_pixel': tape.info['n_pix_ops'], 'parallelism': 'row tiles, no collective',
                       'bit_exact_vs_golden': parity},
            'roofline': {'bound': 'hbm', 'achieved': hbm_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                         'frac': hbm_gbs / PEAK_HBM_GBS, 'traffic': traffic,
                         'kernel_ms': k_ms, 'alg_bytes_per_pixel': 3, 'pixels_per_launch': px_launch,
                         'note': 'achieved = 3 B/pixel (SURVEY §8(d): the RGB8 pixel written is the only mandatory HBM '
                                 'traffic) x pixels of one launch / its HIP-event time; traffic = FETCH_SIZE x2 + WRITE_SIZE '
                                 'of the committed PMC profile of this command, scaled to this launch',
                         'valu_f64_census': {'achieved': census_tops, 'peak': PEAK_F64_TOPS, 'unit': 'TFLOP/s',
                                             'frac': census_tops / PEAK_F64_TOPS, 'alg_ops_per_pixel': ALG_OPS_PER_PIXEL,
                                             'note': 'SURVEY §8(d) census of the scene DAG (10,241 f64 ops per pixel) / kernel '
                                                     'time; far above 1 because regions gated by a boolean that a y-only bound '
                                                     'proves 0 over a 256-pixel tile are skipped, and half of what remains is '
                                                     'boolean algebra on lane masks (scalar unit)'},
                         'executed': executed},
            'cpu_baseline': cpu,
            'cpu_baseline_jit': cpu_jit,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
