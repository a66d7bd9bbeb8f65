#!/usr/bin/env python3
"""bench.py — Mpixels/s of the MI355X evaluator on chess.maray @ 4096x4096.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 it
runs one rank per GPU: under torch.distributed.run when the driver starts it that
way (WORLD_SIZE set), else it starts its own N ranks as child processes and
relays rank 0's line (the parent never touches HIP).  With fewer visible GPUs
than ranks the ranks share them (a rehearsal, said so in the line).  A "step" is one
pass of the hot path over the rank's share (4096 x 4096 pixels) of one image of
the chess scene, outputs resident in HBM.  Pixels are independent, so ranks own
disjoint rows (interleaved 64-row blocks, for balance) and no data-path
collective is issued (weak scaling: pixels per GPU are fixed); the only
collectives are the timing barrier and the max-over-ranks of the elapsed time.
`--scaling strong` renders config 4 instead (chess @ 16384^2, total work fixed).

Rank 0 prints ONE JSON line:
  value / ms_per_step   SURVEY 8(d) metric (i): kernel-side throughput, outputs in HBM
  roofline              the dominant kernel against the HBM roof (3 B/pixel written), HIP events on the launch stream
  end_to_end            SURVEY 8(d) metric (ii): the same frame into a pinned host raster through the host entry
                        point (device -> host DMA included), against the PCIe roof; and into pageable memory
  cold_first_render_ms  context creation + first frame: cold code cache (hiprtc builds both kernels), warm cache
                        (a fresh process loads the code objects), and the interpreter (no build at all)
  cpu_baseline          B1: the oracle = restated Rayon interpreter, on this host's cores (N = 1 only)
  cpu_baseline_jit      B2: stand-in for the reference's wasmer JIT (scene emitted as straight-line C, cc -O2)

Roofline: SURVEY.md 8(d) gives two per-pixel figures for this path, 3 bytes of
mandatory HBM traffic (the RGB8 pixel written) and 10,241 f64 ops (the census of
the scene's DAG).  The kernel evaluates the scene exactly but, by proving whole
shapes absent from a rectangle of 64 x 32 pixels, executes a small fraction of that census
(profiles/), so the roof that binds it is the output store: `roofline.bound` is
"hbm" with 3 B/pixel; the census figure is kept beside it as `valu_f64_census`.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_OPS_PER_PIXEL = 10241          # 10,239 unique non-constant ops + the two rescale muls
PEAK_F64_TOPS = 39.3               # MI355X f64 VALU, non-FMA instr/s: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz
PEAK_HBM_GBS = 8000.0
PEAK_PCIE_GBS = 63.0               # PCIe Gen5 x16 (MI355X_MICROARCH.md)
BLOCK_ROWS = 64                    # rows are dealt to the ranks in blocks of this many (a multiple of the 32-row guard groups)

_FIRST_RENDER = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import maray_amd as M
s = M.Scene(open(os.path.join(%(root)r, 'tests', 'golden', 'chess.maray'), 'rb').read())
s.rescale(4, 4)
tape = s.lower()
M.device_count()                                   # HIP runtime initialisation is not part of the figure
pin = M.PinnedRaster(4096, 4096)
cached = tape.jit_code_cached                      # code objects of this program in MARAY_CACHE_DIR before the context exists
t0 = time.perf_counter()
ctx = M.Context(tape, backend=%(backend)d)
t1 = time.perf_counter()
ctx.render_rows_into(4096, 4096, 0, 4096, pin.array)
t2 = time.perf_counter()
ctx.render_rows_into(4096, 4096, 0, 4096, pin.array)
t3 = time.perf_counter()
print(json.dumps({'ctx_ms': (t1 - t0) * 1e3, 'frame_ms': (t2 - t1) * 1e3, 'second_frame_ms': (t3 - t2) * 1e3, 'cached': cached}))
'''


def first_render(backend, env):
    """Context creation + first 4096^2 frame into a pinned raster, in a fresh process (no warm state of any kind)."""
    out = subprocess.run([sys.executable, '-c', _FIRST_RENDER % dict(root=ROOT, backend=backend)], capture_output=True, text=True,
                         env=dict(os.environ, **env), timeout=900)
    if out.returncode != 0:
        return {'error': out.stderr[-300:]}
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--backend', default=os.environ.get('MARAY_BENCH_BACKEND', 'auto'))
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak (default): N x 4096^2 pixels; strong: config 4, chess @ 16384^2 shared by the N ranks')
    ap.add_argument('--cpu-seconds', type=float, default=10.0, help='budget of each CPU baseline sample (0 = skip both)')
    ap.add_argument('--no-cpu-jit', action='store_true', help='skip the JIT stand-in (B2)')
    ap.add_argument('--no-cold', action='store_true', help='skip the cold / warm first-render measurement')
    ap.add_argument('--no-e2e', action='store_true', help='skip the end-to-end (host raster) measurement')
    ap.add_argument('--long-steps', type=int, default=2000, help='steps of the long loop between two fences (0 = skip)')
    ap.add_argument('--rank-timeout', type=float, default=1500.0, help='seconds the launcher waits for its ranks')
    args = ap.parse_args()

    fake_world = os.environ.get('MARAY_BENCH_FAKE_WORLD')
    # (an environment that exports WORLD_SIZE=1 to everything it starts is not a launcher either; our own ranks carry a mark)
    if args.gpus > 1 and os.environ.get('WORLD_SIZE', '1') in ('', '1') and 'MARAY_BENCH_RANK_OF' not in os.environ and not fake_world:
        # `python bench.py --gpus N` with no launcher around it: this process becomes the launcher.  It has imported
        # neither torch nor the library and never touches HIP; the ranks are fresh children (never an exec), one per
        # GPU, environment as torch.distributed.run sets it.  Rank 0's JSON line is relayed; any rank's failure is ours.
        from maray_amd.sharding import launch_ranks
        rc, out = launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, timeout=args.rank_timeout,
                               env=dict(os.environ, MARAY_BENCH_RANK_OF=str(os.getpid())))
        lines = [ln for ln in out.splitlines() if ln.startswith('{')]
        for ln in out.splitlines():
            if not ln.startswith('{'):
                print(ln, file=sys.stderr)
        if lines:
            print(lines[-1], flush=True)
        elif rc == 0:
            rc = 1
        if rc != 0:
            print('bench.py: the %d-rank run failed (exit code %d)' % (args.gpus, rc), file=sys.stderr)
        raise SystemExit(rc)

    import torch
    import torch.distributed as dist

    import maray_amd as M
    from maray_amd.sharding import device_of_rank

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not fake_world and args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` (it launches its own '
                         'ranks) or under torch.distributed.run with --nproc-per-node N' % (args.gpus, world))
    n_dev = torch.cuda.device_count()          # counting devices does not initialise HIP
    if n_dev == 0:
        raise SystemExit('bench.py needs a MI355X: no HIP device is visible (there is no CPU fallback)')
    device, rehearsal = device_of_rank(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)), n_dev)
    if world > 1:
        # one rank per GPU: RCCL.  Fewer GPUs than ranks (a rehearsal on a one-GPU lease): the ranks share the devices and
        # the timing barrier / max go through gloo, RCCL refuses two ranks on one device.  No data-path collective either way.
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', device))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a MI355X: no HIP device is visible (there is no CPU fallback)')
    torch.cuda.set_device(device)
    red_dev = None if rehearsal else 'cuda'   # where the max-over-ranks tensor lives
    n_gpus = world
    if world == 1 and fake_world:      # rehearse rank r of N on one GPU, no process group
        n_gpus = int(fake_world)
        rank = int(os.environ.get('MARAY_BENCH_FAKE_RANK', '0'))
    solo = rank == 0 and n_gpus == 1 and world == 1

    data = open(os.path.join(ROOT, 'tests', 'golden', 'chess.maray'), 'rb').read()
    scene = M.Scene(data)
    # weak, N = 1: config 3, chess.maray regenerated at 4096 x 4096 (exact power-of-two rescale).  N > 1: the same scene
    # at N x 4096^2 pixels (8192x4096, 8192^2, 16384x8192 = config 4's width).  strong: config 4 itself, 16384^2 for
    # any N.  Rows are dealt to the ranks in interleaved 64-row blocks so that every rank sees sky and board alike:
    # no data-path collective.
    from maray_amd.sharding import interleaved_blocks, interleaved_layout, max_over_ranks, scene_scale
    sx, sy = scene_scale(n_gpus) if args.scaling == 'weak' else (16, 16)
    scene.rescale(sx, sy)
    w_img, h_total = scene.size
    tape = scene.lower(row_guards=os.environ.get('MARAY_BENCH_ROW_GUARDS', '1') != '0')
    blocks = interleaved_blocks(rank, n_gpus, h_total, BLOCK_ROWS)
    layout = interleaved_layout(rank, n_gpus, h_total, BLOCK_ROWS)     # the same rows as one launch (None: ragged, block by block)
    rows_mine = sum(b - a for a, b in blocks)

    # ---- first render of a fresh process: cold (scratch cache directories: hiprtc builds both kernels), warm (the code
    # objects the cold one left), interpreter.  Processes of their own, without PyTorch: a process that imported PyTorch
    # compiles with PyTorch's bundled hiprtc, another compiler version and so another code key than a plain one.
    cold = None
    if solo and not args.no_cold and args.backend in ('auto', 'jit') and args.scaling == 'weak':
        scratch = tempfile.mkdtemp(prefix='maray_bench_')
        env = {'MARAY_CACHE_DIR': os.path.join(scratch, 'maray'), 'AMD_COMGR_CACHE_DIR': os.path.join(scratch, 'comgr')}      # comgr keeps a cache of its own
        cold = {'cold_cache': first_render(M.BACKEND_JIT, env)}
        cold['cold_cache']['what'] = 'a fresh process, scratch MARAY_CACHE_DIR and comgr cache: hiprtc builds both kernels; frame = first 4096^2 frame into a pinned raster'
        cold['warm_cache'] = first_render(M.BACKEND_JIT, env)
        cold['warm_cache']['what'] = 'a fresh process, code objects read from MARAY_CACHE_DIR'
        cold['interpreter'] = first_render(M.BACKEND_TAPE_SMEM, env)
        cold['interpreter']['what'] = 'a fresh process, MARAY_BACKEND_TAPE_SMEM: no build; what MARAY_BACKEND_AUTO takes for a one-shot render'
        import shutil
        shutil.rmtree(scratch, ignore_errors=True)

    backends = {'tape': M.BACKEND_TAPE, 'tape-smem': M.BACKEND_TAPE_SMEM, 'jit': M.BACKEND_JIT}
    order = ['jit', 'tape-smem', 'tape'] if args.backend == 'auto' else [args.backend]
    ctx = None
    t_ctx0 = time.perf_counter()
    for name in order:
        try:
            ctx = M.Context(tape, device=device, backend=backends[name])
            backend_name = name
            break
        except M.MarayError as e:
            if args.backend != 'auto':
                raise
            print('bench.py: backend %s unavailable (%s); trying the next one' % (name, e), file=sys.stderr, flush=True)
            last = e
    if ctx is None:
        raise last
    t_ctx1 = time.perf_counter()

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

    if cold is not None:
        # this process: context creation (a build unless the default cache directory has the kernels) and the very first
        # frame of the context, outputs in HBM (one-time work included)
        step()
        torch.cuda.synchronize()
        t_first = time.perf_counter()
        cold['this_process'] = {'ctx_ms': (t_ctx1 - t_ctx0) * 1e3, 'frame_ms': (t_first - t_ctx1) * 1e3}

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dist if world > 1 else None, dt, device=red_dev)

    px_per_step = w_img * h_total
    value = px_per_step * args.steps / dt / 1e6

    # the same step frame after frame for long enough that one fence is < 0.1 % of the region and a once-a-second utilisation
    # sampler sees the GPU busy: `value` above is --steps steps between two fences, i.e. 0.7 ms at the driver's 20 x 36 us
    long_loop = None
    if args.long_steps > 0:
        fence()
        t = time.perf_counter()
        for _ in range(args.long_steps):
            step()
        fence()
        dtl = max_over_ranks(dist if world > 1 else None, time.perf_counter() - t, device=red_dev)
        long_loop = {'value': px_per_step * args.long_steps / dtl / 1e6, 'unit': 'Mpixels/s', 'steps': args.long_steps,
                     'ms_per_step': dtl / args.long_steps * 1e3, 'seconds': dtl,
                     'what': 'the timed step, %d times between two fences (barrier + synchronize, max over ranks)' % args.long_steps}

    # what this box's HBM takes when nothing is computed: hipMemsetAsync over this rank's raster on the launch stream (the
    # fill the runtime itself issues; profiles/r3_store_floor.json has hand-written stores of several widths beside it)
    attainable = None
    try:
        import ctypes
        hip = ctypes.CDLL('libamdhip64.so')
        hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
        nbytes = out8.numel()
        for _ in range(5):
            hip.hipMemsetAsync(out8.data_ptr(), 0, nbytes, stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 100
        e0.record()
        for _ in range(reps):
            hip.hipMemsetAsync(out8.data_ptr(), 0, nbytes, stream)
        e1.record()
        e1.synchronize()
        fill_ms = e0.elapsed_time(e1) / reps
        attainable = {'value': nbytes / (fill_ms * 1e-3) / 1e9, 'unit': 'GB/s', 'ms': fill_ms, 'bytes': nbytes,
                      'source': 'hipMemsetAsync of this raster, %d in a row on the launch stream, in this run' % reps}
    except (OSError, AttributeError) as e:
        attainable = {'error': str(e)}

    # roofline of the dominant (pixel) kernel: HIP events on the launch stream, around the very launch a step issues (this
    # rank's whole share; its first block when the share is ragged and goes block by block)
    a0, b0 = blocks[0]
    if layout is not None:
        k_ms = ctx.time_blocks(w_img, h_total, *layout, d_rgb8=out8.data_ptr(), reps=max(10, min(args.steps, 50)))
        px_launch = w_img * rows_mine
    else:
        k_ms = ctx.time_rows(w_img, h_total, a0, b0, d_rgb8=out8.data_ptr(), reps=max(10, min(args.steps, 50)))
        px_launch = w_img * (b0 - a0)
    census_tops = ALG_OPS_PER_PIXEL * px_launch / (k_ms * 1e-3) / 1e12
    hbm_gbs = (px_launch * 3) / (k_ms * 1e-3) / 1e9

    # parity spot check of the timed output against the committed golden: pixel (sx*i, sy*j) of the rescaled scene
    # equals pixel (i, j) of the stored one (config 1) bit for bit
    parity = None
    import hashlib
    import numpy as np
    golden = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'chess_1024.json')))
    if rank == 0:
        step()
        torch.cuda.synchronize()
        if n_gpus == 1:
            sub = out8[::sy, ::sx].contiguous().cpu().numpy()
            parity = hashlib.sha256(sub.tobytes()).hexdigest() == golden['rgb8_sha256']
        else:   # rank 0 holds rows [0,64) first: stored row 0
            sub = out8[0, ::sx].contiguous().cpu().numpy()
            parity = hashlib.sha256(sub.tobytes()).hexdigest() == golden['row_sha256']['0']

    # ---- end to end: the same rows into a host raster through the host entry point (DMA included) -----------------
    e2e = None
    if not args.no_e2e:
        tiles = []
        for a, b in blocks:                     # this rank's rows, cut into tiles of <= ~24 MiB of raster (host_pipe.cpp's own choice)
            step_rows = max(32, ((24 << 20) // (w_img * 3)) // 32 * 32)
            tiles += [(y, min(b, y + step_rows)) for y in range(a, b, step_rows)]
        pin = M.PinnedRaster(h_total, w_img)
        ctx.render_tiles(w_img, h_total, tiles, pin.array)          # warm: staging buffers allocated
        times = []
        for _ in range(7):
            if world > 1:
                dist.barrier()
            t = time.perf_counter()
            ctx.render_tiles(w_img, h_total, tiles, pin.array)
            times.append(time.perf_counter() - t)
        t_med = max_over_ranks(dist if world > 1 else None, statistics.median(times), device=red_dev)
        same = None
        if rank == 0 and n_gpus == 1:
            same = bool(np.array_equal(pin.array, out8.cpu().numpy()))
            parity = parity and same
        gbs = px_per_step * 3 / t_med / 1e9
        e2e = {'value': px_per_step / t_med / 1e6, 'unit': 'Mpixels/s', 'ms_per_frame': t_med * 1e3,
               'what': 'median of 7 calls of maray_hip_render_tiles: ROW + PIXEL kernels and the device -> host DMA of the RGB8 '
                       'raster into pinned host memory (maray_host_alloc), tile k copied under tile k+1; excludes file parse '
                       'and PNG encode (SURVEY 8(d) metric ii)',
               'tiles_per_rank': len(tiles), 'equals_device_raster': same,
               'roofline': {'bound': 'pcie', 'achieved': gbs / max(1, n_gpus), 'peak': PEAK_PCIE_GBS, 'unit': 'GB/s per GPU',
                            'frac': gbs / max(1, n_gpus) / PEAK_PCIE_GBS}}
        if solo:
            page = np.zeros((h_total, w_img, 3), np.uint8)
            ctx.render_tiles(w_img, h_total, tiles, page)
            times = []
            for _ in range(5):
                t = time.perf_counter()
                ctx.render_tiles(w_img, h_total, tiles, page)
                times.append(time.perf_counter() - t)
            e2e['pageable'] = {'value': px_per_step / statistics.median(times) / 1e6, 'unit': 'Mpixels/s',
                               'what': 'the same into an ordinary (pageable) numpy raster: DMA into a pinned ring of the context + '
                                       'one host copy by the calling thread'}
            # the façade: what the CLI and a `RenderMethod::Hip` arm call.  First call of a program: lowering + a context
            # (code objects from the process's table) + render + DMA; every later call with the same scene finds tape and
            # context in the library (gen.cpp) and pays the render alone.
            M.gen_cache_clear()
            t = time.perf_counter()
            M.gen_to_image(scene, backend=backends[backend_name], out=pin.array)
            e2e['gen_to_image_pinned_ms'] = (time.perf_counter() - t) * 1e3
            again = []
            for _ in range(5):
                t = time.perf_counter()
                M.gen_to_image(scene, backend=backends[backend_name], out=pin.array)
                again.append((time.perf_counter() - t) * 1e3)
            e2e['gen_to_image_second_call_ms'] = statistics.median(again)
            e2e['gen_to_image_equals_device_raster'] = bool(np.array_equal(pin.array, out8.cpu().numpy()))
            parity = parity and e2e['gen_to_image_equals_device_raster']
            again = []
            for _ in range(5):
                t = time.perf_counter()
                M.gen_to_image(scene, backend=backends[backend_name], out=page)
                again.append((time.perf_counter() - t) * 1e3)
            e2e['gen_to_image_pageable_first_time_this_raster_ms'] = again[0]      # the first registration of a range costs ~2.4 ms per 48 MiB
            e2e['gen_to_image_pageable_second_call_ms'] = statistics.median(again[1:])
            e2e['gen_to_image_pageable_equals_device_raster'] = bool(np.array_equal(page, out8.cpu().numpy()))
            parity = parity and e2e['gen_to_image_pageable_equals_device_raster']
            M.gen_cache_clear()
            e2e['gen_to_image_what'] = ('maray_gen_to_image, whole call.  pinned_ms: first call of a program = lowering + context '
                                        '(code objects from the process cache) + render + DMA; second_call_ms: median of 5 more calls '
                                        'with the same scene (tape and context kept by the library); the pageable raster is '
                                        'registered (pinned) for the call: pageable_first_time_this_raster_ms is the call that registers the range for '
                                        'the first time, pageable_second_call_ms the median of four more')
        pin.close()

    # HBM traffic per launch from the committed PMC profile of this very command (FETCH_SIZE x2 per the gfx950
    # correction of MI355X_MICROARCH.md + WRITE_SIZE); bench.py cannot collect PMC counters itself.  The profile names
    # the kernels it was taken on (code key = hash of the generated sources): `matches_this_build` says whether they are
    # the kernels timed here.
    traffic = None
    executed = None
    traffic_profile = None
    for rnd in ('r4', 'r3', 'r2', 'r1'):
        prof = os.path.join(ROOT, 'profiles', '%s_%s_chess4096_pmc.json' % (rnd, backend_name))
        if not os.path.exists(prof):
            continue
        pj = json.load(open(prof))
        d = pj['derived']
        traffic = (d['hbm_fetch_bytes_x2_gfx950_correction'] + d['hbm_write_bytes']) * px_launch / (4096 * 4096)
        key_now = tape.jit_code_key if backend_name == 'jit' else None
        traffic_profile = {'file': os.path.relpath(prof, ROOT), 'commit': pj.get('commit'), 'code_key': pj.get('code_key'),
                           'code_key_of_this_run': key_now,
                           'matches_this_build': (pj.get('code_key') == key_now) if key_now and pj.get('code_key') else None}
        # what the kernel actually issues (committed PMC profile): wave-level short circuits skip most of the
        # boolean-gated work, so the executed VALU stream is far shorter than the algorithmic op count
        executed = {'source': os.path.relpath(prof, ROOT), 'valu_insts_per_wave': d['valu_insts_per_wave'],
                    'salu_insts_per_wave': d['salu_insts_per_wave'],
                    'cycles_per_valu_inst_per_simd': d['cycles_per_valu_inst_per_simd'],
                    # a wave64 f64 VALU op occupies its SIMD for 4 cycles, a 32-bit one for 2: issue-port occupancy bounds
                    'valu_port_busy_frac_bounds': [2.0 / d['cycles_per_valu_inst_per_simd'], 4.0 / d['cycles_per_valu_inst_per_simd']]}
        break

    # ---- N > 1 in the default (weak) mode: config 4 itself beside the headline -- BASELINE.json names chess @16384^2 for the
    # 8-GPU run and the driver passes no --scaling -- same rows-per-rank deal (interleaved 64-row blocks, one launch),
    # same timing discipline (warm-up, barrier + synchronize on both sides, max over ranks)
    config4 = None
    if n_gpus > 1 and args.scaling == 'weak':
        sc4 = M.Scene(data)
        sc4.rescale(16, 16)
        t4 = sc4.lower()
        c4 = M.Context(t4, device=device, backend=backends[backend_name])
        lay4 = interleaved_layout(rank, n_gpus, 16384, BLOCK_ROWS)
        blk4 = interleaved_blocks(rank, n_gpus, 16384, BLOCK_ROWS)
        rows4 = sum(b - a for a, b in blk4)
        buf4 = torch.empty((rows4, 16384, 3), dtype=torch.uint8, device='cuda')

        def step4():
            if lay4 is not None:
                c4.render_blocks_device(16384, 16384, *lay4, d_rgb8=buf4.data_ptr(), stream=stream)
                return
            off = 0
            for a, b in blk4:
                c4.render_rows_device(16384, 16384, a, b, d_rgb8=buf4.data_ptr() + off * 16384 * 3, stream=stream)
                off += b - a
        for _ in range(args.warmup):
            step4()
        fence()
        t = time.perf_counter()
        for _ in range(args.steps):
            step4()
        fence()
        dt4 = max_over_ranks(dist if world > 1 else None, time.perf_counter() - t, device=red_dev)
        ok4 = None
        if rank == 0:       # rank 0's first row is stored row 0
            ok4 = hashlib.sha256(buf4[0, ::16].contiguous().cpu().numpy().tobytes()).hexdigest() == golden['row_sha256']['0']
        config4 = {'value': 16384 * 16384 * args.steps / dt4 / 1e6, 'unit': 'Mpixels/s', 'ms_per_step': dt4 / args.steps * 1e3,
                   'scaling': 'strong', 'pixels_per_step': 16384 * 16384, 'rows_per_rank': rows4, 'bit_exact_vs_golden': ok4,
                   'workload': 'SURVEY.md 8(d) config 4: data/chess.maray rescaled to 16384 x 16384, total work fixed, rows dealt to '
                               'the %d ranks in interleaved 64-row blocks, one launch per rank and step, no collective' % n_gpus}
        c4.close()
        del buf4

    cpu = None
    cpu_jit = None
    if solo and args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from oracle_ffi import Scene as OScene
        o = OScene(scene.encode())
        W, H = w_img, h_total
        threads = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        batch = max(8, threads // 4)               # rows per call: keeps every thread busy
        y_first = H // 2                           # the board's first rows: the dearest part of the image for a CPU as well
        rows, ct = 0, 0.0
        while ct < args.cpu_seconds and y_first + rows + batch <= H:
            t = time.perf_counter()
            o.render_rows(W, H, y_first + rows, y_first + rows + batch, threads=threads, want_f64=False)
            ct += time.perf_counter() - t
            rows += batch
        cpu = {'value': W * rows / ct / 1e6, 'unit': 'Mpixels/s', 'cores': threads, 'kind': 'port',
               'sample': '%d rows x %d px of the same %dx%d chess scene (rows %d..%d), oracle = restated '
                         'ParallelInterpreted (src/render.rs:35-99), %.1f s' % (rows, W, W, H, y_first, y_first + rows, ct)}
        if not args.no_cpu_jit:
            from oracle_ffi import JitBaseline
            t = time.perf_counter()
            jb = JitBaseline(o)
            build_s = time.perf_counter() - t
            batch = max(32, threads)
            rows, ct = 0, 0.0
            while ct < args.cpu_seconds and y_first + rows + batch <= H:
                t = time.perf_counter()
                jb.render_rows(W, y_first + rows, y_first + rows + batch, threads=threads)
                ct += time.perf_counter() - t
                rows += batch
            cpu_jit = {'value': W * rows / ct / 1e6, 'unit': 'Mpixels/s', 'cores': threads, 'kind': 'port',
                       'gpu_over_this': value / (W * rows / ct / 1e6),
                       'over_cpu_baseline': (W * rows / ct / 1e6) / cpu['value'],
                       'sample': '%d rows x %d px (rows %d..), stand-in for RenderMethod::JIT (src/render.rs:102-192): scene emitted '
                                 'as straight-line C like src/wasm.rs gen_expr (un-shared, one function per channel, out-of-line '
                                 'recip/step/sin), cc -O2, build %.1f s, run %.1f s; the reference README.md:47-48 puts its JIT at '
                                 '10-20x its interpreter: over_cpu_baseline is that ratio here' % (rows, W, y_first, build_s, ct)}

    if rank == 0 or (world == 1 and n_gpus > 1):
        line = {
            'metric': 'Mpixels/s on chess.maray @4096x4096',
            'value': value, 'unit': 'Mpixels/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'data/chess.maray rescaled to %d x %d (%s), rows dealt to the ranks in interleaved 64-row blocks'
                                   % (w_img, h_total, 'SURVEY.md 8(d) config 3 at N=1; N x 4096^2 pixels of the same scene at N>1'
                                      if args.scaling == 'weak' else 'SURVEY.md 8(d) config 4: total work fixed'),
                       'backend': backend_name, 'kernel': ctx.kernel_name, 'pixels_per_step': px_per_step,
                       'tape_ops_per_pixel': tape.info['n_pix_ops'], 'parallelism': 'row tiles, no collective',
                       'bit_exact_vs_golden': parity, 'code_key': tape.jit_code_key if backend_name == 'jit' else None},
            'roofline': {'bound': 'hbm', 'achieved': hbm_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                         'frac': hbm_gbs / PEAK_HBM_GBS, 'traffic': traffic, 'traffic_profile': traffic_profile,
                         'attainable_peak': attainable,
                         'frac_of_attainable': (hbm_gbs / attainable['value']) if attainable and attainable.get('value') else None,
                         'kernel_ms': k_ms, 'alg_bytes_per_pixel': 3, 'pixels_per_launch': px_launch,
                         'note': 'achieved = 3 B/pixel (SURVEY 8(d): the RGB8 pixel written is the only mandatory HBM '
                                 'traffic) x pixels of one launch / its HIP-event time; traffic = FETCH_SIZE x2 + WRITE_SIZE '
                                 'of the committed PMC profile of this command, scaled to this launch',
                         'valu_f64_census': {'achieved': census_tops, 'peak': PEAK_F64_TOPS, 'unit': 'TFLOP/s',
                                             'frac': census_tops / PEAK_F64_TOPS, 'alg_ops_per_pixel': ALG_OPS_PER_PIXEL,
                                             'note': 'SURVEY 8(d) census of the scene DAG (10,241 f64 ops per pixel) / kernel '
                                                     'time; far above 1 because regions gated by a boolean that a y-only bound '
                                                     'proves 0 over a rectangle of 64 x 32 pixels are skipped, and half of what remains is '
                                                     'boolean algebra on lane masks (scalar unit)'},
                         'executed': executed},
            'long_loop': long_loop,
            'rehearsal': ('%d ranks share %d visible device(s): sharding, launches and timing discipline are the real ones, '
                          'the ranks queue for one GPU, so this is not a scaling figure (gloo for the barrier / max)' % (world, n_dev))
                         if rehearsal else None,
            'end_to_end': e2e,
            'config4_strong': config4,
            'cold_first_render_ms': cold,
            'cpu_baseline': cpu,
            'cpu_baseline_jit': cpu_jit,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
