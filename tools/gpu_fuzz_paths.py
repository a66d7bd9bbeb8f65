#!/usr/bin/env python3
"""tools/gpu_fuzz_paths.py LO HI [W H] — random scenes through the OTHER entry points of a context, each against the context's own
full render (which the other sweeps hold against the oracle): rows in two unequal parts, row tiles into a host raster (pinned and
pageable), blocks of rows a stride apart on the device (a rank's interleaved share), the façade (maray_gen_to_image).  RGB8, byte
for byte; both specialised kernels and the interpreter.  Exits non-zero on a mismatch."""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np
import maray_amd as M
import scenes
from test_fuzz import lowered

lo, hi = int(sys.argv[1]), int(sys.argv[2])
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (600, 96)
hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
dbuf = C.c_void_p()
assert hip.hipMalloc(C.byref(dbuf), W * H * 3) == 0
tex = scenes.textures(scale=64)
bad = done = 0
t0 = time.time()
for seed in range(lo, hi):
    n_tex = 2 if seed % 3 == 0 else 0
    data, tape = lowered(seed, n_tex, W, H)
    if tape is None:
        continue
    t = tex if n_tex else None
    for b in (M.BACKEND_JIT, M.BACKEND_TAPE_SMEM):
        ctx = M.Context(tape, textures=t, backend=b)
        full, _ = ctx.render_rows(W, H, 0, H, want_f64=False)
        what = []
        cut = 1 + seed % (H - 1)
        a, _ = ctx.render_rows(W, H, 0, cut, want_f64=False)
        c, _ = ctx.render_rows(W, H, cut, H, want_f64=False)
        what.append(('two parts at %d' % cut, np.concatenate([a, c])))
        step = 8 * (1 + seed % 3)
        tiles = [(y, min(H, y + step)) for y in range(0, H, step)]
        with M.PinnedRaster(H, W) as pin:
            ctx.render_tiles(W, H, tiles, pin.array)
            what.append(('tiles of %d rows, pinned' % step, pin.array.copy()))
        page = np.zeros((H, W, 3), np.uint8)
        ctx.render_tiles(W, H, tiles[::-1], page)
        what.append(('tiles backwards, pageable', page))
        # blocks: rank r of n takes block_rows rows every n * block_rows (sharding.py's deal)
        n, br = 2 + seed % 3, 8
        for r in range(n):
            nb = len([k for k in range(H) if r * br + k * n * br + br <= H])      # whole blocks of this rank (a ragged last one goes through the range entry point)
            if nb == 0:
                continue
            hip.hipMemset(dbuf, 0, W * H * 3)
            ctx.render_blocks_device(W, H, r * br, br, n * br, nb, d_rgb8=dbuf.value)
            got = np.empty((nb * br, W, 3), np.uint8)
            hip.hipDeviceSynchronize()
            hip.hipMemcpy(got.ctypes.data, dbuf, nb * br * W * 3, 2)
            for k in range(nb):
                y = r * br + k * n * br
                if not np.array_equal(got[k * br:(k + 1) * br], full[y:y + br]):
                    bad += 1
                    print('MISMATCH seed %d backend %d blocks rank %d of %d block %d' % (seed, b, r, n, k), flush=True)
                    break
        for name, got in what:
            if not np.array_equal(got, full):
                bad += 1
                print('MISMATCH seed %d backend %d: %s' % (seed, b, name), flush=True)
        ctx.close()
    s = M.Scene(data)
    got = M.gen_to_image(s, textures=t, backend=M.BACKEND_JIT, size=(W, H))
    if not np.array_equal(got, full):
        bad += 1
        print('MISMATCH seed %d: gen_to_image' % seed, flush=True)
    done += 1
    if done % 20 == 0:
        print('%d scenes, %d mismatches, %.0f s' % (done, bad, time.time() - t0), flush=True)
M.gen_cache_clear()
print('done: %d scenes x 2 back-ends through rows in parts, tiles (pinned, pageable), blocks and the façade at %d x %d: %d mismatches' % (done, W, H, bad))
sys.exit(1 if bad else 0)
