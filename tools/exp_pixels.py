#!/usr/bin/env python3
"""tools/exp_pixels.py CONFIG...  — A/B of specialised-kernel variants on chess, one process, one line of JSON per config.

CONFIG = NAME[:VAR=VALUE[,VAR=VALUE...]] (environment knobs of DESIGN.md 7.1, read at context creation / launch).
Per config: pixel-kernel time (HIP events, pixel kernel only) of the 4096^2 frame, of 4096^2 pixels of sky and of
board (chess stretched 16x vertically), the whole step (ROW + PIXEL kernels, outputs in HBM) and the build time;
the frame is checked against the golden hash.  EXP_LIBM=1: configs 3b and 2 @4096^2 as well."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import maray_amd as M  # noqa: E402

hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
data = open(os.path.join(ROOT, 'tests', 'golden', 'chess.maray'), 'rb').read()
golden = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'chess_1024.json')))
dbuf = C.c_void_p()
assert hip.hipMalloc(C.byref(dbuf), 4096 * 4096 * 3) == 0


def run(name, env):
    for k, v in env.items():
        os.environ[k] = v
    out = {'config': name, 'env': env}
    try:
        s = M.Scene(data)
        s.rescale(4, 4)
        t0 = time.perf_counter()
        ctx = M.Context(s.lower(), backend=M.BACKEND_JIT)
        out['build_s'] = round(time.perf_counter() - t0, 2)
        got8, _ = ctx.render_rows(4096, 4096, 0, 4096, want_f64=False)
        out['parity'] = hashlib.sha256(np.ascontiguousarray(got8[::4, ::4]).tobytes()).hexdigest() == golden['rgb8_sha256']
        out['pix_us'] = round(ctx.time_rows(4096, 4096, 0, 4096, d_rgb8=dbuf.value, reps=30) * 1e3, 2)
        for _ in range(5):
            ctx.render_rows_device(4096, 4096, 0, 4096, d_rgb8=dbuf.value)
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ctx.render_rows_device(4096, 4096, 0, 4096, d_rgb8=dbuf.value)
        hip.hipDeviceSynchronize()
        out['step_us'] = round((time.perf_counter() - t0) / 200 * 1e6, 2)
        ctx.close()
        if not os.environ.get('EXP_FRAME_ONLY'):
            s = M.Scene(data)
            s.rescale(4, 16)
            ctx = M.Context(s.lower(), backend=M.BACKEND_JIT)
            out['sky_us'] = round(ctx.time_rows(4096, 16384, 0, 4096, d_rgb8=dbuf.value, reps=30) * 1e3, 2)
            out['board_us'] = round(ctx.time_rows(4096, 16384, 8192, 12288, d_rgb8=dbuf.value, reps=30) * 1e3, 2)
            ctx.close()
        if os.environ.get('EXP_LIBM'):          # config 3b (sin / exp / ln / sqrt) and config 2 (sqrt) @4096^2, checked against the interpreter
            sys.path.insert(0, os.path.join(ROOT, 'tests'))
            import scenes
            from marayb import encode
            for nm, col in (('allops', scenes.all_ops(4096, 4096)), ('radial', scenes.radial_gradient())):
                t = M.Scene(encode((4096, 4096), col)).lower()
                ctx = M.Context(t, backend=M.BACKEND_JIT)
                ref = M.Context(t, backend=M.BACKEND_TAPE_SMEM)
                a8, a64 = ctx.render_rows(4096, 4096, 1000, 1256, want_f64=True)
                b8, b64 = ref.render_rows(4096, 4096, 1000, 1256, want_f64=True)
                out[nm + '_parity'] = bool(np.array_equal(a8, b8) and np.array_equal(a64.view(np.uint64), b64.view(np.uint64)))
                out[nm + '_us'] = round(ctx.time_rows(4096, 4096, 0, 4096, d_rgb8=dbuf.value, reps=20) * 1e3, 2)
                ctx.close()
                ref.close()
    except Exception as e:      # noqa: BLE001
        out['error'] = str(e)[-400:]
    for k in env:
        del os.environ[k]
    print(json.dumps(out), flush=True)


for arg in sys.argv[1:]:
    name, _, rest = arg.partition(':')
    run(name, dict(kv.split('=', 1) for kv in rest.split(',') if kv))
