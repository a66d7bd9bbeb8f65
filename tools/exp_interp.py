#!/usr/bin/env python3
"""tools/exp_interp.py CONFIG... — tools/exp_pixels.py for the tape interpreter (MARAY_TAPE_* knobs): pixel kernel and whole
step on chess @4096^2, golden hash."""
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
        for b, tag in ((M.BACKEND_TAPE_SMEM, 'smem'), (M.BACKEND_TAPE, 'lds')):
            ctx = M.Context(s.lower(), backend=b)
            got8, _ = ctx.render_rows(4096, 4096, 0, 4096, want_f64=False)
            out[tag + '_parity'] = hashlib.sha256(np.ascontiguousarray(got8[::4, ::4]).tobytes()).hexdigest() == golden['rgb8_sha256']
            out[tag + '_pix_ms'] = round(ctx.time_rows(4096, 4096, 0, 4096, d_rgb8=dbuf.value, reps=5), 3)
            for _ in range(2):
                ctx.render_rows_device(4096, 4096, 0, 4096, d_rgb8=dbuf.value)
            hip.hipDeviceSynchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                ctx.render_rows_device(4096, 4096, 0, 4096, d_rgb8=dbuf.value)
            hip.hipDeviceSynchronize()
            out[tag + '_step_ms'] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
            ctx.close()
    except Exception as e:      # noqa: BLE001
        out['error'] = str(e)[-400:]
    for k in env:
        del os.environ[k]
    print(json.dumps(out), flush=True)


for arg in sys.argv[1:]:
    name, _, rest = arg.partition(':')
    run(name, dict(kv.split('=', 1) for kv in rest.split(',') if kv))
