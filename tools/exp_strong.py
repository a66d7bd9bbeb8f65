#!/usr/bin/env python3
"""tools/exp_strong.py CONFIG... — tools/exp_pixels.py for config 4 on one GPU: chess @16384^2 (8 blocks per row), pixel kernel and
whole step, every 16th pixel checked against the golden hash."""
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

N = 16384
hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
data = open(os.path.join(ROOT, 'tests', 'golden', 'chess.maray'), 'rb').read()
golden = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'chess_1024.json')))
dbuf = C.c_void_p()
assert hip.hipMalloc(C.byref(dbuf), N * N * 3) == 0


def run(name, env):
    for k, v in env.items():
        os.environ[k] = v
    out = {'config': name, 'env': env}
    try:
        s = M.Scene(data)
        s.rescale(16, 16)
        ctx = M.Context(s.lower(), backend=M.BACKEND_JIT)
        got8, _ = ctx.render_rows(N, N, 0, N, want_f64=False)
        out['parity'] = hashlib.sha256(np.ascontiguousarray(got8[::16, ::16]).tobytes()).hexdigest() == golden['rgb8_sha256']
        del got8
        out['pix_us'] = round(ctx.time_rows(N, N, 0, N, d_rgb8=dbuf.value, reps=20) * 1e3, 1)
        for _ in range(5):
            ctx.render_rows_device(N, N, 0, N, d_rgb8=dbuf.value)
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            ctx.render_rows_device(N, N, 0, N, d_rgb8=dbuf.value)
        hip.hipDeviceSynchronize()
        out['step_us'] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
        out['Mpx_s'] = round(N * N / out['step_us'])
        ctx.close()
    except Exception as e:      # noqa: BLE001
        out['error'] = str(e)[-400:]
    for k in env:
        del os.environ[k]
    print(json.dumps(out), flush=True)


for arg in sys.argv[1:]:
    name, _, rest = arg.partition(':')
    run(name, dict(kv.split('=', 1) for kv in rest.split(',') if kv))
