#!/usr/bin/env python3
"""tools/exp_sizes.py — chess rescaled to 1024^2 ... 16384^2 on one GPU, specialised kernels, frame after frame with outputs in
HBM: us per step, Mpixel/s, and the golden hash at every k-th pixel."""
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
assert hip.hipMalloc(C.byref(dbuf), 16384 * 16384 * 3) == 0
out = {}
for k in (1, 2, 4, 8, 16):
    n = 1024 * k
    s = M.Scene(data)
    if k > 1:
        s.rescale(k, k)
    ctx = M.Context(s.lower(), backend=M.BACKEND_JIT)
    got8, _ = ctx.render_rows(n, n, 0, n, want_f64=False)
    ok = hashlib.sha256(np.ascontiguousarray(got8[::k, ::k]).tobytes()).hexdigest() == golden['rgb8_sha256']
    del got8
    for _ in range(5):
        ctx.render_rows_device(n, n, 0, n, d_rgb8=dbuf.value)
    hip.hipDeviceSynchronize()
    reps = 200 if k <= 4 else 40
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.render_rows_device(n, n, 0, n, d_rgb8=dbuf.value)
    hip.hipDeviceSynchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    out['%d^2' % n] = {'us_per_step': round(us, 1), 'pixel_kernel_us': round(ctx.time_rows(n, n, 0, n, d_rgb8=dbuf.value, reps=10) * 1e3, 1),
                       'Mpx_s': round(n * n / us), 'parity': ok}
    ctx.close()
print(json.dumps(out))
