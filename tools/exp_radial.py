#!/usr/bin/env python3
"""tools/exp_radial.py CONFIG...  — A/B of specialised-kernel variants on config 2 (radial gradient, 8192^2, RGB8 in HBM), one
process, one line of JSON per config.  CONFIG = NAME[:VAR=VALUE[,VAR=VALUE...]] as in tools/exp_pixels.py."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402

import maray_amd as M  # noqa: E402
import scenes  # noqa: E402
from marayb import encode  # noqa: E402

N = 8192
hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
dbuf = C.c_void_p()
assert hip.hipMalloc(C.byref(dbuf), N * N * 3) == 0
data = encode((N, N), scenes.radial_gradient())
yy, xx = np.mgrid[0:64, 0:N].astype(np.float64)


def run(name, env):
    for k, v in env.items():
        os.environ[k] = v
    out = {'config': name, 'env': env}
    try:
        ctx = M.Context(M.Scene(data).lower(), backend=M.BACKEND_JIT)
        got8, _ = ctx.render_rows(N, N, 4000, 4064, want_f64=False)
        want = np.minimum(np.floor(np.sqrt(xx * xx + (yy + 4000) ** 2)), 255).astype(np.uint8)
        out['parity'] = bool(np.array_equal(got8[:, :, 0], want))
        ms = ctx.time_rows(N, N, 0, N, d_rgb8=dbuf.value, reps=20)
        out['us'] = round(ms * 1e3, 2)
        out['TB_s'] = round(N * N * 3 / (ms * 1e-3) / 1e12, 3)
        ctx.close()
    except Exception as e:      # noqa: BLE001
        out['error'] = str(e)[-400:]
    for k in env:
        del os.environ[k]
    print(json.dumps(out), flush=True)


for arg in sys.argv[1:]:
    name, _, rest = arg.partition(':')
    run(name, dict(kv.split('=', 1) for kv in rest.split(',') if kv))
