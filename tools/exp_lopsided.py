#!/usr/bin/env python3
"""tools/exp_lopsided.py — a scene that is busy on one side only (300 textured triangles in the left half of a 4096^2 image):
pixel kernel with and without the rotation of a row's strips over the XCDs (MARAY_JIT_SWIZZLE), one process."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np  # noqa: E402

import fuzz_scenes  # noqa: E402
import maray_amd as M  # noqa: E402
from marayb import encode  # noqa: E402

hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), 4096 * 4096 * 3) == 0
data = encode((4096, 4096), fuzz_scenes.polygon_soup(1, 300, 2048, 4096, mixed=False))
tape = M.Scene(data).lower()
out = {}
ref = None
for name, env in (('in place (default)', {}), ('rotated', {'MARAY_JIT_SWIZZLE': '1'}), ('in place again', {})):
    os.environ.update(env)
    ctx = M.Context(tape, backend=M.BACKEND_JIT)
    got, _ = ctx.render_rows(4096, 4096, 1000, 1064, want_f64=False)
    ref = got if ref is None else ref
    out[name] = {'pixel_kernel_us': round(ctx.time_rows(4096, 4096, 0, 4096, d_rgb8=d.value, reps=20) * 1e3, 1), 'same_pixels': bool(np.array_equal(got, ref))}
    ctx.close()
    for k in env:
        del os.environ[k]
print(json.dumps(out))
