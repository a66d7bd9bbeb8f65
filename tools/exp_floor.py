#!/usr/bin/env python3
"""tools/exp_floor.py ROWS [N] — N launches of the specialised path over ROWS rows of chess @4096^2 (for rocprofv3): what a
launch costs when there is next to nothing to compute."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import maray_amd as M  # noqa: E402

rows = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), 4096 * 4096 * 3) == 0
s = M.Scene(open(os.path.join(ROOT, 'tests', 'golden', 'chess.maray'), 'rb').read())
s.rescale(4, 4)
ctx = M.Context(s.lower(), backend=M.BACKEND_JIT)
for _ in range(n):
    ctx.render_rows_device(4096, 4096, 2048, 2048 + rows, d_rgb8=d.value)
hip.hipDeviceSynchronize()
