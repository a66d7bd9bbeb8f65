#!/usr/bin/env python3
"""tools/exp_launch_overhead.py — host cost of one launch of the specialised path (ROW + PIXEL kernels): steps over a
tiny row range are bound by the host, not the GPU.  Prints us per step for 8 rows and for the whole 4096^2 frame."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import maray_amd as M  # noqa: E402

hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), 4096 * 4096 * 3) == 0
s = M.Scene(open(os.path.join(ROOT, 'tests', 'golden', 'chess.maray'), 'rb').read())
s.rescale(4, 4)
ctx = M.Context(s.lower(), backend=M.BACKEND_JIT)
out = {}
for name, (y0, y1) in (('8 rows', (2048, 2056)), ('frame', (0, 4096))):
    for _ in range(20):
        ctx.render_rows_device(4096, 4096, y0, y1, d_rgb8=d.value)
    hip.hipDeviceSynchronize()
    t = time.perf_counter()
    for _ in range(500):
        ctx.render_rows_device(4096, 4096, y0, y1, d_rgb8=d.value)
    t_issue = time.perf_counter() - t
    hip.hipDeviceSynchronize()
    t_all = time.perf_counter() - t
    out[name] = {'host_issue_us_per_step': round(t_issue / 500 * 1e6, 2), 'us_per_step': round(t_all / 500 * 1e6, 2)}
print(json.dumps(out))
