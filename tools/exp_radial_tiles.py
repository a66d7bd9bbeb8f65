#!/usr/bin/env python3
"""tools/exp_radial_tiles.py — config 2 (radial gradient, RGB8) at several sizes, by tiles per wavefront (MARAY_JIT_TILES)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import maray_amd as M
import scenes
from marayb import encode
hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
buf = C.c_void_p()
assert hip.hipMalloc(C.byref(buf), 8192 * 8192 * 3) == 0
out = {}
for size in (1024, 2048, 4096, 8192):
    tape = M.Scene(encode((size, size), scenes.radial_gradient())).lower()
    row = {}
    for tiles in ('1', '2', '4', '8', '16', ''):
        if tiles:
            os.environ['MARAY_JIT_TILES'] = tiles
        else:
            os.environ.pop('MARAY_JIT_TILES', None)
        ctx = M.Context(tape, backend=M.BACKEND_JIT)
        us = ctx.time_rows(size, size, 0, size, d_rgb8=buf.value, reps=30) * 1e3
        ctx.close()
        row[tiles or 'default'] = round(us, 2)
    out[size] = row
    print(json.dumps({size: row}), flush=True)
