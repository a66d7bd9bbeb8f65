#!/usr/bin/env python3
"""tools/exp_tiles.py — configs 3b (all ops) and 5 (two textures), RGB8, by tiles per wavefront (MARAY_JIT_TILES)."""
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
for name, size in (('allops', 1024), ('allops', 2048), ('allops', 4096), ('textured', 1024), ('textured', 4096), ('textured', 8192)):
    col = scenes.all_ops(size, size) if name == 'allops' else scenes.textured(size)
    tex = scenes.textures(1) if name == 'textured' else None
    tape = M.Scene(encode((size, size), col)).lower()
    row = {}
    for tiles in ('1', '2', '4', '8', ''):
        if tiles:
            os.environ['MARAY_JIT_TILES'] = tiles
        else:
            os.environ.pop('MARAY_JIT_TILES', None)
        ctx = M.Context(tape, textures=tex, backend=M.BACKEND_JIT)
        row[tiles or 'default'] = round(ctx.time_rows(size, size, 0, size, d_rgb8=buf.value, reps=20) * 1e3, 2)
        ctx.close()
    print(json.dumps({name + ' %d' % size: row}), flush=True)
