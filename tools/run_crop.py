#!/usr/bin/env python3
"""tools/run_crop.py SCENE CROP [REPS] — launches of the specialised kernels over one 4096 x 4096 crop, for rocprofv3.

SCENE: chess (config 3) | allops (config 3b) | radial (config 2) | textured (config 5) | soupN (N random textured triangles).  CROP (chess only): frame | sky | board — the
frame itself, or 4096^2 pixels of nothing but sky / nothing but board rows (chess stretched 16x vertically).
Backend from MARAY_BENCH_BACKEND (jit | tape-smem | tape; default jit).  Prints the HIP-event time per launch of the
pixel kernel."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import maray_amd as M  # noqa: E402

scene_name, crop = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
dbuf = C.c_void_p()
assert hip.hipMalloc(C.byref(dbuf), 4096 * 4096 * 3) == 0
textures = None
backend = {'jit': M.BACKEND_JIT, 'tape-smem': M.BACKEND_TAPE_SMEM, 'tape': M.BACKEND_TAPE}[os.environ.get('MARAY_BENCH_BACKEND', 'jit')]
if scene_name == 'chess':
    s = M.Scene(open(os.path.join(ROOT, 'tests', 'golden', 'chess.maray'), 'rb').read())
    sy = 4 if crop == 'frame' else 16
    s.rescale(4, sy)
    h = 1024 * sy
    y0 = {'frame': 0, 'sky': 0, 'board': 8192}[crop]
elif scene_name.startswith('soup'):        # soup1000, soup300 ...: tests/fuzz_scenes.py polygon_soup, one tree for the three channels
    import fuzz_scenes
    from marayb import encode
    s = M.Scene(encode((4096, 4096), fuzz_scenes.polygon_soup(1, int(scene_name[4:]), 4096, 4096, mixed=False)))
    h, y0 = 4096, 0
else:
    import scenes
    from marayb import encode
    if scene_name == 'textured':            # config 5: two textures, six App ops per pixel
        s = M.Scene(encode((4096, 4096), scenes.textured(4096)))
        textures = scenes.textures(1)
    else:
        s = M.Scene(encode((4096, 4096), scenes.all_ops(4096, 4096) if scene_name == 'allops' else scenes.radial_gradient()))
    h, y0 = 4096, 0
tape = s.lower()
ctx = M.Context(tape, textures=textures, backend=backend)
for _ in range(reps):
    ctx.render_rows_device(4096, h, y0, y0 + 4096, d_rgb8=dbuf.value)
hip.hipDeviceSynchronize()
ms = ctx.time_rows(4096, h, y0, y0 + 4096, d_rgb8=dbuf.value, reps=reps)
print(json.dumps({'scene': scene_name, 'crop': crop, 'kernel': ctx.kernel_name, 'pixel_kernel_us': round(ms * 1e3, 2),
                  'pixels': 4096 * 4096, 'tape_ops_per_pixel': tape.info['n_pix_ops'], 'alg_ops': tape.info['alg_ops'],
                  'code_key': tape.jit_code_key if backend == M.BACKEND_JIT else None, 'library': M.version(),
                  'op_histogram': tape.info['op_histogram']}))
