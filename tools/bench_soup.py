#!/usr/bin/env python3
"""tools/bench_soup.py [N] [mixed|colours] — a scene of another kind than chess: N random textured triangles (tests/fuzz_scenes.py,
polygon_soup) at 4096 x 4096.  Build time, kernel time of both evaluators, parity of a band of rows with the oracle."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np
import maray_amd as M
import fuzz_scenes
from marayb import encode
from oracle_ffi import Scene as OScene
from test_lowering import same_f64

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
w = h = 4096
mixed = {'mixed': True, 'colours': 'colours'}.get(sys.argv[2] if len(sys.argv) > 2 else '', False)   # mixed: shapes shared by two
                                                            # channels' trees; colours: channel = max_i(shape_i * c_i)
data = encode((w, h), fuzz_scenes.polygon_soup(1, n, w, h, mixed=mixed))
t0 = time.time()
tape = M.Scene(data).lower()
out = {'shapes': n, 'mixed': mixed, 'lower_s': round(time.time() - t0, 2), 'pix_ops': tape.info['n_pix_ops'], 'row_ops': tape.info['n_row_ops'],
       'y_values': tape.info['n_yvals'], 'skip_ops': tape.info['skip_ops']}
want8, want64 = OScene(data).render_rows(w, h, 2000, 2016)
for name, b in (('jit', M.BACKEND_JIT), ('tape-smem', M.BACKEND_TAPE_SMEM)):
    t0 = time.time()
    ctx = M.Context(tape, backend=b)
    build = time.time() - t0
    got8, got64 = ctx.render_rows(w, h, 2000, 2016)
    ok = bool(same_f64(got64, want64) and np.array_equal(got8, want8))
    ms = ctx.time_rows(w, h, 0, h, reps=5)
    ctx.close()
    out[name] = {'create_s': round(build, 1), 'kernel_ms': round(ms, 3), 'mpx_s': round(w * h / ms / 1e3), 'bit_exact_rows_2000_2016': ok}
print(json.dumps(out))
