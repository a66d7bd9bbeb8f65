#!/usr/bin/env python3
"""tools/gpu_fuzz.py LO HI [W H] — one-off wider sweep of tests/test_fuzz.py's random scenes on the GPU box: every evaluator
against the oracle, f64 planes bit for bit and RGB8 byte for byte.  Prints progress; exits non-zero on a mismatch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np
import maray_amd as M
import scenes
from oracle_ffi import Scene as OScene
from test_fuzz import lowered, W, H
from test_lowering import same_f64
import tape_eval

lo, hi = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 4:
    W, H = int(sys.argv[3]), int(sys.argv[4])          # another geometry: several tiles per row, whole groups of rows
tex = scenes.textures(scale=64)
bad = done = guarded = 0
t0 = time.time()
for seed in range(lo, hi):
    n_tex = 2 if seed % 3 == 0 else 0
    data, tape = lowered(seed, n_tex, W, H)
    if tape is None:
        continue
    t = tex if n_tex else None
    want8, want64 = OScene(data).render_rows(W, H, 0, H, t)
    guarded += tape_eval.guards_reading_y(tape)[0] > 0
    for b in ((M.BACKEND_JIT, M.BACKEND_TAPE_SMEM, M.BACKEND_TAPE) if os.environ.get('MARAY_FUZZ_ALL_BACKENDS') == '1' else (M.BACKEND_JIT, M.BACKEND_TAPE_SMEM)):
        ctx = M.Context(tape, textures=t, backend=b)
        got8, got64 = ctx.render_rows(W, H, 0, H)
        ctx.close()
        if not (same_f64(got64, want64) and np.array_equal(got8, want8)):
            bad += 1
            print('MISMATCH seed %d backend %d' % (seed, b), flush=True)
    done += 1
    if done % 20 == 0:
        print('%d scenes (%d with guards), %d mismatches, %.0f s' % (done, guarded, bad, time.time() - t0), flush=True)
print('done: %d scenes (%d with guards), %d mismatches' % (done, guarded, bad))
sys.exit(1 if bad else 0)
