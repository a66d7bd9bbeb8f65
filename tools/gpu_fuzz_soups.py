#!/usr/bin/env python3
"""tools/gpu_fuzz_soups.py LO HI [N] — one-off wider sweep of tests/test_fuzz.py's polygon soups on the GPU box (N shapes, three
mixing modes by seed): the specialised kernels and the scalar-cache interpreter against the oracle, f64 planes bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np
import maray_amd as M
from fuzz_scenes import polygon_soup
from marayb import encode
from oracle_ffi import Scene as OScene
from test_lowering import same_f64

lo, hi = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 70
w, h = 1024, 136
bad = 0
t0 = time.time()
for seed in range(lo, hi):
    data = encode((w, h), polygon_soup(seed, n, w, h, mixed=(True, False, 'colours')[seed % 3]))
    tape = M.Scene(data).lower()
    want8, want64 = OScene(data).render_rows(w, h, 0, h)
    for b in (M.BACKEND_JIT, M.BACKEND_TAPE_SMEM):
        ctx = M.Context(tape, backend=b)
        got8, got64 = ctx.render_rows(w, h, 0, h)
        ctx.close()
        if not (same_f64(got64, want64) and np.array_equal(got8, want8)):
            bad += 1
            print('MISMATCH seed %d backend %d' % (seed, b), flush=True)
    print('seed %d done, %d mismatches, %.0f s' % (seed, bad, time.time() - t0), flush=True)
print('done: %d soups, %d mismatches' % (hi - lo, bad))
sys.exit(1 if bad else 0)
