#!/usr/bin/env python3
"""tools/gpu_fuzz_products.py LO HI [N] — one-off sweep of the product / square / root / reciprocal soups (tests/fuzz_scenes.py:
product_soup) on the GPU box: specialised kernels, scalar-cache interpreter and the guard-free lowering against the oracle on every
pixel, u8 and f64 planes bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import maray_amd as M
from fuzz_scenes import product_soup
from marayb import encode
from oracle_ffi import Scene as OScene
from test_lowering import same_f64

lo, hi = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
w, h = 768, 96
bad = 0
t0 = time.time()
for s0 in range(lo, hi, 8):
    seeds = list(range(s0, min(hi, s0 + 8)))
    datas = [encode((w, h), product_soup(seed, n, w, h)) for seed in seeds]
    tapes = [M.Scene(d).lower() for d in datas]
    with ThreadPoolExecutor(8) as pool:
        jits = list(pool.map(lambda t: M.Context(t, backend=M.BACKEND_JIT), tapes))
        wants = list(pool.map(lambda d: OScene(d).render_rows(w, h, 0, h, threads=2), datas))
    for seed, data, tape, jit, (want8, want64) in zip(seeds, datas, tapes, jits, wants):
        for name, ctx in (('jit', jit), ('tape-smem', M.Context(tape, backend=M.BACKEND_TAPE_SMEM)),
                          ('guard-free', M.Context(M.Scene(data).lower(skips=False), backend=M.BACKEND_TAPE_SMEM))):
            got8, got64 = ctx.render_rows(w, h, 0, h)
            ctx.close()
            if not (same_f64(got64, want64) and np.array_equal(got8, want8)):
                bad += 1
                print('MISMATCH seed %d %s' % (seed, name), flush=True)
    print('seeds %d..%d done, %d mismatches, %.0f s' % (seeds[0], seeds[-1], bad, time.time() - t0), flush=True)
print('done: %d product soups of %d shapes at %d x %d, 3 evaluations each against the oracle on every pixel: %d mismatches' % (hi - lo, n, w, h, bad))
sys.exit(1 if bad else 0)
