#!/usr/bin/env python3
"""tools/time_bands.py — pixel-kernel time per 512-row band of chess @4096^2 (HIP events, kernel only): shows how the
cost follows the content (sky rows skip every shape; board rows enter a few per tile)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import maray_amd as M
s = M.Scene(open(os.path.join(ROOT, 'tests', 'golden', 'chess.maray'), 'rb').read())
s.rescale(4, 4)
t = s.lower()
name = sys.argv[1] if len(sys.argv) > 1 else 'jit'
ctx = M.Context(t, backend={'jit': M.BACKEND_JIT, 'tape': M.BACKEND_TAPE, 'tape-smem': M.BACKEND_TAPE_SMEM}[name])
out = {}
for y0 in range(0, 4096, 512):
    ms = ctx.time_rows(4096, 4096, y0, y0 + 512, reps=20)
    out['%d-%d' % (y0, y0 + 512)] = round(ms * 1e3, 2)
out['all'] = round(ctx.time_rows(4096, 4096, 0, 4096, reps=20) * 1e3, 2)
print(json.dumps({'backend': name, 'kernel': ctx.kernel_name, 'us_per_band': out}))
