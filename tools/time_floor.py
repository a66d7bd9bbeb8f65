#!/usr/bin/env python3
"""tools/time_floor.py — pixel-kernel time for 4096 x 4096 pixels of nothing but sky (chess stretched 16x vertically,
rows 0..4095): what a launch costs when every shape is skipped, to set beside the real frame."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import maray_amd as M
data = open(os.path.join(ROOT, 'tests', 'golden', 'chess.maray'), 'rb').read()
out = {}
for name, (sx, sy), rows in (('frame', (4, 4), (0, 4096)), ('sky only', (4, 16), (0, 4096)), ('board only', (4, 16), (8192, 12288))):
    s = M.Scene(data)
    s.rescale(sx, sy)
    ctx = M.Context(s.lower(), backend=M.BACKEND_JIT)
    out[name] = round(ctx.time_rows(4096, 1024 * sy, rows[0], rows[1], reps=20) * 1e3, 2)
    ctx.close()
print(json.dumps({'us_per_4096x4096_launch': out}))
