#!/usr/bin/env python3
"""tools/exp_e2e_tiles.py — chess @4096^2 into a pinned host raster (maray_hip_render_tiles through render_rows_into) for
several tile sizes of the host pipeline (MARAY_TILE_MIB) and both evaluators: ms per frame, median of 9."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import maray_amd as M  # noqa: E402

s = M.Scene(open(os.path.join(ROOT, 'tests', 'golden', 'chess.maray'), 'rb').read())
s.rescale(4, 4)
tape = s.lower()
pin = M.PinnedRaster(4096, 4096)
out = {}
for backend, name in ((M.BACKEND_JIT, 'jit'), (M.BACKEND_TAPE_SMEM, 'tape-smem')):
    for mib in (4, 8, 16, 24, 48, 64):
        os.environ['MARAY_TILE_MIB'] = str(mib)
        ctx = M.Context(tape, backend=backend)
        ts = []
        for _ in range(11):
            t = time.perf_counter()
            ctx.render_rows_into(4096, 4096, 0, 4096, pin.array)
            ts.append(time.perf_counter() - t)
        ctx.close()
        out['%s %d MiB' % (name, mib)] = round(sorted(ts[2:])[len(ts[2:]) // 2] * 1e3, 3)
print(json.dumps(out))
