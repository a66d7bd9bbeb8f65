"""Throughput of the other SURVEY §8(d) configurations on one MI355X (outputs resident in HBM, HIP-event timing).
Prints one JSON object; results are quoted in DESIGN.md §7."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch  # noqa: E402

import maray_amd as M  # noqa: E402
import scenes  # noqa: E402
from marayb import encode  # noqa: E402

out = {}


def run(name, data, w, h, textures=None, backend=M.BACKEND_JIT, reps=10):
    tape = M.Scene(data).lower()
    ctx = M.Context(tape, textures=textures, backend=backend)
    d8 = torch.empty((h, w, 3), dtype=torch.uint8, device='cuda')
    d64 = torch.empty((h, w, 3), dtype=torch.float64, device='cuda')
    r = {'pix_ops': tape.info['n_pix_ops']}
    ms8 = ctx.time_rows(w, h, 0, h, d_rgb8=d8.data_ptr(), reps=reps)
    ms64 = ctx.time_rows(w, h, 0, h, d_rgb64=d64.data_ptr(), reps=reps)
    px = w * h
    r['rgb8'] = {'ms': ms8, 'mpx_s': px / ms8 / 1e3, 'store_gb_s': px * 3 / ms8 / 1e6}
    r['rgb64'] = {'ms': ms64, 'mpx_s': px / ms64 / 1e3, 'store_gb_s': px * 24 / ms64 / 1e6}
    ctx.close()
    out[name] = r


run('config2 radial gradient 1024^2', encode((1024, 1024), scenes.radial_gradient()), 1024, 1024)
run('config2 radial gradient 8192^2', encode((8192, 8192), scenes.radial_gradient()), 8192, 8192)
run('config3b all-ops 4096^2', encode((4096, 4096), scenes.all_ops(4096, 4096)), 4096, 4096)
run('config5 textured 4096^2', encode((4096, 4096), scenes.textured(4096)), 4096, 4096, textures=scenes.textures(1))
run('config2 radial gradient 8192^2 (interpreter, LDS tape)', encode((8192, 8192), scenes.radial_gradient()), 8192, 8192,
    backend=M.BACKEND_TAPE, reps=3)
print(json.dumps(out, indent=1))
