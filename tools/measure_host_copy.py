"""PCIe-inclusive rate of the host-pointer entry point (maray_hip_render_rows) on chess @ 4096^2.
Not the bench `value` (that one keeps outputs resident in HBM); recorded in DESIGN.md §7."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import maray_amd as M  # noqa: E402

s = M.Scene(open(os.path.join(ROOT, 'tests', 'golden', 'chess.maray'), 'rb').read())
s.rescale(4, 4)
tape = s.lower()
out = {}
for name, b in (('jit', M.BACKEND_JIT),):
    t0 = time.perf_counter()
    ctx = M.Context(tape, backend=b)
    out[name + '_ctx_create_s'] = time.perf_counter() - t0
    ctx.render_rows(4096, 4096, 0, 4096, want_f64=False)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.render_rows(4096, 4096, 0, 4096, want_f64=False)
        ts.append(time.perf_counter() - t0)
    out[name + '_rgb8_host_ms'] = min(ts) * 1e3
    out[name + '_rgb8_host_mpx_s'] = 4096 * 4096 / min(ts) / 1e6
    t0 = time.perf_counter()
    ctx.render_rows(4096, 4096, 0, 1024, want_u8=False, want_f64=True)
    out[name + '_rgb64_host_1024rows_ms'] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    img = M.gen_to_image(s, backend=b)
    out[name + '_gen_to_image_s (lower + hiprtc + render + copy)'] = time.perf_counter() - t0
    ctx.close()
print(json.dumps(out, indent=1))
