import json, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import maray_amd as M
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
s = M.Scene(open(os.path.join(root, 'tests', 'golden', 'chess.maray'), 'rb').read())
s.rescale(4, 4)
tape = s.lower()
M.device_count()
pin = M.PinnedRaster(4096, 4096)
t0 = time.perf_counter()
ctx = M.Context(tape, backend=int(sys.argv[1]))
t1 = time.perf_counter()
ctx.render_rows_into(4096, 4096, 0, 4096, pin.array)
t2 = time.perf_counter()
ctx.render_rows_into(4096, 4096, 0, 4096, pin.array)
t3 = time.perf_counter()
print(json.dumps({'ctx_ms': (t1 - t0) * 1e3, 'frame_ms': (t2 - t1) * 1e3, 'frame2_ms': (t3 - t2) * 1e3}))
