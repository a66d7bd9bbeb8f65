"""Generate tests/golden/chess_1024.json from the CPU oracle.

Run only after tests/test_oracle_known_answers.py and test_oracle_var_fixer.py
pass (the oracle is then pinned to the reference's known answers).  Records the
SHA-256 of the RGB8 raster of data/chess.maray at its stored 1024x1024, the
white-pixel count, the mismatch set against images/chess.png, and f64 probes.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle_ffi import Scene  # noqa: E402

G = os.path.join(ROOT, 'tests', 'golden')
data = open(os.path.join(G, 'chess.maray'), 'rb').read()
png = np.asarray(Image.open(os.path.join(G, 'chess.png')).convert('RGB'))
s = Scene(data)
t = time.time()
rgb8, rgb64 = s.render_rows(1024, 1024, 0, 1024)
dt = time.time() - t
diff = np.any(rgb8 != png, axis=2)
ys, xs = np.nonzero(diff)
probes = [(282, 512), (283, 512), (290, 512), (295, 512), (100, 100), (512, 700), (300, 600), (97, 512), (850, 704),
          (0, 0), (1023, 1023), (512, 511), (512, 513), (640, 703), (640, 705)]
out = {
    'source': 'oracle/maray_oracle.c on tests/golden/chess.maray (copy of reference data/chess.maray)',
    'size': [1024, 1024],
    'rgb8_sha256': hashlib.sha256(rgb8.tobytes()).hexdigest(),
    'white_pixels': int((rgb8[:, :, 0] == 255).sum()),
    'png_white_pixels': int((png[:, :, 0] == 255).sum()),
    'png_mismatch_pixels': int(diff.sum()),
    'png_mismatch_rows': {str(int(r)): int((ys == r).sum()) for r in sorted(set(ys.tolist()))},
    'png_mismatch_x_range': [int(xs.min()), int(xs.max())] if len(xs) else None,
    'probes_xy_rgb64': [[x, y, [float(v) for v in rgb64[y, x]]] for x, y in probes],
    'row_sha256': {str(r): hashlib.sha256(rgb8[r].tobytes()).hexdigest() for r in (0, 100, 511, 512, 513, 600, 704, 1023)},
}
with open(os.path.join(G, 'chess_1024.json'), 'w') as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'probes_xy_rgb64'}, indent=1))
print('oracle render: %.1f s, %d threads' % (dt, os.cpu_count()))
