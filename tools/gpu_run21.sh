#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=/tmp/maray_cli_cache
rm -rf $MARAY_CACHE_DIR
python - <<'PY'
import subprocess, time, hashlib, json, numpy as np
from PIL import Image
g = json.load(open('tests/golden/chess_1024.json'))
for b in ('auto', 'auto', 'jit', 'jit', 'auto'):
    t = time.perf_counter()
    r = subprocess.run(['maray_amd/maray', '-c', '8', '--backend', b, '-i', 'tests/golden/chess.maray', '-o', '/tmp/chess_cli.png'], capture_output=True, text=True)
    dt = time.perf_counter() - t
    ok = r.returncode == 0 and hashlib.sha256(np.asarray(Image.open('/tmp/chess_cli.png').convert('RGB')).tobytes()).hexdigest() == g['rgb8_sha256']
    print('maray --backend %s: %.2f s wall, rc %d, raster ok %s' % (b, dt, r.returncode, ok), flush=True)
PY
