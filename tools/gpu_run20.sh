#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
echo "== CLI, cold cache dir (AUTO -> interpreter), then jit forced (cold build), then jit again (warm cache)"
export MARAY_CACHE_DIR=/tmp/maray_cli_cache
rm -rf $MARAY_CACHE_DIR
for b in auto jit jit; do
  /usr/bin/time -f "$b: %e s wall" maray_amd/maray -c 8 --backend $b -i tests/golden/chess.maray -o /tmp/chess_$b.png 2>&1 | tail -1
done
python - <<'PY'
import hashlib, json, numpy as np
from PIL import Image
g = json.load(open('tests/golden/chess_1024.json'))
for b in ('auto', 'jit'):
    print(b, hashlib.sha256(np.asarray(Image.open('/tmp/chess_%s.png' % b).convert('RGB')).tobytes()).hexdigest() == g['rgb8_sha256'])
PY
echo "== rank rehearsals (weak) and strong mode"
bash tools/rehearse_ranks.sh 2>&1 | tail -4
python bench.py --scaling strong --steps 5 --warmup 2 --cpu-seconds 0 --no-cold 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('strong N=1: %.0f Mpx/s ms/step %.3f parity %s e2e %.0f Mpx/s | %s' % (d['value'], d['ms_per_step'], d['config']['bit_exact_vs_golden'], d['end_to_end']['value'], d['config']['workload'][:70]))"
MARAY_BENCH_FAKE_WORLD=8 MARAY_BENCH_FAKE_RANK=3 python bench.py --scaling strong --steps 5 --warmup 2 --cpu-seconds 0 --no-cold 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('strong N=8 rank 3 alone: %.0f Mpx/s ms/step %.3f e2e %.0f | tiles %d' % (d['value'], d['ms_per_step'], d['end_to_end']['value'], d['end_to_end']['tiles_per_rank']))"
echo "== fuzz"
export MARAY_CACHE_DIR=off
timeout -k 10 500 python tools/gpu_fuzz.py 2000 2150 2>&1 | tail -3
timeout -k 10 300 python tools/gpu_fuzz.py 3000 3060 600 20 2>&1 | tail -2
