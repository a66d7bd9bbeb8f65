#!/bin/bash
# hiprtc at -O2 instead of -O3 (MARAY_JIT_OPT=-O2: the PIXEL kernel of chess builds a quarter faster and differs in a handful of
# instructions): parity, then step, crops and the cold build per setting
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
MARAY_JIT_OPT=-O2 timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "chess_4096 or golden or guarded or soup or textured or variants" > gpurun_out/gpu_tests_ab.log 2>&1; rc=$?
tail -2 gpurun_out/gpu_tests_ab.log
[ $rc -eq 0 ] || exit $rc
for v in -O3 -O2 -O3 -O2; do
  export MARAY_JIT_OPT=$v
  for crop in frame board sky; do
    timeout -k 10 200 python tools/run_crop.py chess $crop 20 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OPT=$v', j['crop'], j['pixel_kernel_us'])" || exit 1
  done
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-e2e 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['cold_first_render_ms']; print('OPT=$v bench', round(j['value']), j['ms_per_step'], round(j['long_loop']['value']), 'cold ctx_ms', round(c['cold_cache']['ctx_ms']))"
  timeout -k 10 200 python tools/bench_configs.py 2>/dev/null | python -c "
import json,sys; j=json.load(sys.stdin); print('OPT=$v configs', {k.split()[0]+k.split()[-1]: round(v['rgb8']['ms']*1e3,1) for k,v in j.items() if isinstance(v,dict) and 'rgb8' in v and 'interpreter' not in k})"
done
