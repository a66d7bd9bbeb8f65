#!/bin/bash
# the sign of a bounded Step(Sin) from the half period its argument lies in (MARAY_JIT_QUICK_SIN): parity on the chess and
# fuzz tests, then frame / board / sky crops with the knob off and on, each in a process of its own
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "chess or golden or variants or soup or knob" > gpurun_out/gpu_tests_n.log 2>&1; rc=$?
tail -4 gpurun_out/gpu_tests_n.log
[ $rc -eq 0 ] || exit $rc
for v in 0 1 0 1; do
  for crop in frame board sky; do
    MARAY_JIT_QUICK_SIN=$v timeout -k 10 200 python tools/run_crop.py chess $crop 20 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('QUICK_SIN=$v', j['crop'], j['pixel_kernel_us'])" || exit 1
  done
done
MARAY_JIT_QUICK_SIN=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('QUICK_SIN=0 bench', j['value'], j['ms_per_step'], j.get('ctx_ms'), j.get('long_loop'))"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('QUICK_SIN=1 bench', j['value'], j['ms_per_step'], j.get('ctx_ms'), j.get('long_loop'))"
