#!/bin/bash
# the ROW kernel compiled at -O1 (MARAY_JIT_ROW_OPT; its time has followed nothing its code does): cold build, step, parity on chess
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
MARAY_JIT_ROW_OPT=-O1 timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "chess_4096 or golden or guarded or soup" > gpurun_out/gpu_tests_aa.log 2>&1; rc=$?
tail -2 gpurun_out/gpu_tests_aa.log
[ $rc -eq 0 ] || exit $rc
for v in "" -O1 "" -O1; do
  if [ -z "$v" ]; then unset MARAY_JIT_ROW_OPT; else export MARAY_JIT_ROW_OPT=$v; fi
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-e2e 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['cold_first_render_ms']; print('ROW_OPT=${v:-default}', round(j['value']), j['ms_per_step'], round(j['long_loop']['value']), 'cold ctx_ms', round(c['cold_cache']['ctx_ms']), 'warm', round(c['warm_cache']['ctx_ms']))"
done
