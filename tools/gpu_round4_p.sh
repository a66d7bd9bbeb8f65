#!/bin/bash
# rows that may show shapes cut into shorter strips than the rows of sky (MARAY_JIT_BUSY_STRIPS=1, an experiment: the grid keeps its
# rectangle, the sky rows' surplus blocks exit at once): parity on chess, then frame / board / sky crops, a process per run
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
MARAY_JIT_BUSY_STRIPS=1 timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "chess_4096 or golden or knob" > gpurun_out/gpu_tests_p.log 2>&1; rc=$?
tail -4 gpurun_out/gpu_tests_p.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
for cfg in "0 0" "1 0" "1 4" "0 4"; do
  set -- $cfg
  for crop in frame board sky; do
    MARAY_JIT_BUSY_STRIPS=$1 MARAY_JIT_TILES=$2 timeout -k 10 200 python tools/run_crop.py chess $crop 20 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BUSY_STRIPS=$1 TILES=$2', j['crop'], j['pixel_kernel_us'])" || exit 1
  done
done
done
