#!/bin/bash
# block placement of the sky variant (MARAY_JIT_SKY_LIKELY) against the quick Step(Sin) and the two-row kernel, whose sky crops
# were slower than the default kernel's: frame / board / sky crops, a process per run
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for rep in 1 2; do
for cfg in "0 0 0" "0 1 0" "1 1 0" "0 0 1" "0 1 1" "1 1 1"; do
  set -- $cfg
  for crop in frame board sky; do
    MARAY_JIT_QUICK_SIN=$1 MARAY_JIT_SKY_LIKELY=$2 MARAY_JIT_ROWS2=$3 timeout -k 10 200 python tools/run_crop.py chess $crop 20 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('QUICK_SIN=$1 SKY_LIKELY=$2 ROWS2=$3', j['crop'], j['pixel_kernel_us'])" || exit 1
  done
done
done
