#!/bin/bash
# a wavefront's chain in front of its first store: rows the order lists without a guard bit do not fetch their words
# (MARAY_JIT_SKY_ROWS), the kernel's arguments in one batch of scalar loads (MARAY_JIT_ARGS_UPFRONT).  Parity, then crops.
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "chess or golden or variants or knob or soup or one_launch" > gpurun_out/gpu_tests_q.log 2>&1; rc=$?
tail -4 gpurun_out/gpu_tests_q.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  for crop in frame board sky; do
    MARAY_JIT_SKY_ROWS=$1 MARAY_JIT_ARGS_UPFRONT=$2 timeout -k 10 200 python tools/run_crop.py chess $crop 20 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SKY_ROWS=$1 ARGS_UPFRONT=$2', j['crop'], j['pixel_kernel_us'])" || exit 1
  done
done
done
for cfg in "0 0" "1 1"; do
  set -- $cfg
  MARAY_JIT_SKY_ROWS=$1 MARAY_JIT_ARGS_UPFRONT=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SKY_ROWS=$1 ARGS_UPFRONT=$2 bench', j['value'], j['ms_per_step'], j['long_loop']['value'], j['roofline']['frac'])"
done
