#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
export MARAY_CACHE_DIR=/tmp/maray_cache
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export EXP_FRAME_ONLY=1
timeout -k 10 900 python tools/exp_pixels.py "default:" "noderived:MARAY_JIT_DERIVED=0" "noov:MARAY_JIT_ROW_OVERLAP=0" "noov_noderived:MARAY_JIT_ROW_OVERLAP=0,MARAY_JIT_DERIVED=0" > gpurun_out/exp15.jsonl 2> gpurun_out/exp15.err; cat gpurun_out/exp15.jsonl; tail -3 gpurun_out/exp15.err
unset EXP_FRAME_ONLY
timeout -k 10 900 python tools/exp_pixels.py "default_crops:" > gpurun_out/exp15b.jsonl 2>> gpurun_out/exp15.err; cat gpurun_out/exp15b.jsonl
cd /tmp && export TMPDIR=/tmp
for v in "X=1" "MARAY_JIT_ROW_PART=2" "MARAY_JIT_ROW_PART=1" "MARAY_JIT_ROW_BLOCK=128" "MARAY_JIT_ROW_BLOCK=64"; do
  echo "== rows $v"
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rows_trace -- python3 $GRAFT_REPO_ROOT/tools/run_crop.py chess frame 10 > /dev/null 2>&1
  grep "maray_jit" $GRAFT_REPO_ROOT/gpurun_out/rows_trace/*/*_kernel_stats.csv | cut -d, -f1-4
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/rows_trace
done
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu -k "chess_4096 or corner or huge or all_ops or textured_scene or radial or ragged or boolean_that or guarded_shapes or libm_sweep or spill or hoisting or knobs or random_scenes or soups" > gpurun_out/gpu_tests15.log 2>&1; tail -8 gpurun_out/gpu_tests15.log
