#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
export MARAY_CACHE_DIR=/tmp/maray_cache
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python tools/exp_pixels.py "px1:MARAY_JIT_PX=1" "wave_t1:MARAY_JIT_TILES=1" "wave_t2:MARAY_JIT_TILES=2" "wave_t4:MARAY_JIT_TILES=4" "wave_t2_noorder:MARAY_JIT_TILES=2,MARAY_JIT_NO_ORDER=1" "wave_t4_noorder:MARAY_JIT_TILES=4,MARAY_JIT_NO_ORDER=1" "wave_t2_wide:MARAY_JIT_WIDE=1" > gpurun_out/exp3.jsonl 2> gpurun_out/exp3.err; cat gpurun_out/exp3.jsonl; tail -3 gpurun_out/exp3.err
cd /tmp && export TMPDIR=/tmp
for v in "" "MARAY_JIT_ROW_PART=1" "MARAY_JIT_ROW_BLOCK=64" "MARAY_JIT_ROW_BLOCK=128"; do
  echo "== rows $v"
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rows_trace -- python3 $GRAFT_REPO_ROOT/tools/run_crop.py chess frame 10 > /dev/null 2>&1
  cat $GRAFT_REPO_ROOT/gpurun_out/rows_trace/*/*_kernel_stats.csv | cut -d, -f1-4 | head -5
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/rows_trace
done
