#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=/tmp/mc
EXP_FRAME_ONLY=1 python tools/exp_pixels.py "default:" "rb128:MARAY_JIT_ROW_BLOCK=128" "default again:"
cd /tmp && export TMPDIR=/tmp
for v in "X=1" "MARAY_JIT_ROW_PART=1" "MARAY_JIT_ROW_PART=2"; do
  echo "== rows $v"
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rows_trace -- python3 $GRAFT_REPO_ROOT/tools/run_crop.py chess frame 10 > /dev/null 2>&1
  grep "maray_jit_[pr]" $GRAFT_REPO_ROOT/gpurun_out/rows_trace/*/*_kernel_stats.csv | cut -d, -f1-4
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/rows_trace
done
