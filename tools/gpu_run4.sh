#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
export MARAY_CACHE_DIR=/tmp/maray_cache
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python tools/exp_pixels.py "wave_t1:MARAY_JIT_TILES=1" "wave_t2:MARAY_JIT_TILES=2" "wave_t4:MARAY_JIT_TILES=4" "wave_t2_noorder:MARAY_JIT_TILES=2,MARAY_JIT_NO_ORDER=1" "wave_t2_w8:MARAY_JIT_WAVES=8" "wave_t2_w4:MARAY_JIT_WAVES=4" > gpurun_out/exp4.jsonl 2> gpurun_out/exp4.err; cat gpurun_out/exp4.jsonl; tail -3 gpurun_out/exp4.err
