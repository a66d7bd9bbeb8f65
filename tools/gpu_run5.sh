#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
export MARAY_CACHE_DIR=/tmp/maray_cache
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python tools/exp_pixels.py "coop_t8:" "coop_t4:MARAY_JIT_TILES=4" "coop_t16:MARAY_JIT_TILES=16" "coop_t8_noorder:MARAY_JIT_NO_ORDER=1" "coop_t8_w8:MARAY_JIT_WAVES=8" > gpurun_out/exp5.jsonl 2> gpurun_out/exp5.err; cat gpurun_out/exp5.jsonl; tail -3 gpurun_out/exp5.err
export EXP_FRAME_ONLY=1
timeout -k 10 600 python tools/exp_pixels.py "rows_c128:MARAY_JIT_ROW_CHUNK_OPS=128" "rows_c64:MARAY_JIT_ROW_CHUNK_OPS=64" "rows_c32:MARAY_JIT_ROW_CHUNK_OPS=32" "rows_c16:MARAY_JIT_ROW_CHUNK_OPS=16" > gpurun_out/exp5b.jsonl 2> gpurun_out/exp5b.err; cat gpurun_out/exp5b.jsonl; tail -3 gpurun_out/exp5b.err
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "chess_4096 or corner or huge or all_ops or textured_scene or radial or ragged or boolean_that or guarded_shapes or libm_sweep" > gpurun_out/gpu_tests5.log 2>&1; tail -8 gpurun_out/gpu_tests5.log
