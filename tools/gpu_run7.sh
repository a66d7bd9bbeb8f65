#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
export MARAY_CACHE_DIR=/tmp/maray_cache
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python tools/exp_pixels.py "px1:MARAY_JIT_PX=1" "px1_w6:MARAY_JIT_PX=1,MARAY_JIT_WAVES=6" "coop_t8:" "coop_t16:MARAY_JIT_TILES=16" "coop_t16_noyb:MARAY_JIT_TILES=16,MARAY_JIT_YBOOL=0" "wave_t2:MARAY_JIT_LAYOUT=wave" "wave_t4:MARAY_JIT_LAYOUT=wave,MARAY_JIT_TILES=4" > gpurun_out/exp7.jsonl 2> gpurun_out/exp7.err; cat gpurun_out/exp7.jsonl; tail -3 gpurun_out/exp7.err
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "chess_4096 or corner or huge or all_ops or textured_scene or radial or ragged or boolean_that or guarded_shapes or libm_sweep or spill" > gpurun_out/gpu_tests7.log 2>&1; tail -8 gpurun_out/gpu_tests7.log
