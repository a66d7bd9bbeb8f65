#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
export MARAY_CACHE_DIR=/tmp/maray_cache
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export EXP_FRAME_ONLY=1
timeout -k 10 900 python tools/exp_pixels.py "wave_t2:MARAY_JIT_LAYOUT=wave" "wave_t2_noov:MARAY_JIT_LAYOUT=wave,MARAY_JIT_ROW_OVERLAP=0" "coop_t16:MARAY_JIT_TILES=16" "px1:MARAY_JIT_PX=1" "px1_noov:MARAY_JIT_PX=1,MARAY_JIT_ROW_OVERLAP=0" > gpurun_out/exp12.jsonl 2> gpurun_out/exp12.err; cat gpurun_out/exp12.jsonl; tail -3 gpurun_out/exp12.err
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "chess_4096 or corner or huge or all_ops or textured_scene or radial or ragged or boolean_that or guarded_shapes or libm_sweep or spill or hoisting or host_rasters or row_blocks" > gpurun_out/gpu_tests12.log 2>&1; tail -8 gpurun_out/gpu_tests12.log
MARAY_JIT_LAYOUT=wave timeout -k 10 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-cold > gpurun_out/bench12.json 2> gpurun_out/bench12.err; python - <<'PY'
import json
for l in open('gpurun_out/bench12.json'):
    if l.startswith('{'):
        d=json.loads(l); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel_ms'], d['end_to_end'])
PY
tail -3 gpurun_out/bench12.err
