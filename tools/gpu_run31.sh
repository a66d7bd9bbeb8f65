#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=/tmp/mc
python tools/exp_launch_overhead.py
EXP_FRAME_ONLY=1 python tools/exp_pixels.py "default:" "overlap:MARAY_JIT_ROW_OVERLAP=1" "px1:MARAY_JIT_PX=1"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-cold --no-e2e 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
