#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_workers or host_rasters or progress or gen_to_image or knobs or chess_4096" > gpurun_out/gpu_tests26.log 2>&1; tail -6 gpurun_out/gpu_tests26.log
export MARAY_CACHE_DIR=/tmp/mc
EXP_FRAME_ONLY=1 timeout -k 10 600 python tools/exp_pixels.py "default:" "noorder:MARAY_JIT_NO_ORDER=1" "default again:"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-cold 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['traffic_profile'], d['end_to_end']['value'], d['end_to_end']['ms_per_frame'], d['end_to_end'].get('pageable'), d['end_to_end'].get('gen_to_image_pinned_ms'))"
