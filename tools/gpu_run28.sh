#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=off
timeout -k 10 600 python tools/bench_soup.py 1000
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knobs" 2>&1 | tail -3
EXP_FRAME_ONLY=1 timeout -k 10 300 python tools/exp_pixels.py "default:" "gw many:MARAY_JIT_GW_MANY=1"
