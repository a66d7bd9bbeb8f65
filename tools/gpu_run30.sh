#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=/tmp/mc
timeout -k 10 300 python tools/bench_soup.py 1000 | cut -c150-400
timeout -k 10 300 python tools/bench_soup.py 300 | cut -c150-400
timeout -k 10 300 python tools/bench_soup.py 300 colours | cut -c150-400
timeout -k 10 600 python tools/exp_pixels.py "default:" "minreg0:MARAY_JIT_MIN_REGION=0" "minreg12:MARAY_JIT_MIN_REGION=12" "minreg40:MARAY_JIT_MIN_REGION=40"
