#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=off
timeout -k 10 300 python tools/bench_soup.py 300
timeout -k 10 300 python tools/bench_soup.py 300 colours
timeout -k 10 600 python tools/bench_soup.py 1000
MARAY_JIT_PX=1 timeout -k 10 300 python tools/bench_soup.py 300
