#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=/tmp/mc
python tools/exp_launch_overhead.py
python tools/exp_launch_overhead.py
