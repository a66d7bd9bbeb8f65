#!/bin/bash
# tools/gpu_tests.sh — what the driver runs at round end, on a fresh box: build, GPU tests, smoke, bench.
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
t0=$(date +%s)
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc=$rc in $(( $(date +%s) - t0 )) s"; tail -25 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
