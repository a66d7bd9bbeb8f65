#!/bin/bash
# after the two fixes of the specialised kernel's texture lookups: the GPU suite, then the sweep step that had stopped at seed 6462
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
t0=$(date +%s)
timeout -k 10 600 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc=$rc in $(( $(date +%s) - t0 )) s"; tail -12 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
{ echo "## tools/gpu_fuzz.py 6000 6600 (jit + tape-smem), after the fix"; timeout -k 10 500 python tools/gpu_fuzz.py 6000 6600 2>&1 | tail -2
  echo "## tools/gpu_fuzz.py 8100 8200 (jit + tape-smem; seed 8157)"; timeout -k 10 200 python tools/gpu_fuzz.py 8100 8200 2>&1 | tail -2; } > gpurun_out/r4_fuzz_sweep_3.txt 2>&1
cat gpurun_out/r4_fuzz_sweep_3.txt
