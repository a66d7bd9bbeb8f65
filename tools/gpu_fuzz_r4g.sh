#!/bin/bash
# the LDS-tape interpreter too (MARAY_FUZZ_ALL_BACKENDS=1): 400 + 200 random scenes on all three evaluators -> gpurun_out/r4_fuzz_sweep_7.txt
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
{
echo "## MARAY_FUZZ_ALL_BACKENDS=1 tools/gpu_fuzz.py 70000 70400"; MARAY_FUZZ_ALL_BACKENDS=1 timeout -k 10 400 python tools/gpu_fuzz.py 70000 70400 2>&1 | tail -2
echo "## MARAY_FUZZ_ALL_BACKENDS=1 tools/gpu_fuzz.py 71000 71200 333 77"; MARAY_FUZZ_ALL_BACKENDS=1 timeout -k 10 300 python tools/gpu_fuzz.py 71000 71200 333 77 2>&1 | tail -2
} > gpurun_out/r4_fuzz_sweep_7.txt 2>&1
cat gpurun_out/r4_fuzz_sweep_7.txt
