#!/bin/bash
# after hiprtc went to -O2: 600 + 300 random scenes and 40 soups more -> gpurun_out/r4_fuzz_sweep_8.txt
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
{
echo "## tools/gpu_fuzz.py 80000 80600"; timeout -k 10 400 python tools/gpu_fuzz.py 80000 80600 2>&1 | tail -2
echo "## tools/gpu_fuzz.py 81000 81300 333 77"; timeout -k 10 300 python tools/gpu_fuzz.py 81000 81300 333 77 2>&1 | tail -2
echo "## tools/gpu_fuzz_soups.py 400 412"; timeout -k 10 200 python tools/gpu_fuzz_soups.py 400 412 2>&1 | tail -2
echo "## tools/gpu_fuzz_curved.py 4000 4030 40"; timeout -k 10 200 python tools/gpu_fuzz_curved.py 4000 4030 40 2>&1 | tail -2
} > gpurun_out/r4_fuzz_sweep_8.txt 2>&1
cat gpurun_out/r4_fuzz_sweep_8.txt
