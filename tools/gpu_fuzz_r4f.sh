#!/bin/bash
# a last sweep of round 4: 1,000 + 500 more random scenes (default geometry and 600 x 96), 40 + 40 soups -> gpurun_out/r4_fuzz_sweep_6.txt
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
{
echo "## tools/gpu_fuzz.py 50000 51000"; timeout -k 10 600 python tools/gpu_fuzz.py 50000 51000 2>&1 | tail -2
echo "## tools/gpu_fuzz.py 52000 52500 600 96"; timeout -k 10 300 python tools/gpu_fuzz.py 52000 52500 600 96 2>&1 | tail -2
echo "## tools/gpu_fuzz_products.py 4000 4040 24"; timeout -k 10 200 python tools/gpu_fuzz_products.py 4000 4040 24 2>&1 | tail -2
echo "## tools/gpu_fuzz_curved.py 3000 3040 40"; timeout -k 10 200 python tools/gpu_fuzz_curved.py 3000 3040 40 2>&1 | tail -2
} > gpurun_out/r4_fuzz_sweep_6.txt 2>&1
cat gpurun_out/r4_fuzz_sweep_6.txt
