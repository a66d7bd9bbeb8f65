#!/bin/bash
# one-off wider parity sweeps of round 4 on the final build -> gpurun_out/r4_fuzz_sweep.txt
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
{
echo "## tools/gpu_fuzz.py 4000 4400 (jit + tape-smem)"; timeout -k 10 400 python tools/gpu_fuzz.py 4000 4400 2>&1 | tail -2
echo "## tools/gpu_fuzz.py 5000 5120 600 96"; timeout -k 10 300 python tools/gpu_fuzz.py 5000 5120 600 96 2>&1 | tail -2
echo "## tools/gpu_fuzz_soups.py 200 216 (70 polygons each, three mixing modes)"; timeout -k 10 300 python tools/gpu_fuzz_soups.py 200 216 2>&1 | tail -2
echo "## tools/gpu_fuzz_products.py 2000 2096 24"; timeout -k 10 400 python tools/gpu_fuzz_products.py 2000 2096 24 2>&1 | tail -2
} > gpurun_out/r4_fuzz_sweep.txt 2>&1
cat gpurun_out/r4_fuzz_sweep.txt
