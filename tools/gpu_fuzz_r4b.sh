#!/bin/bash
# a second, wider parity sweep on round 4's final tree (after the lowering's tables were rebuilt) -> gpurun_out/r4_fuzz_sweep_2.txt
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
{
echo "## tools/gpu_fuzz.py 6000 6600 (jit + tape-smem)"; timeout -k 10 500 python tools/gpu_fuzz.py 6000 6600 2>&1 | tail -2
echo "## tools/gpu_fuzz.py 7000 7160 600 96"; timeout -k 10 300 python tools/gpu_fuzz.py 7000 7160 600 96 2>&1 | tail -2
echo "## tools/gpu_fuzz_soups.py 300 324 (70 polygons each, three mixing modes)"; timeout -k 10 300 python tools/gpu_fuzz_soups.py 300 324 2>&1 | tail -2
echo "## tools/gpu_fuzz_products.py 3000 3120 24"; timeout -k 10 400 python tools/gpu_fuzz_products.py 3000 3120 24 2>&1 | tail -2
echo "## tools/gpu_fuzz_curved.py 2000 2120 40"; timeout -k 10 400 python tools/gpu_fuzz_curved.py 2000 2120 40 2>&1 | tail -2
} > gpurun_out/r4_fuzz_sweep_2.txt 2>&1
cat gpurun_out/r4_fuzz_sweep_2.txt
