#!/bin/bash
# soups at -O2: 20 polygon, 100 product, 80 curved more -> gpurun_out/r4_fuzz_sweep_9.txt
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
{
echo "## tools/gpu_fuzz_soups.py 500 520"; timeout -k 10 300 python tools/gpu_fuzz_soups.py 500 520 2>&1 | tail -2
echo "## tools/gpu_fuzz_products.py 5000 5100 24"; timeout -k 10 300 python tools/gpu_fuzz_products.py 5000 5100 24 2>&1 | tail -2
echo "## tools/gpu_fuzz_curved.py 5000 5080 40"; timeout -k 10 400 python tools/gpu_fuzz_curved.py 5000 5080 40 2>&1 | tail -2
} > gpurun_out/r4_fuzz_sweep_9.txt 2>&1
cat gpurun_out/r4_fuzz_sweep_9.txt
