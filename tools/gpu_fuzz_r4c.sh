#!/bin/bash
# a third sweep: 1,400 more random scenes (two geometries) on the final tree -> gpurun_out/r4_fuzz_sweep_4.txt
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
{
echo "## tools/gpu_fuzz.py 10000 11000 (jit + tape-smem)"; timeout -k 10 600 python tools/gpu_fuzz.py 10000 11000 2>&1 | tail -3
echo "## tools/gpu_fuzz.py 12000 12400 600 96"; timeout -k 10 400 python tools/gpu_fuzz.py 12000 12400 600 96 2>&1 | tail -3
} > gpurun_out/r4_fuzz_sweep_4.txt 2>&1
cat gpurun_out/r4_fuzz_sweep_4.txt
