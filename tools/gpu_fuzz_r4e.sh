#!/bin/bash
# random scenes through the other entry points of a context (tools/gpu_fuzz_paths.py) -> gpurun_out/r4_fuzz_paths.txt
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
{
for a in "15000 15150 600 96" "15150 15250 333 77" "15250 15350 1000 64"; do
  echo "## tools/gpu_fuzz_paths.py $a"; timeout -k 10 350 python tools/gpu_fuzz_paths.py $a 2>&1 | tail -4
done
} > gpurun_out/r4_fuzz_paths.txt 2>&1
cat gpurun_out/r4_fuzz_paths.txt
