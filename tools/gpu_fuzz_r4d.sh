#!/bin/bash
# random scenes at ragged geometries (rows that do not end on a tile, widths that break the stores' alignment) -> gpurun_out/r4_fuzz_sweep_5.txt
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
{
for a in "13000 13300 333 77" "13300 13600 1000 37" "13600 13800 4097 9" "13800 14000 255 200" "14000 14200 65 130"; do
  echo "## tools/gpu_fuzz.py $a"; timeout -k 10 300 python tools/gpu_fuzz.py $a 2>&1 | tail -2
done
} > gpurun_out/r4_fuzz_sweep_5.txt 2>&1
cat gpurun_out/r4_fuzz_sweep_5.txt
