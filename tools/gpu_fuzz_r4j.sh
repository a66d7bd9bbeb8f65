#!/bin/bash
# the round's last sweep: 1,200 + 400 + 400 random scenes at -O2 (three geometries), in steps short enough to keep the log moving
# -> gpurun_out/r4_fuzz_sweep_10.txt
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
: > gpurun_out/r4_fuzz_sweep_10.txt
for a in "100000 100400" "100400 100800" "100800 101200" "102000 102400 600 96" "103000 103400 1000 37"; do
  { echo "## tools/gpu_fuzz.py $a"; timeout -k 10 300 python tools/gpu_fuzz.py $a 2>&1 | tail -1; } >> gpurun_out/r4_fuzz_sweep_10.txt 2>&1
  tail -1 gpurun_out/r4_fuzz_sweep_10.txt
done
