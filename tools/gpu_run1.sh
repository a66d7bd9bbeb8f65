#!/bin/bash
# GPU run 1: tests, bench, variants, host copy rates
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export MARAY_CACHE_DIR=/tmp/maray_cache
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
echo "== hostcopy"; python tools/exp_hostcopy.py > gpurun_out/hostcopy2.json 2>&1; tail -40 gpurun_out/hostcopy2.json
echo "== exp"; timeout -k 10 900 python tools/exp_pixels.py "px1:MARAY_JIT_PX=1" "px4_t1:MARAY_JIT_TILES=1" "px4_t2:MARAY_JIT_TILES=2" "px4_t4:MARAY_JIT_TILES=4" "px4_t8:MARAY_JIT_TILES=8" "px4_t2_noorder:MARAY_JIT_TILES=2,MARAY_JIT_NO_ORDER=1" "px4_t2_r16:MARAY_JIT_MIN_REGION=16" "px4_t2_O1:MARAY_JIT_OPT=-O1" > gpurun_out/exp1.jsonl 2> gpurun_out/exp1.err; cat gpurun_out/exp1.jsonl; tail -3 gpurun_out/exp1.err
echo "== smoke"; timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench"; timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; tail -c 3000 gpurun_out/bench1.json; tail -5 gpurun_out/bench1.err
