#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
export MARAY_CACHE_DIR=/tmp/maray_cache
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python tools/exp_pixels.py "default:" "px1:MARAY_JIT_PX=1" "minreg8:MARAY_JIT_MIN_REGION=8" "minreg12:MARAY_JIT_MIN_REGION=12" "minreg16:MARAY_JIT_MIN_REGION=16" "minreg24:MARAY_JIT_MIN_REGION=24" "minreg40:MARAY_JIT_MIN_REGION=40" "tiles4:MARAY_JIT_TILES=4" "default again:" > gpurun_out/exp18.jsonl 2> gpurun_out/exp18.err; cat gpurun_out/exp18.jsonl; tail -3 gpurun_out/exp18.err
