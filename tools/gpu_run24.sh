#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=/tmp/mc
timeout -k 10 900 python tools/exp_pixels.py "persist7:" "nopersist:MARAY_JIT_PERSIST=0" "persist5:MARAY_JIT_BLOCKS_PER_CU=5" "persist6:MARAY_JIT_BLOCKS_PER_CU=6" "persist8:MARAY_JIT_BLOCKS_PER_CU=8" "persist12:MARAY_JIT_BLOCKS_PER_CU=12" "persist7_t1:MARAY_JIT_TILES=1" "persist7_t4:MARAY_JIT_TILES=4" "persist7_noorder:MARAY_JIT_NO_ORDER=1" "persist7 again:" > gpurun_out/exp24.jsonl 2> gpurun_out/exp24.err; cat gpurun_out/exp24.jsonl; tail -3 gpurun_out/exp24.err
