#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 500 python tools/exp_pixels.py base rows2_rpw1:MARAY_JIT_ROWS2=1,MARAY_RPW1=1 rows2_rpw1_t2:MARAY_JIT_ROWS2=1,MARAY_RPW1=1,MARAY_JIT_TILES=2 rows2t2:MARAY_JIT_ROWS2=1,MARAY_JIT_TILES=2 rows2t2w8:MARAY_JIT_ROWS2=1,MARAY_JIT_TILES=2,MARAY_JIT_ROWS2_WAVES=8 rows2t2h64:MARAY_JIT_ROWS2=1,MARAY_JIT_TILES=2,MARAY_JIT_GUARD_H=64 base_t1:MARAY_JIT_TILES=1 > gpurun_out/r4_rows2_ab2.jsonl 2> gpurun_out/r4_rows2_ab2.err; cat gpurun_out/r4_rows2_ab2.jsonl | cut -c1-300
