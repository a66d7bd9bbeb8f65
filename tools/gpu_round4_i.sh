#!/bin/bash
# launch order: empty groups of rows dealt among the busy ones (MARAY_ORDER_INTERLEAVE=1), A/B in one process
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 500 python tools/exp_pixels.py base interleave:MARAY_ORDER_INTERLEAVE=1 base interleave:MARAY_ORDER_INTERLEAVE=1 interleave_t4:MARAY_ORDER_INTERLEAVE=1,MARAY_JIT_TILES=4 interleave_t1:MARAY_ORDER_INTERLEAVE=1,MARAY_JIT_TILES=1 interleave_rows2:MARAY_ORDER_INTERLEAVE=1,MARAY_JIT_ROWS2=1,MARAY_JIT_TILES=2 > gpurun_out/r4_interleave_ab.jsonl 2> gpurun_out/r4_interleave_ab.err; cat gpurun_out/r4_interleave_ab.jsonl | cut -c1-300; tail -3 gpurun_out/r4_interleave_ab.err
