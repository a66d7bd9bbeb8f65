#!/bin/bash
# tools/gpu_round4_c.sh — two rows per wavefront (MARAY_JIT_ROWS2=1): parity, then A/B in one process.
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
MARAY_JIT_ROWS2=1 timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "chess_4096 or variants or soups or thousand or chess_1024 or hoisting or row_blocks or ragged or more_rows or textured or inf_and_nan or random_scenes or 16384_in_one" > gpurun_out/gpu_tests_c.log 2>&1; rc=$?
tail -12 gpurun_out/gpu_tests_c.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/exp_pixels.py base rows2:MARAY_JIT_ROWS2=1 rows2w8:MARAY_JIT_ROWS2=1,MARAY_JIT_ROWS2_WAVES=8 rows2t1:MARAY_JIT_ROWS2=1,MARAY_JIT_TILES=1 rows2t2:MARAY_JIT_ROWS2=1,MARAY_JIT_TILES=2 base > gpurun_out/r4_rows2_ab.jsonl 2> gpurun_out/r4_rows2_ab.err; cat gpurun_out/r4_rows2_ab.jsonl | cut -c1-400
