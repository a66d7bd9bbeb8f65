#!/bin/bash
# Step(Sin) constants in vector registers: parity, then A/B in one process; tiles 3
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "chess_4096 or variants or soups or thousand or chess_1024 or libm or sin_of_huge or random_scenes" > gpurun_out/gpu_tests_g.log 2>&1; rc=$?
tail -6 gpurun_out/gpu_tests_g.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/exp_pixels.py base sin_from_table:MARAY_JIT_SIN_REGS=0 base sin_from_table:MARAY_JIT_SIN_REGS=0 tiles3:MARAY_JIT_TILES=3 > gpurun_out/r4_sinregs_ab.jsonl 2> gpurun_out/r4_sinregs_ab.err; cat gpurun_out/r4_sinregs_ab.jsonl | cut -c1-300
