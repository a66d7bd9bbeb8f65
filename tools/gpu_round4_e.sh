#!/bin/bash
# fused Step(x + k) compares: parity, then A/B in one process
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "step_of_a_sum or chess_4096 or variants or soups or thousand or chess_1024 or corner or random_scenes or inf_and_nan or all_ops or libm" > gpurun_out/gpu_tests_e.log 2>&1; rc=$?
tail -12 gpurun_out/gpu_tests_e.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/exp_pixels.py base nofuse:MARAY_JIT_FUSE_CMP=0 base nofuse:MARAY_JIT_FUSE_CMP=0 rows2t2:MARAY_JIT_ROWS2=1,MARAY_JIT_TILES=2 > gpurun_out/r4_fuse_ab.jsonl 2> gpurun_out/r4_fuse_ab.err; cat gpurun_out/r4_fuse_ab.jsonl | cut -c1-300
