#!/bin/bash
# strips drawn from a queue (MARAY_JIT_QUEUE, default on): the whole GPU suite, then A/B in one process
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 400 python tools/exp_pixels.py base grid:MARAY_JIT_QUEUE=0 base grid:MARAY_JIT_QUEUE=0 t1:MARAY_JIT_TILES=1 t4:MARAY_JIT_TILES=4 > gpurun_out/r4_queue_ab.jsonl 2> gpurun_out/r4_queue_ab.err; cat gpurun_out/r4_queue_ab.jsonl | cut -c1-300; tail -3 gpurun_out/r4_queue_ab.err
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/gpu_tests_j.log 2>&1; rc=$?
tail -14 gpurun_out/gpu_tests_j.log
exit $rc
