#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=/tmp/mc
for v in "X=1" "MARAY_BENCH_NULL_STREAM=1" "X=1" "MARAY_BENCH_NULL_STREAM=1"; do
env $v timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-cold --no-e2e 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['bit_exact_vs_golden'])"
done
