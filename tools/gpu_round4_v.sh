#!/bin/bash
# where the first maray_gen_to_image call of bench.py spends its time (the lowering's and the context's traces on stderr)
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
MARAY_TRACE_LOWER=1 MARAY_TRACE_INIT=1 timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_trace.json 2> gpurun_out/bench_trace.err || { tail -5 gpurun_out/bench_trace.err; exit 1; }
grep -E "maray lower:|maray init:" gpurun_out/bench_trace.err | grep -v "interval rules\|guards over" | tail -60
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_trace.json').read().strip().splitlines() if x.startswith('{')][-1]
b=json.loads(l); print('gen_to_image_pinned_ms', b['end_to_end']['gen_to_image_pinned_ms'])
PY
