#!/bin/bash
# scenes named from their bytes (+ rescale factors) and lowered without a copy: the façade's GPU tests, then the first call inside bench.py
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "gen_to_image or cli or png or authored" > gpurun_out/gpu_tests_w.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_w.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
MARAY_TRACE_LOWER=1 timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_trace.json 2> gpurun_out/bench_trace.err || { tail -5 gpurun_out/bench_trace.err; exit 1; }
grep -E "maray lower: fix_color" gpurun_out/bench_trace.err | tail -1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_trace.json').read().strip().splitlines() if x.startswith('{')][-1]
b=json.loads(l); e=b['end_to_end']; print('value', round(b['value']), 'gen_to_image_pinned_ms', e['gen_to_image_pinned_ms'], 'second', e['gen_to_image_second_call_ms'], 'parity', e['gen_to_image_equals_device_raster'], b['roofline']['traffic_profile']['matches_this_build'])
PY
done
