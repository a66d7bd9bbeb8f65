#!/bin/bash
# tools/gpu_round4_b.sh — texel-once emitter + façade first call: parity of what changed, then the figures.
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "textured or textures or thousand or gen_to_image or host_rasters or random_scenes or remembers" > gpurun_out/gpu_tests_b.log 2>&1; rc=$?
tail -8 gpurun_out/gpu_tests_b.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_configs.py > gpurun_out/r4_other_configs.json 2> gpurun_out/r4_other_configs.err && python -c "
import json; j=json.load(open('gpurun_out/r4_other_configs.json'))
for k,v in j.items(): print(k, v['pix_ops'], round(v['rgb8']['ms'],4), round(v['rgb64']['ms'],4))" &&
MARAY_JIT_TEXEL_ONCE=0 timeout -k 10 300 python tools/bench_configs.py > gpurun_out/r4_other_configs_texel_per_app.json 2>/dev/null && python -c "
import json; j=json.load(open('gpurun_out/r4_other_configs_texel_per_app.json'))
for k,v in j.items(): print('per-app', k, v['pix_ops'], round(v['rgb8']['ms'],4), round(v['rgb64']['ms'],4))" &&
timeout -k 10 300 python tools/exp_first_call.py > gpurun_out/r4_first_call.txt 2>&1; tail -60 gpurun_out/r4_first_call.txt
