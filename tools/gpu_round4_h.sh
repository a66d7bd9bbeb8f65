#!/bin/bash
# texels through a buffer resource: parity, config 5 timing, textured crop counters
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "texel or textured or textures or thousand or random_scenes or torch_first" > gpurun_out/gpu_tests_h.log 2>&1; rc=$?
tail -8 gpurun_out/gpu_tests_h.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_configs.py > gpurun_out/other_configs_r4.json 2> gpurun_out/other_configs_r4.err && python -c "
import json; j=json.load(open('gpurun_out/other_configs_r4.json'))
for k,v in j.items(): print(k, v['pix_ops'], round(v['rgb8']['ms'],4), round(v['rgb64']['ms'],4))" &&
MARAY_JIT_TEXEL_ONCE=0 timeout -k 10 300 python tools/bench_configs.py > gpurun_out/other_configs_r4_texel_per_app.json 2>/dev/null
timeout -k 10 200 python tools/bench_soup.py 1000 2>&1 | cut -c1-300
export MARAY_CACHE_DIR=/tmp/maray_cache
bash tools/pmc_crop.sh r4_textured textured x > gpurun_out/crop_r4_textured.log 2>&1 || tail -3 gpurun_out/crop_r4_textured.log
python -c "
import json; p=json.load(open('gpurun_out/pmc_r4_textured.json'))
k=[v for n,v in p['kernels'].items() if 'pixels' in n][0]
print({a:(round(b,3) if isinstance(b,float) else b) for a,b in k['derived'].items()}); print(p.get('kernel_stats'))"
