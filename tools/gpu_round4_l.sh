#!/bin/bash
# textured programs four pixels per lane (MARAY_JIT_WIDE_APP=1): parity on the textured tests, config 5 timing
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
MARAY_JIT_WIDE_APP=1 timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "texel or textured or textures" > gpurun_out/gpu_tests_l.log 2>&1; rc=$?
tail -4 gpurun_out/gpu_tests_l.log
[ $rc -eq 0 ] || exit $rc
for v in 0 1 0 1; do MARAY_JIT_WIDE_APP=$v timeout -k 10 200 python tools/bench_configs.py 2>/dev/null | python -c "
import json,sys; j=json.load(sys.stdin); v=j['config5 textured 4096^2']; print('WIDE_APP=$v config5', round(v['rgb8']['ms']*1e3,1), round(v['rgb64']['ms']*1e3,1))"; done
