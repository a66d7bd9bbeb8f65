#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=/tmp/mc
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "radial" 2>&1 | tail -3
python tools/bench_configs.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print('  %-60s rgb8 %8.3f ms %7.0f GB/s   rgb64 %8.3f ms %7.0f GB/s' % (k, v['rgb8']['ms'], v['rgb8']['store_gb_s'], v['rgb64']['ms'], v['rgb64']['store_gb_s']))
"
