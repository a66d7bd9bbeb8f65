#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
export MARAY_CACHE_DIR=/tmp/maray_cache
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for v in "MARAY_JIT_PX=1" "X=1" "MARAY_JIT_WIDE=0" "MARAY_JIT_TILES=4" "MARAY_JIT_TILES=8" "MARAY_JIT_WIDE=1"; do
  echo "== $v"
  env $v python tools/bench_configs.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print('  %-60s rgb8 %8.3f ms %7.0f GB/s   rgb64 %8.3f ms %7.0f GB/s' % (k, v['rgb8']['ms'], v['rgb8']['store_gb_s'], v['rgb64']['ms'], v['rgb64']['store_gb_s']))
"
done
