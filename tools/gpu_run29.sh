#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=/tmp/mc
for v in "MARAY_JIT_PX=1" "MARAY_JIT_DERIVED=0" "MARAY_JIT_YBOOL=0" "MARAY_JIT_MIN_REGION=0" "MARAY_JIT_NO_ORDER=1"; do
  echo "== $v"; env $v timeout -k 10 300 python tools/bench_soup.py 1000 | cut -c150-400
done
