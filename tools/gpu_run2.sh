#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
export MARAY_CACHE_DIR=/tmp/maray_cache
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_JIT_PX=1
bash tools/pmc_crop.sh px1_board chess board 2>&1 | tail -100
bash tools/pmc_crop.sh px1_sky chess sky 2>&1 | tail -70
