#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=/tmp/mc
EXP_FRAME_ONLY=1 python tools/exp_pixels.py "rb256:" "rb512:MARAY_JIT_ROW_BLOCK=512" "rb768:MARAY_JIT_ROW_BLOCK=768" "rb1024:MARAY_JIT_ROW_BLOCK=1024" "rb128:MARAY_JIT_ROW_BLOCK=128" "rb256 again:"
