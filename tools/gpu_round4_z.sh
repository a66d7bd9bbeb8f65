#!/bin/bash
# the ROW kernel's LDS: a chunk stages 16 / 8 / 4 y values per row (35 / 17 / 9 KB a block: 4 / 9 / 18 blocks a CU; the launch has ~1,250
# blocks, most of which need no LDS at all) -- parity at 8, then maray_jit_rows and the step per setting
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
MARAY_JIT_ROW_STAGE=8 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "chess or golden or knob or soup or one_launch or guarded or authored" > gpurun_out/gpu_tests_z.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_z.log
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
for v in 16 8 4 16 8 4; do
  rm -rf /tmp/tr_$v
  MARAY_JIT_ROW_STAGE=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python3 $GRAFT_REPO_ROOT/tools/run_crop.py chess frame 10 > /tmp/run_$v.json 2> /tmp/err_$v.txt || { tail -5 /tmp/err_$v.txt; exit 1; }
  f=$(find /tmp/tr_$v -name '*kernel_stats.csv' | head -1)
  echo "ROW_STAGE=$v $(grep -E 'maray_jit_rows' $f | cut -d, -f1-4 | tr '\n' ' ')"
done
cd $GRAFT_REPO_ROOT
for v in 16 8 4 16 8 4; do
  MARAY_JIT_ROW_STAGE=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-cold --no-e2e 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ROW_STAGE=$v bench', round(j['value']), j['ms_per_step'], round(j['long_loop']['value']), j['long_loop']['ms_per_step'])"
done
