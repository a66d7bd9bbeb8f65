#!/bin/bash
# EXPERIMENT (timing only, wrong guard bits): the ROW kernel's 17 guard jobs sharing the code of the first 17 / 4 / 1 of them -- the same
# work per wavefront from less code: is the kernel's time instruction delivery?
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
for v in all g c all g c; do
  rm -rf /tmp/tr_$v
  if [ $v = all ]; then unset MARAY_JIT_EXP_ROW_ONLY; else export MARAY_JIT_EXP_ROW_ONLY=$v; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python3 $GRAFT_REPO_ROOT/tools/run_crop.py chess frame 10 > /tmp/run_$v.json 2> /tmp/err_$v.txt || { tail -5 /tmp/err_$v.txt; exit 1; }
  f=$(find /tmp/tr_$v -name '*kernel_stats.csv' | head -1)
  echo "ROW_ONLY=$v $(grep -E 'maray_jit_rows' $f | cut -d, -f1-4 | tr '\n' ' ')"
done
