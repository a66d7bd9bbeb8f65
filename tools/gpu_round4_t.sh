#!/bin/bash
# EXPERIMENT (timing only, the guard bits it writes are wrong): the ROW kernel's guard jobs of 8 / 4 / 2 / 1 guards each --
# is the kernel's time the vector work of its most loaded CU?  rocprofv3 kernel trace of a few frames per setting.
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
for bits in 8 4 2 1 8 4 2; do
  rm -rf /tmp/tr_$bits
  MARAY_JIT_EXP_JOB_BITS=$bits timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$bits -- python3 $GRAFT_REPO_ROOT/tools/run_crop.py chess frame 10 > /tmp/run_$bits.json 2> /tmp/err_$bits.txt || { tail -5 /tmp/err_$bits.txt; exit 1; }
  f=$(find /tmp/tr_$bits -name '*kernel_stats.csv' | head -1)
  echo "JOB_BITS=$bits $(grep -E 'maray_jit_rows|maray_jit_pixels' $f | cut -d, -f1-4 | tr '\n' ' ')"
done
