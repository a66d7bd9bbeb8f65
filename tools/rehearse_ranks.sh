#!/bin/bash
# Rehearse the N>1 sharding of bench.py on ONE GPU: rank r of N without a process group.
for nw in "2 0" "2 1" "8 0" "8 7"; do set -- $nw
  echo -n "world=$1 rank=$2: "
  MARAY_BENCH_FAKE_WORLD=$1 MARAY_BENCH_FAKE_RANK=$2 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.0f Mpx/s (as if all %d ranks ran like this one) ms/step %.3f parity %s | %s' % (d['value'], d['n_gpus'], d['ms_per_step'], d['config']['bit_exact_vs_golden'], d['config']['workload'][:60]))"
done
