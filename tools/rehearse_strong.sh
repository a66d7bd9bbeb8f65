#!/bin/bash
# Rehearse `bench.py --scaling strong` (config 4, chess @16384^2 shared by N ranks) on ONE GPU: rank r of N without a process group.
cd ${GRAFT_REPO_ROOT:-.}
for nw in "1 0" "2 1" "4 0" "8 7"; do set -- $nw
  echo -n "world=$1 rank=$2: "
  MARAY_BENCH_FAKE_WORLD=$1 MARAY_BENCH_FAKE_RANK=$2 python bench.py --scaling strong --steps 10 --warmup 3 --cpu-seconds 0 --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms/step %.4f kernel_ms %.4f value %.0f (as if all ranks ran like this one) parity %s | %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['config']['bit_exact_vs_golden'], d['config']['workload'][:48]))"
done
