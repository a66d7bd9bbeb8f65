#!/bin/bash
# tools/ab_ylds.sh — A/B of MARAY_JIT_YLDS (y values from LDS vs scalar loads), three alternating runs each
root=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do for v in 1 0; do
  echo -n "YLDS=$v: "
  MARAY_JIT_YLDS=$v timeout -k 10 200 python "$root/bench.py" --cpu-seconds 0 --steps 50 --warmup 10 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step %.4f ms kernel %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
