#!/bin/bash
# tools/ab_knob.sh VAR A B — alternating A/B of one environment knob on bench.py (three runs each)
root=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do for v in "$2" "$3"; do
  echo -n "$1=$v: "
  env "$1=$v" timeout -k 10 200 python "$root/bench.py" --cpu-seconds 0 --steps 50 --warmup 10 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step %.4f ms kernel %.4f ms parity %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['bit_exact_vs_golden']))"
done; done
