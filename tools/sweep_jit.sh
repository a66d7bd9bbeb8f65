#!/bin/bash
# Sweep the hiprtc back-end's occupancy target on the GPU box; one bench line per setting.
for w in 4 5 6 7 8; do
  echo -n "MARAY_JIT_WAVES=$w : "
  MARAY_JIT_WAVES=$w python bench.py --backend jit --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f Mpx/s step %.3f ms kernel %.3f ms parity %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['bit_exact_vs_golden']))"
done
