#!/bin/bash
# Sweep the hiprtc back-end's tuning knobs on the GPU box; one bench line per setting.
for w in 0 4 6 8; do for y in 0 1; do
  echo -n "MARAY_JIT_WAVES=$w MARAY_JIT_YLDS=$y : "
  MARAY_JIT_WAVES=$w MARAY_JIT_YLDS=$y python bench.py --backend jit --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f Mpx/s kernel %.3f ms parity %s' % (d['value'], d['roofline']['kernel_ms'], d['config']['bit_exact_vs_golden']))"
done; done
