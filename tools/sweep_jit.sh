#!/bin/bash
for cfg in "MARAY_BENCH_ROW_GUARDS=1 A=0" "MARAY_BENCH_ROW_GUARDS=1 MARAY_JIT_NO_EXPECT=1" "MARAY_BENCH_ROW_GUARDS=0 A=0" "MARAY_BENCH_ROW_GUARDS=0 MARAY_JIT_NO_EXPECT=1"; do
  echo -n "$cfg : "
  env $cfg python bench.py --backend jit --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f Mpx/s step %.3f ms kernel %.3f ms parity %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['bit_exact_vs_golden']))"
done
