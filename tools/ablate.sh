#!/bin/bash
# tools/ablate.sh — re-measure every ablation knob of DESIGN.md §7.1 on chess @4096² (one bench.py run each, no CPU baseline).
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ablate.log
: > "$out"
run() {
  echo "== $*" >> "$out"
  env "$@" timeout -k 10 200 python "$root/bench.py" --cpu-seconds 0 --steps 30 --warmup 5 2>/dev/null | tail -n 1 |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%s: %.0f Mpx/s, step %.4f ms, dominant kernel %.4f ms, parity %s' % (d['config']['backend'], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['bit_exact_vs_golden']))" >> "$out" 2>&1 || echo "failed" >> "$out"
}
run MARAY_X=default
run MARAY_JIT_ROW_GUARDS=0
run MARAY_JIT_TILES=2
run MARAY_JIT_TILES=4
run MARAY_JIT_TILES=16
run MARAY_JIT_KTAB=0
run MARAY_JIT_YLDS=1
run MARAY_JIT_GLDS=0
run MARAY_JIT_NO_ORDER=1
run MARAY_JIT_ROWS_REVERSED=1
run MARAY_BENCH_BACKEND=tape-smem
run MARAY_BENCH_BACKEND=tape-smem MARAY_TAPE_GENERIC=1
run MARAY_BENCH_BACKEND=tape-smem MARAY_TAPE_ROW_GUARDS=1
run MARAY_BENCH_BACKEND=tape-smem MARAY_TAPE_KEEP_ORDER=1
run MARAY_BENCH_BACKEND=tape
cat "$out"
