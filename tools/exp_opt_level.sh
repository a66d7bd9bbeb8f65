#!/bin/bash
# tools/exp_opt_level.sh — hiprtc -O3 (default) against -O1: build time and kernel time over the configurations (one GPU call)
cd $GRAFT_REPO_ROOT
export MARAY_CACHE_DIR=off
for opt in "" "-O1"; do
  echo "== MARAY_JIT_OPT='$opt'"
  export MARAY_JIT_OPT=$opt; [ -z "$opt" ] && unset MARAY_JIT_OPT
  EXP_LIBM=1 timeout -k 10 200 python tools/exp_pixels.py "chess" 2>&1 | cut -c1-400
  timeout -k 10 100 python tools/exp_tiles.py 2>&1 | grep -E "allops 4096|textured 4096" | cut -c1-200
  for a in "300" "300 colours" "1000"; do timeout -k 10 250 python tools/bench_soup.py $a 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['shapes'], d['mixed'], 'create_s', d['jit']['create_s'], 'kernel_ms', d['jit']['kernel_ms'], d['jit']['bit_exact_rows_2000_2016'])"; done
done
