#!/bin/bash
# tools/bench_env.sh LOGFILE VAR=VALUE... -- runs bench.py (no CPU baseline) with the given environment under a timeout
log=$1; shift
env "$@" tools/gpu_step.sh 120 "$log" python bench.py --cpu-seconds 0 --steps 30 --warmup 5
