#!/bin/bash
# tools/profile_bench.sh TAG [bench args...] — rocprofv3 runs of bench.py on the GPU box.
# Kernel trace + stats in one run; PMC counters in their own runs (never combined with trace domains).
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" "$@" > "$out/bench_trace.json" 2> "$out/trace.err" || exit 1
tail -n 1 "$out/bench_trace.json"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pass"
  rocprofv3 --pmc $pass --output-format csv -d "$out/pmc_$name" -- python3 "$root/bench.py" "$@" --steps 3 --warmup 1 > "$out/bench_pmc_$name.json" 2> "$out/pmc_$name.err" || { tail -5 "$out/pmc_$name.err"; exit 1; }
done
find "$out" -name "*.csv" | head -40
