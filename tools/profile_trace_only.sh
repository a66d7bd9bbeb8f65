#!/bin/bash
# tools/profile_trace_only.sh TAG [bench args...] — one rocprofv3 kernel-trace + stats run of bench.py.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" "$@" > "$out/bench_trace.json" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
tail -n 1 "$out/bench_trace.json" | cut -c1-300
cat "$out"/trace/*/*_kernel_stats.csv | cut -c1-200 | head -8
