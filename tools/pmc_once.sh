#!/bin/bash
# tools/pmc_once.sh "COUNTER COUNTER ..." [bench args]  — one rocprofv3 --pmc run of bench.py; prints per-launch averages for maray_jit_pixels
pmc=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_once_$$
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $pmc --output-format csv -d "$out" -- python3 "$root/bench.py" "$@" --steps 3 --warmup 1 --cpu-seconds 0 > "$out/bench.json" 2> "$out/err.txt" || { tail -5 "$out/err.txt"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, collections, sys, os
f = glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if (os.environ.get('KERNEL') or 'maray_jit_pixels') in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v)/len(v)) for k, v in agg.items()})
PY
