#!/bin/bash
# tools/pmc_crop_once.sh "COUNTER COUNTER ..." SCENE CROP — one rocprofv3 --pmc run of tools/run_crop.py; per-launch averages per maray kernel
pmc=$1; scene=$2; crop=$3
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_crop_once_$$
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc $pmc --output-format csv -d "$out" -- python3 "$root/tools/run_crop.py" $scene $crop 4 > "$out/run.json" 2> "$out/err.txt" || { tail -5 "$out/err.txt"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'maray' in r['Kernel_Name']: agg[r['Kernel_Name'][:24]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items(): print(k, {n: round(sum(v)/len(v)) for n, v in c.items()})
PY
