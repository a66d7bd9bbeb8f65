#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
export MARAY_CACHE_DIR=/tmp/maray_cache
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/counters_sqc.txt
cat $GRAFT_REPO_ROOT/gpurun_out/counters_sqc.txt; echo
for cfg in "px1:MARAY_JIT_PX=1" "wave:MARAY_JIT_LAYOUT=wave"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_WAVE_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE"; do
    out=$GRAFT_REPO_ROOT/gpurun_out/ic_$name
    rm -rf $out; mkdir -p $out
    env $envs rocprofv3 --pmc $pass --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/run_crop.py chess board 4 > /dev/null 2> $out/err.txt || { echo "failed: $pass"; tail -3 $out/err.txt; continue; }
    python3 - $out "$name" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'maray_jit_pixels' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: round(sum(v)/len(v)) for k, v in agg.items()})
PY
  done
done
