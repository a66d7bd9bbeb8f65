#!/bin/bash
# the first maray_gen_to_image call of a scene after the lowering's tables were rebuilt (tools/exp_first_call.py), then the bench
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 300 python tools/exp_first_call.py > gpurun_out/r4_first_call_2.txt 2>&1 || { tail -20 gpurun_out/r4_first_call_2.txt; exit 1; }
grep -E "first call|second call|lowering of|maray lower:" gpurun_out/r4_first_call_2.txt | tail -30
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_after_lowering.json 2> gpurun_out/bench.err || { tail gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4_bench_after_lowering.json').read().strip().splitlines()[-1])
e=j['e2e']
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'parity', j.get('bit_exact_vs_golden'))
print({k:v for k,v in e.items() if 'gen_to_image' in k and isinstance(v,(int,float))})
print('traffic profile matches', j['roofline']['traffic_profile']['matches_this_build'])
PY
