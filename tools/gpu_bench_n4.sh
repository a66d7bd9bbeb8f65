#!/bin/bash
# python bench.py --gpus 4 on the one-GPU box (a rehearsal: the ranks share the device, gloo carries the barrier) on the final tree
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 500 python bench.py --gpus 4 --steps 20 --warmup 5 > gpurun_out/bench_n4.json 2> gpurun_out/bench_n4.err; rc=$?
tail -3 gpurun_out/bench_n4.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_n4.json').read().strip().splitlines() if x.startswith('{')][-1]
b=json.loads(l); print({k:b[k] for k in ('n_gpus','value','ms_per_step','scaling','rehearsal')}, 'config4', (b.get('config4_strong') or {}).get('value'), 'parity', b.get('bit_exact_vs_golden'))
PY
exit $rc
