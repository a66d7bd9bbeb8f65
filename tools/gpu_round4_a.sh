#!/bin/bash
# tools/gpu_round4_a.sh — first GPU call of round 4: the GPU suite (new: guard-free full rasters, curved soups, the bench's
# own rank launcher, the façade's per-device cache), smoke, the driver's bench line, and `bench.py --gpus 2` rehearsed.
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
t0=$(date +%s)
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc=$rc in $(( $(date +%s) - t0 )) s"; tail -30 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 &&
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err &&
timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/bench_n2.json 2> gpurun_out/bench_n2.err
echo "bench rc=$?"; python - <<'PY'
import json
for f in ('gpurun_out/bench_n1.json', 'gpurun_out/bench_n2.json'):
    try:
        j = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, j['n_gpus'], round(j['value']), j['ms_per_step'], j['long_loop'], j['roofline']['frac'], j['roofline']['attainable_peak'], j.get('rehearsal'), (j.get('config4_strong') or {}).get('value'), j['config']['bit_exact_vs_golden'])
    except Exception as e:
        print(f, 'unreadable', e)
PY
