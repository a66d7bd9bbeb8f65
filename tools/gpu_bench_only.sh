#!/bin/bash
# the driver's bench command alone -> gpurun_out/bench_r4.json (tools/collect_r4.py copies it into profiles/): run after a
# collection, so that the line's roofline.traffic_profile is compared with the PMC profile that collection produced
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r4.json 2> gpurun_out/bench_r4.err || { tail -5 gpurun_out/bench_r4.err; exit 1; }
tail -c 600 gpurun_out/bench_r4.json
