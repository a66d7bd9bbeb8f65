#!/bin/bash
# tools/gpu_profile_r3.sh — everything DESIGN.md section 7 quotes for round 3, in one GPU call; summaries go to gpurun_out/
# and from there (tools/summarize_profile.py --round r3, tools/collect_r3.py) into profiles/r3_*.
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
echo "== bench (driver's command)"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r3.json 2> gpurun_out/bench_r3.err || { tail -5 gpurun_out/bench_r3.err; exit 1; }
tail -c 600 gpurun_out/bench_r3.json
export MARAY_CACHE_DIR=/tmp/maray_cache
echo "== profile jit"
bash tools/profile_bench.sh jit --cpu-seconds 0 --no-cold --no-e2e --steps 30 --warmup 5 > gpurun_out/profile_jit.log 2>&1 || { tail -5 gpurun_out/profile_jit.log; exit 1; }
echo "== crops"
bash tools/pmc_crop.sh r3_board chess board > gpurun_out/crop_r3_board.log 2>&1 || tail -3 gpurun_out/crop_r3_board.log
bash tools/pmc_crop.sh r3_sky chess sky > gpurun_out/crop_r3_sky.log 2>&1 || tail -3 gpurun_out/crop_r3_sky.log
bash tools/pmc_crop.sh r3_allops allops x > gpurun_out/crop_r3_allops.log 2>&1 || tail -3 gpurun_out/crop_r3_allops.log
bash tools/pmc_crop.sh r3_radial radial x > gpurun_out/crop_r3_radial.log 2>&1 || tail -3 gpurun_out/crop_r3_radial.log
echo "== ablations"
timeout -k 10 600 python tools/exp_pixels.py "default:" "tree walked as written (round 2):MARAY_JIT_REDUCE=0" "tiles 1:MARAY_JIT_TILES=1" "tiles 4:MARAY_JIT_TILES=4" "row guards off:MARAY_JIT_ROW_GUARDS=0" "guards 256x8 (round 1):MARAY_JIT_GUARD_W=256,MARAY_JIT_GUARD_H=8" "guards 128x16:MARAY_JIT_GUARD_W=128,MARAY_JIT_GUARD_H=16" "guards 64x8:MARAY_JIT_GUARD_H=8" "guards 64x16:MARAY_JIT_GUARD_H=16" "guards 64x64:MARAY_JIT_GUARD_H=64" "regions all:MARAY_JIT_MIN_REGION=0" "regions from 24:MARAY_JIT_MIN_REGION=24" "regions none:MARAY_JIT_MIN_REGION=100000" "-O1:MARAY_JIT_OPT=-O1" "default again:" > gpurun_out/ablations_r3.jsonl 2> gpurun_out/ablations_r3.err; cat gpurun_out/ablations_r3.jsonl
echo "== other configs"
python tools/bench_configs.py > gpurun_out/other_configs_r3.json 2>/dev/null; head -c 400 gpurun_out/other_configs_r3.json
echo "== soups"
(for a in "300" "300 colours" "1000"; do timeout -k 10 200 python tools/bench_soup.py $a; MARAY_JIT_REDUCE=0 timeout -k 10 200 python tools/bench_soup.py $a; done) > gpurun_out/soups_r3.jsonl 2>&1; cat gpurun_out/soups_r3.jsonl
echo "== config 4 on one GPU"
timeout -k 10 400 python tools/exp_strong.py "default:" "tree walked as written:MARAY_JIT_REDUCE=0" "default again:" > gpurun_out/strong_r3.jsonl 2>&1; cat gpurun_out/strong_r3.jsonl
timeout -k 10 300 python bench.py --scaling strong --steps 20 --warmup 5 --cpu-seconds 0 --no-cold --no-e2e > gpurun_out/bench_strong_r3.json 2>/dev/null; tail -c 300 gpurun_out/bench_strong_r3.json
echo "== sizes"
timeout -k 10 300 python tools/exp_sizes.py > gpurun_out/sizes_r3.jsonl 2>&1; cat gpurun_out/sizes_r3.jsonl
echo "== interpreters"
bash tools/profile_trace_only.sh tape_smem --backend tape-smem --cpu-seconds 0 --no-cold --no-e2e --steps 30 --warmup 5 | tail -6
bash tools/profile_trace_only.sh tape_lds --backend tape --cpu-seconds 0 --no-cold --no-e2e --steps 10 --warmup 2 | tail -6
