#!/bin/bash
# tools/gpu_profile_r4.sh — everything DESIGN.md section 7 quotes for round 4, in one GPU call; summaries go to gpurun_out/
# and from there (tools/collect_r4.py) into profiles/r4_*.
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
echo "== bench (driver's command)"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r4.json 2> gpurun_out/bench_r4.err || { tail -5 gpurun_out/bench_r4.err; exit 1; }
tail -c 400 gpurun_out/bench_r4.json
echo "== bench --gpus 2 (its own ranks; a rehearsal when one GPU is visible)"
timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/bench_r4_n2.json 2> gpurun_out/bench_r4_n2.err || tail -5 gpurun_out/bench_r4_n2.err
tail -c 300 gpurun_out/bench_r4_n2.json
export MARAY_CACHE_DIR=/tmp/maray_cache
echo "== profile jit"
bash tools/profile_bench.sh jit --cpu-seconds 0 --no-cold --no-e2e --steps 30 --warmup 5 --long-steps 0 > gpurun_out/profile_jit.log 2>&1 || { tail -5 gpurun_out/profile_jit.log; exit 1; }
echo "== crops"
for c in "board chess board" "sky chess sky" "textured textured x" "allops allops x" "radial radial x"; do
  set -- $c
  bash tools/pmc_crop.sh r4_$1 $2 $3 > gpurun_out/crop_r4_$1.log 2>&1 || tail -3 gpurun_out/crop_r4_$1.log
done
echo "== ablations"
timeout -k 10 600 python tools/exp_pixels.py "default:" "Step(x + k) as add + compare (round 3):MARAY_JIT_FUSE_CMP=0" "two rows per wavefront, 2 tiles:MARAY_JIT_ROWS2=1,MARAY_JIT_TILES=2" "two rows per wavefront, 1 tile:MARAY_JIT_ROWS2=1" "tree walked as written (round 2):MARAY_JIT_REDUCE=0" "tiles 1:MARAY_JIT_TILES=1" "tiles 3:MARAY_JIT_TILES=3" "tiles 4:MARAY_JIT_TILES=4" "row guards off:MARAY_JIT_ROW_GUARDS=0" "guards 256x8 (round 1):MARAY_JIT_GUARD_W=256,MARAY_JIT_GUARD_H=8" "guards 64x16:MARAY_JIT_GUARD_H=16" "guards 64x64:MARAY_JIT_GUARD_H=64" "regions from 12 (round 3):MARAY_JIT_MIN_REGION=12" "small textured programs one pixel per lane:MARAY_JIT_WIDE_APP=0" "default again:" > gpurun_out/ablations_r4.jsonl 2> gpurun_out/ablations_r4.err; cat gpurun_out/ablations_r4.jsonl | cut -c1-260
echo "== other configs"
python tools/bench_configs.py > gpurun_out/other_configs_r4.json 2>/dev/null; head -c 300 gpurun_out/other_configs_r4.json
MARAY_JIT_TEXEL_ONCE=0 python tools/bench_configs.py > gpurun_out/other_configs_r4_texel_per_app.json 2>/dev/null
echo "== soups"
(for a in "300" "300 colours" "1000"; do timeout -k 10 200 python tools/bench_soup.py $a; done) > gpurun_out/soups_r4.jsonl 2>&1; cat gpurun_out/soups_r4.jsonl | cut -c1-300
echo "== config 4 on one GPU"
timeout -k 10 300 python bench.py --scaling strong --steps 20 --warmup 5 --cpu-seconds 0 --no-cold --no-e2e > gpurun_out/bench_strong_r4.json 2>/dev/null; tail -c 300 gpurun_out/bench_strong_r4.json
echo "== sizes"
timeout -k 10 300 python tools/exp_sizes.py > gpurun_out/sizes_r4.jsonl 2>&1; cat gpurun_out/sizes_r4.jsonl | cut -c1-200
echo "== first call"
timeout -k 10 300 python tools/exp_first_call.py > gpurun_out/r4_first_call.txt 2>&1; grep -E "first call|second call|pageable|textured|alone" gpurun_out/r4_first_call.txt
echo "== interpreters"
bash tools/profile_trace_only.sh tape_smem --backend tape-smem --cpu-seconds 0 --no-cold --no-e2e --steps 30 --warmup 5 --long-steps 0 | tail -4
bash tools/profile_trace_only.sh tape_lds --backend tape --cpu-seconds 0 --no-cold --no-e2e --steps 10 --warmup 2 --long-steps 0 | tail -4
