#!/bin/bash
# tools/gpu_profile_r2.sh — everything DESIGN.md section 7 quotes, in one GPU call; summaries go to gpurun_out/ and from
# there (tools/summarize_profile.py, by hand) into profiles/r2_*.
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
echo "== bench (driver's command)"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r2.json 2> gpurun_out/bench_r2.err || { tail -5 gpurun_out/bench_r2.err; exit 1; }
tail -c 1500 gpurun_out/bench_r2.json
export MARAY_CACHE_DIR=/tmp/maray_cache
echo "== profile jit"
bash tools/profile_bench.sh jit --cpu-seconds 0 --no-cold --no-e2e --steps 30 --warmup 5 > gpurun_out/profile_jit.log 2>&1 || { tail -5 gpurun_out/profile_jit.log; exit 1; }
echo "== crops"
bash tools/pmc_crop.sh board chess board > gpurun_out/crop_board.log 2>&1 || tail -3 gpurun_out/crop_board.log
bash tools/pmc_crop.sh sky chess sky > gpurun_out/crop_sky.log 2>&1 || tail -3 gpurun_out/crop_sky.log
bash tools/pmc_crop.sh allops allops x > gpurun_out/crop_allops.log 2>&1 || tail -3 gpurun_out/crop_allops.log
bash tools/pmc_crop.sh radial radial x > gpurun_out/crop_radial.log 2>&1 || tail -3 gpurun_out/crop_radial.log
echo "== ablations"
timeout -k 10 1000 python tools/exp_pixels.py "default:" "px1 (round-1 layout):MARAY_JIT_PX=1" "coop:MARAY_JIT_LAYOUT=coop" "wide general:MARAY_JIT_WIDE=1" "ybool off:MARAY_JIT_YBOOL=0" "derived off:MARAY_JIT_DERIVED=0" "gw sload:MARAY_JIT_GW=sload" "tiles 1:MARAY_JIT_TILES=1" "tiles 4:MARAY_JIT_TILES=4" "no order:MARAY_JIT_NO_ORDER=1" "row overlap:MARAY_JIT_ROW_OVERLAP=1" "ktab off:MARAY_JIT_KTAB=0" "row guards off:MARAY_JIT_ROW_GUARDS=0" "guards 256x8 (round 1):MARAY_JIT_GUARD_W=256,MARAY_JIT_GUARD_H=8" "guards 128x16:MARAY_JIT_GUARD_W=128,MARAY_JIT_GUARD_H=16" "guards 64x8:MARAY_JIT_GUARD_H=8" "guards 64x16:MARAY_JIT_GUARD_H=16" "guards 64x64:MARAY_JIT_GUARD_H=64" "pass sky:MARAY_JIT_PASS_SKY=1" "two pixels per lane, 256x8:MARAY_JIT_GUARD_W=256,MARAY_JIT_GUARD_H=8,MARAY_JIT_NARROW=2" "regions from 24:MARAY_JIT_MIN_REGION=24" "regions all:MARAY_JIT_MIN_REGION=0" "ROW regions all:MARAY_JIT_ROW_MIN_REGION=0" "ROW regions none:MARAY_JIT_ROW_MIN_REGION=100000" "strips rotated by the row:MARAY_JIT_SWIZZLE=1" "every region unlikely:MARAY_JIT_EXPECT=0" "two loops over a strip:MARAY_JIT_TWO_LOOPS=1" "a loop of its own for strips of sky:MARAY_JIT_TWO_LOOPS=2" "machine LICM off:MARAY_JIT_EXTRA=-mllvm -disable-machine-licm" "grid padded x3 (blocks that exit at once):MARAY_JIT_PAD_GRID=3" "grid padded x2 (real blocks on half the XCDs):MARAY_JIT_PAD_GRID=2" "ROW y values only (wrong pixels):MARAY_JIT_ROW_PART=1" "ROW guards only (wrong pixels):MARAY_JIT_ROW_PART=2" "ROW empty (wrong pixels):MARAY_JIT_ROW_PART=3" "default again:" > gpurun_out/ablations_r2.jsonl 2> gpurun_out/ablations_r2.err; cat gpurun_out/ablations_r2.jsonl
echo "== other configs"
python tools/bench_configs.py > gpurun_out/other_configs_r2.json 2>/dev/null; head -c 600 gpurun_out/other_configs_r2.json
echo "== config 4 on one GPU"
timeout -k 10 400 python tools/exp_strong.py "default:" "strips rotated by the row:MARAY_JIT_SWIZZLE=1" "every region unlikely:MARAY_JIT_EXPECT=0" "default again:" > gpurun_out/strong_r2.jsonl 2>&1; cat gpurun_out/strong_r2.jsonl
echo "== interpreters"
bash tools/profile_trace_only.sh tape_smem --backend tape-smem --cpu-seconds 0 --no-cold --no-e2e --steps 30 --warmup 5 | tail -6
bash tools/profile_trace_only.sh tape_lds --backend tape --cpu-seconds 0 --no-cold --no-e2e --steps 10 --warmup 2 | tail -6
