#!/bin/bash
# tools/pmc_crop.sh TAG SCENE CROP — rocprofv3 of tools/run_crop.py: kernel trace + stats, then PMC groups in their own
# runs (never combined with a trace domain).  Summaries -> gpurun_out/pmc_TAG.json (tools/summarize_crop.py).
tag=$1; scene=$2; crop=$3
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/crop_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/tools/run_crop.py" $scene $crop 10 > "$out/run.json" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
cat "$out/run.json"
i=0
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_SMEM" "FETCH_SIZE" "WRITE_SIZE" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH" "SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_ANY SQ_IFETCH_LEVEL" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $pass --output-format csv -d "$out/pmc_$i" -- python3 "$root/tools/run_crop.py" $scene $crop 4 > "$out/run_pmc_$i.json" 2> "$out/pmc_$i.err" || { echo "pmc group $i failed: $pass"; tail -3 "$out/pmc_$i.err"; }
done
python3 "$root/tools/summarize_crop.py" "$out" > "$root/gpurun_out/pmc_$tag.json"
cat "$root/gpurun_out/pmc_$tag.json"
