#!/bin/bash
# why is the sky slower in the two-row kernel?  counters of the sky crop, both kernels (one pass of <= 4 SQ counters each)
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for v in 0 1; do
  export MARAY_JIT_ROWS2=$v MARAY_JIT_TILES=2
  echo "ROWS2=$v"
  bash tools/pmc_crop_once.sh "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" chess sky
  bash tools/pmc_crop_once.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" chess sky
  bash tools/pmc_crop_once.sh "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH" chess sky
  bash tools/pmc_crop_once.sh "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM" chess sky
done 2>&1 | tee gpurun_out/r4_rows2_sky_pmc.txt
