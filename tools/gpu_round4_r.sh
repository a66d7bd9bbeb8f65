#!/bin/bash
# tools/native/sky_cost.hip: the sky variant alone in kernels that differ in what they reserve (LDS, VGPRs, SGPRs)
set -o pipefail
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/native/sky_cost.hip -o /tmp/sky_cost || exit 1
timeout -k 10 200 /tmp/sky_cost > gpurun_out/r4_sky_cost.json || exit 1
cat gpurun_out/r4_sky_cost.json
