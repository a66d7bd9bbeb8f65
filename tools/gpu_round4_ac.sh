#!/bin/bash
# back-end flags through MARAY_JIT_EXTRA (no code change): scheduling strategies against a kernel whose busy tiles are bound by their
# own chains -- frame / board / sky crops per flag, a process per run (a flag the compiler does not know fails its run: skipped)
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
run() {
  for crop in frame board sky; do
    MARAY_JIT_EXTRA="$1" timeout -k 10 200 python tools/run_crop.py chess $crop 20 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
print('EXTRA=[$1]', (lambda j: (j['crop'], j['pixel_kernel_us']))(json.loads(t[-1])) if t else 'failed')"
  done
}
run ""
run "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy"
run "-mllvm -amdgpu-schedule-relaxed-occupancy=1"
run "-mllvm -enable-post-misched=0"
run "-mllvm -amdgpu-disable-unclustered-high-rp-reschedule=1"
run "-mllvm -amdgpu-use-divergent-register-indexing"
run "-mllvm -misched=gcn-iterative-ilp"
run ""
