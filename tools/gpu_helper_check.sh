#!/bin/bash
# tools/gpu_helper_check.sh — the out-of-process builds (maray_jitc) on the GPU box: cold context with helpers, in-process, and
# from a process that imported PyTorch first (its own hiprtc); then the GPU tests that create many contexts.
cd $GRAFT_REPO_ROOT
export MARAY_TRACE_INIT=1 AMD_COMGR_CACHE=0
echo "== cold, helpers"; MARAY_CACHE_DIR=/tmp/mc_h1 python tools/exp_init_trace.py 2 2>&1 | grep -E "code objects|ctx_ms"
echo "== cold, in-process"; MARAY_JIT_HELPER=0 MARAY_CACHE_DIR=/tmp/mc_h2 python tools/exp_init_trace.py 2 2>&1 | grep -E "code objects|ctx_ms"
echo "== a process with PyTorch"
python - <<'PY' 2>&1 | tail -4
import torch, os, time, sys
torch.cuda.init(); x = torch.zeros(4, device="cuda")
os.environ["MARAY_CACHE_DIR"] = "/tmp/mc_h3"
root = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, root)
import maray_amd as M
s = M.Scene(open(os.path.join(root, "tests/golden/chess.maray"), "rb").read()); s.rescale(4, 4)
t = time.perf_counter(); ctx = M.Context(s.lower(), backend=M.BACKEND_JIT); print("ctx after torch", round(time.perf_counter() - t, 2), "s")
a, _ = ctx.render_rows(4096, 4096, 2048, 2056, want_f64=False); print("sum", int(a.sum()))
PY
unset AMD_COMGR_CACHE MARAY_TRACE_INIT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "knobs or gen_to_image" 2>&1 | tail -3
