#!/bin/bash
# bench.py's rank paths on the one-GPU box: its own launcher with 4 ranks, and under torch.distributed.run with 2
set -o pipefail
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 500 python bench.py --gpus 4 --steps 10 --warmup 3 --long-steps 200 --no-e2e > gpurun_out/bench_n4_rehearsal.json 2> gpurun_out/bench_n4_rehearsal.err; echo "own launcher, 4 ranks: rc=$?"; tail -c 700 gpurun_out/bench_n4_rehearsal.json; tail -3 gpurun_out/bench_n4_rehearsal.err
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --long-steps 200 --no-e2e > gpurun_out/bench_n2_torchrun.json 2> gpurun_out/bench_n2_torchrun.err; echo "torchrun, 2 ranks: rc=$?"; grep '^{' gpurun_out/bench_n2_torchrun.json | tail -c 700; tail -3 gpurun_out/bench_n2_torchrun.err
