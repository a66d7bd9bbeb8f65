"""N > 1 path on CPU: two gloo ranks shard a scene by rows exactly like bench.py / gen.cpp, evaluate their tiles
(with the numpy tape evaluator standing in for the device — test infrastructure), gather on rank 0 without any
data-path collective other than the final gather of the test itself, and the stitched raster equals the oracle's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from maray_amd.sharding import interleaved_blocks, interleaved_layout, max_over_ranks, scene_scale, strong_rows, weak_rows


def test_row_ranges_partition_the_image():
    for world in (1, 2, 3, 4, 8):
        for h in (1, 7, 8, 1000, 16384):
            edges = [strong_rows(r, world, h) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == h
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
        rows = [weak_rows(r, world, 4096) for r in range(world)]
        assert rows[-1][1] == rows[-1][2] == 4096 * world
        assert all(b - a == 4096 for a, b, _ in rows)


def test_bench_sharding_is_a_balanced_partition():
    for world in (1, 2, 3, 4, 8):
        sx, sy = scene_scale(world)
        w, h = 1024 * sx, 1024 * sy
        assert w * h == world * 4096 * 4096                      # weak scaling: 4096^2 pixels per rank
        seen = np.zeros(h, np.int32)
        for r in range(world):
            blocks = interleaved_blocks(r, world, h, 256)
            assert sum(b - a for a, b in blocks) == h // world   # equal rows per rank
            for a, b in blocks:
                seen[a:b] += 1
            # the same share as ONE launch of maray_hip_render_blocks_device
            y0, br, stride, nb = interleaved_layout(r, world, h, 256)
            assert [(y0 + k * stride, y0 + k * stride + br) for k in range(nb)] == blocks
        assert (seen == 1).all()                                  # every row exactly once
    assert interleaved_layout(1, 2, 1000, 256) is None           # ragged last block: no regular pattern
    assert interleaved_layout(0, 1, 1000, 256) == (0, 1000, 0, 1)
    assert [scene_scale(n) for n in (1, 2, 4, 8)] == [(4, 4), (8, 4), (8, 8), (16, 8)]


def _worker(rank, world, port, data, w, h, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import maray_amd as M
    import tape_eval
    dist.init_process_group('gloo', rank=rank, world_size=world)
    tape = M.Scene(data).lower()
    y0, y1 = strong_rows(rank, world, h)
    tile = tape_eval.cast_u8(tape_eval.render_rows(tape, w, y0, y1))          # this rank's rows only
    dist.barrier()
    elapsed = max_over_ranks(dist, float(rank + 1))                           # the bench's max-over-ranks
    parts = [None] * world if rank == 0 else None
    dist.gather_object((y0, y1, tile), parts, dst=0)                          # host-side gather (gen.cpp does this by pointer)
    if rank == 0:
        img = np.zeros((h, w, 3), np.uint8)
        for a, b, t in parts:
            img[a:b] = t
        np.save(out_path, img)
        assert elapsed == float(world)
    dist.destroy_process_group()


def test_two_ranks_render_disjoint_row_tiles(tmp_path):
    from marayb import add, div, encode, mul, nat, sin, step, sub, x, y
    from oracle_ffi import Scene as OScene
    w, h = 96, 37                                                           # odd height: uneven split
    c = [mul(step(sin(add(mul(x(), div(nat(1), nat(3))), y()))), nat(255)), add(x(), mul(y(), nat(2))), sub(nat(200), y())]
    data = encode((w, h), c)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'img.npy')
    mp.spawn(_worker, args=(2, port, data, w, h, out), nprocs=2, join=True)
    want8, _ = OScene(data).render_rows(w, h, 0, h)
    assert np.array_equal(np.load(out), want8)


def test_launcher_env_arithmetic():
    """What `bench.py --gpus N` gives each of the ranks it starts (the driver's plain command line, no torchrun)."""
    from maray_amd.sharding import device_of_rank, rank_envs
    envs = rank_envs(4, 29511, base={'PATH': '/bin', 'WORLD_SIZE': 'stale'})
    assert [e['RANK'] for e in envs] == ['0', '1', '2', '3'] == [e['LOCAL_RANK'] for e in envs]
    assert all(e['WORLD_SIZE'] == '4' and e['LOCAL_WORLD_SIZE'] == '4' for e in envs)
    assert all(e['MASTER_ADDR'] == '127.0.0.1' and e['MASTER_PORT'] == '29511' and e['PATH'] == '/bin' for e in envs)
    assert all(e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' for e in envs)
    assert rank_envs(1, 1, base={'HSA_ENABLE_IPC_MODE_LEGACY': '1'})[0]['HSA_ENABLE_IPC_MODE_LEGACY'] == '1'   # the caller's choice stays
    # one rank per device; fewer devices than ranks: shared round-robin, and said to be a rehearsal
    assert [device_of_rank(r, 8, 8) for r in range(8)] == [(r, False) for r in range(8)]
    assert [device_of_rank(r, 2, 1) for r in range(2)] == [(0, True), (0, True)]
    assert [device_of_rank(r, 4, 2) for r in range(4)] == [(0, True), (1, True), (0, True), (1, True)]
    assert device_of_rank(0, 1, 8) == (0, False)
    with pytest.raises(ValueError):
        device_of_rank(0, 2, 0)


def test_launcher_relays_rank_zero_and_fails_with_any_rank():
    from maray_amd.sharding import launch_ranks
    prog = ("import os, sys, time\n"
            "r = int(os.environ['RANK'])\n"
            "print('rank %d of %s local %s' % (r, os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK']), flush=True)\n"
            "sys.exit(int(sys.argv[1]) if r == int(sys.argv[2]) else 0) if len(sys.argv) > 2 and sys.argv[3] == 'now' else None\n"
            "time.sleep(60 if len(sys.argv) > 2 and r != int(sys.argv[2]) else 0)\n")
    rc, out = launch_ranks([sys.executable, '-c', prog], 3)
    assert rc == 0 and out.strip() == 'rank 0 of 3 local 0'              # only rank 0's stdout comes back
    import time
    t = time.monotonic()
    rc, out = launch_ranks([sys.executable, '-c', prog, '7', '2', 'now'], 3)   # rank 2 fails, the others would sleep a minute
    assert rc == 7 and time.monotonic() - t < 30
    rc, _ = launch_ranks([sys.executable, '-c', 'import time; time.sleep(60)'], 2, timeout=0.5)
    assert rc == 124


def test_bench_starts_its_own_ranks_and_fails_loudly_without_a_device():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks; here they find no HIP device,
    so there is no JSON line and the exit code is not 0 (never a CPU fallback, never an `n_gpus: 1` line)."""
    import subprocess
    if torch.cuda.device_count() > 0:
        pytest.skip('a GPU is visible: covered by the GPU suite')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and '{' not in r.stdout
    # (a rank that fails ends the run: the launcher stops the other one, which may not have got as far as its own message)
    assert 1 <= r.stderr.count('no HIP device is visible') <= 2 and '2-rank run failed' in r.stderr
    # an environment that exports WORLD_SIZE=1 to everything is not a launcher: the parent still starts its own ranks
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=dict(env, WORLD_SIZE='1'), timeout=300)
    assert r.returncode != 0 and 1 <= r.stderr.count('no HIP device is visible') <= 2 and '2-rank run failed' in r.stderr
    # a rank count that does not match --gpus is refused rather than printed as something else
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=dict(env, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0'), timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr
