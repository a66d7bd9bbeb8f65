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
