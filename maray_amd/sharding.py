"""Row-tile sharding used by bench.py and the multi-rank tests.

Pixels are independent (src/render.rs:85), so N ranks own N disjoint, contiguous
row ranges and never exchange pixel data; the only collectives are the timing
barrier and the max-over-ranks of the elapsed time."""


def weak_rows(rank, world, rows_per_rank):
    """Weak scaling: every rank owns `rows_per_rank` rows of a rows_per_rank*world tall image."""
    return rank * rows_per_rank, (rank + 1) * rows_per_rank, rows_per_rank * world


def strong_rows(rank, world, h):
    """Strong scaling (what maray_gen_to_image does per device, gen.cpp): rows [h*r/N, h*(r+1)/N)."""
    return (h * rank) // world, (h * (rank + 1)) // world


def scene_scale(world):
    """(sx, sy) multipliers of the stored 1024x1024 scene so that N ranks share one image of N x 4096^2 pixels:
    N=1 -> 4096x4096, 2 -> 8192x4096, 4 -> 8192x8192, 8 -> 16384x8192 (non powers of two stretch vertically)."""
    if world & (world - 1) == 0:
        k = 1
        while k * k < world:
            k *= 2
        return 4 * k, 4 * (world // k)
    return 4, 4 * world


def interleaved_blocks(rank, world, h, block_rows):
    """Row blocks of `block_rows` dealt round-robin to the ranks: [(y0, y1), ...] for this rank.  Balances
    data-dependent cost (empty sky vs chess board) without any exchange; with world == 1 it is the whole image."""
    if world == 1:
        return [(0, h)]
    return [(y, min(h, y + block_rows)) for b, y in enumerate(range(0, h, block_rows)) if b % world == rank]


def interleaved_layout(rank, world, h, block_rows):
    """The same share as interleaved_blocks, as one regular pattern (y0, block_rows, block_stride, n_blocks) for
    maray_hip_render_blocks_device -- or None when the last block is ragged and the blocks have to go one by one."""
    blocks = interleaved_blocks(rank, world, h, block_rows)
    if not blocks:
        return None
    if len(blocks) == 1:
        return blocks[0][0], blocks[0][1] - blocks[0][0], 0, 1
    if any(b - a != block_rows for a, b in blocks):
        return None
    return blocks[0][0], block_rows, world * block_rows, len(blocks)


def max_over_ranks(dist, value, device=None):
    """MAX all-reduce of a python float (no-op when not initialised)."""
    import torch
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rank_envs(n, port, base=None, addr='127.0.0.1'):
    """Environment of each of the n ranks of ONE node, as torch.distributed.run sets it (RANK, LOCAL_RANK, WORLD_SIZE,
    LOCAL_WORLD_SIZE, MASTER_ADDR, MASTER_PORT).  HSA_ENABLE_IPC_MODE_LEGACY=0 is kept (set when absent): the hosts this
    runs on support dmabuf IPC only, and RCCL needs it across processes."""
    import os
    base = dict(os.environ if base is None else base)
    out = []
    for r in range(n):
        e = dict(base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR=addr,
                 MASTER_PORT=str(port))
        e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        out.append(e)
    return out


def device_of_rank(local_rank, world, n_devices):
    """(device index, rehearsal) for a local rank of `world` ranks on one node with n_devices visible: one rank per device;
    with fewer devices than ranks the ranks share them round-robin -- a rehearsal (the sharding, the launches and the
    timing discipline are the real ones; the figure is not a scaling figure), for which the process group is gloo:
    RCCL refuses two ranks on one device."""
    if n_devices <= 0:
        raise ValueError('no HIP device is visible')
    return local_rank % n_devices, n_devices < world


def free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(argv, n, timeout=None, env=None):
    """Start n fresh processes `argv`, one per rank (environment: rank_envs), wait for all of them and return
    (exit code, rank 0's stdout).  Nothing is exec'ed: the ranks are children, and the caller need not be GPU-free for
    them to start clean (it should be, though: a parent that holds a device context costs every rank memory).  Ranks
    other than 0 write their stdout to our stderr.  The exit code is 0 only when every rank's is; when one rank fails or
    the time is up the others are ended -- they would wait for it at the next barrier for ever."""
    import subprocess
    import sys
    import threading
    import time
    procs = []
    for r, e in enumerate(rank_envs(n, free_port(), base=env)):
        procs.append(subprocess.Popen(argv, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    box = {}
    reader = threading.Thread(target=lambda: box.setdefault('out', procs[0].stdout.read()), daemon=True)
    reader.start()
    t0 = time.monotonic()
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            c = procs[r].poll()
            if c is not None:
                pending.discard(r)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 128 - c            # a signal reads as the shell prints it
        if rc == 0 and timeout is not None and time.monotonic() - t0 > timeout:
            rc = 124
        if rc != 0:
            for r in pending:
                procs[r].terminate()
            deadline = time.monotonic() + 10
            for r in pending:
                try:
                    procs[r].wait(max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            break
        if pending:
            time.sleep(0.05)
    reader.join(10)
    return rc, box.get('out', '')
