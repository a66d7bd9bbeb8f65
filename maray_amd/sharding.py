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
