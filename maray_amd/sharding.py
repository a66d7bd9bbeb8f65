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


def max_over_ranks(dist, value, device=None):
    """MAX all-reduce of a python float (no-op when not initialised)."""
    import torch
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
