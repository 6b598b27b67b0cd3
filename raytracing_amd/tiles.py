"""Pixel-tile sharding of one frame over the GPUs of a node, one process per GPU.

Pixels are independent in render() (src/render.hpp:218-233 has no cross-pixel dependency), so the frame
is cut into horizontal bands dealt round-robin to the ranks; every rank holds the whole scene.  The only
exchange step is the gather of the finished bands to rank 0 (RCCL over xGMI when the backend is "nccl",
gloo on CPU for tests).  The RNG is keyed by the global pixel index, so the assembled image does not
depend on the number of ranks.
"""
import numpy as np


def band_rows(height, world_size, bands_per_rank=4):
    """Row ranges [(y0, rows), ...] of all bands, in frame order.  Band b belongs to rank b % world_size."""
    n = max(1, min(height, world_size * bands_per_rank))
    edges = np.linspace(0, height, n + 1).round().astype(int)
    return [(int(a), int(b - a)) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def rank_bands(height, world_size, rank, bands_per_rank=4):
    return [(i, y0, rows) for i, (y0, rows) in enumerate(band_rows(height, world_size, bands_per_rank))
            if i % world_size == rank]


def max_rank_rows(height, world_size, bands_per_rank=4):
    return max(sum(r for _, _, r in rank_bands(height, world_size, k, bands_per_rank)) for k in range(world_size))


def render_sharded(render_rect, width, height, rank, world_size, buffer, bands_per_rank=4):
    """Render this rank's bands back to back into `buffer` (a flat float32 torch tensor with room for
    max_rank_rows*width*3 values).  render_rect(x0, y0, w, h, out_tensor_view) renders one rectangle and
    returns its stats dict.  Returns the summed stats."""
    total = {}
    cursor = 0
    for _, y0, rows in rank_bands(height, world_size, rank, bands_per_rank):
        n = rows * width * 3
        st = render_rect(0, y0, width, rows, buffer[cursor:cursor + n])
        cursor += n
        for k, v in (st or {}).items():
            if isinstance(v, (int, float)):
                total[k] = total.get(k, 0) + v
    return total


def gather_frame(buffer, width, height, rank, world_size, bands_per_rank=4, dist=None):
    """All ranks call this; rank 0 gets the assembled (height, width, 3) tensor, the others None.
    One collective per frame: every rank contributes its (equal-sized, padded) band buffer."""
    import torch
    if world_size == 1:
        rows = sum(r for _, _, r in rank_bands(height, 1, 0, bands_per_rank))
        return buffer[: rows * width * 3].view(height, width, 3)
    gathered = torch.empty(world_size * buffer.numel(), dtype=buffer.dtype, device=buffer.device)
    dist.all_gather_into_tensor(gathered, buffer)
    if rank != 0:
        return None
    frame = torch.empty(height, width, 3, dtype=buffer.dtype, device=buffer.device)
    per = buffer.numel()
    for k in range(world_size):
        cursor = k * per
        for _, y0, rows in rank_bands(height, world_size, k, bands_per_rank):
            n = rows * width * 3
            frame[y0:y0 + rows] = gathered[cursor:cursor + n].view(rows, width, 3)
            cursor += n
    return frame
