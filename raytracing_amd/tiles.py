"""Pixel-tile sharding of one frame over the GPUs of a node, one process per GPU.

Pixels are independent in render() (src/render.hpp:218-233 has no cross-pixel dependency), so the frame
is cut into horizontal bands dealt round-robin to the ranks; every rank holds the whole scene.  The only
exchange step is the gather of the finished bands to rank 0 (RCCL over xGMI when the backend is "nccl",
gloo on CPU for tests).  The RNG is keyed by the global pixel index, so the assembled image does not
depend on the number of ranks.
"""
import numpy as np


def group_size(height, world_size, bands_per_rank=4):
    """Rows per band: every band but the frame's last has exactly this many rows, which is what lets one
    render call cover all of a rank's bands (rt_render_params.row_group / row_stride)."""
    n = max(1, min(height, world_size * bands_per_rank))
    return -(-height // n)


def band_rows(height, world_size, bands_per_rank=4):
    """Row ranges [(y0, rows), ...] of all bands, in frame order.  Band b belongs to rank b % world_size."""
    g = group_size(height, world_size, bands_per_rank)
    return [(y, min(g, height - y)) for y in range(0, height, g)]


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


def render_interleaved(render_rows, width, height, rank, world_size, buffer, bands_per_rank=4):
    """Render all of this rank's bands with ONE call: render_rows(y0, rows, row_group, row_stride, out) renders
    `rows` rows starting at image row y0 and skipping row_group * (row_stride - 1) rows after every row_group
    rows, into `out` in that order -- the layout render_sharded produces with one call per band.  A bounce of
    a deep scene costs every call a fixed latency tail, so one call per frame instead of one per band matters
    at high rank counts.  Returns the call's stats."""
    bands = rank_bands(height, world_size, rank, bands_per_rank)
    rows = sum(r for _, _, r in bands)
    if rows == 0:
        return {}
    g = group_size(height, world_size, bands_per_rank)
    return render_rows(bands[0][1], rows, g, world_size, buffer[: rows * width * 3]) or {}


def gather_frame(buffer, width, height, rank, world_size, bands_per_rank=4, dist=None):
    """All ranks call this; rank 0 gets the assembled (height, width, 3) tensor, the others None.  One-shot
    form that allocates its receive buffers; a render loop should hold a FrameGatherer instead."""
    g = FrameGatherer(width, height, rank, world_size, bands_per_rank, dist, buffer=buffer)
    return g.gather()


class FrameGatherer:
    """The frame's only exchange step: every rank's finished bands go to rank 0 (torch.distributed gather:
    RCCL point-to-point transfers over xGMI with the "nccl" backend, gloo on CPU), which puts them back into
    frame order.  All buffers are allocated once, here."""

    def __init__(self, width, height, rank, world_size, bands_per_rank=4, dist=None, device=None, buffer=None,
                 always_gather=False):
        import torch
        self.width, self.height, self.rank, self.world, self.bpr, self.dist = width, height, rank, world_size, bands_per_rank, dist
        self.collective = world_size > 1 or always_gather       # always_gather: a single rank goes through the collective too
        per = max_rank_rows(height, world_size, bands_per_rank) * width * 3
        self.buffer = buffer if buffer is not None else torch.zeros(per, dtype=torch.float32, device=device)
        assert self.buffer.numel() >= per
        self.parts = self.frame = None
        if self.collective and rank == 0:
            self.parts = [torch.empty_like(self.buffer) for _ in range(world_size)]
            self.frame = torch.empty(height, width, 3, dtype=self.buffer.dtype, device=self.buffer.device)

    def gather(self):
        if not self.collective:
            return self.buffer[: self.height * self.width * 3].view(self.height, self.width, 3)
        self.dist.gather(self.buffer, self.parts if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        for k in range(self.world):
            cursor = 0
            for _, y0, rows in rank_bands(self.height, self.world, k, self.bpr):
                n = rows * self.width * 3
                self.frame[y0:y0 + rows] = self.parts[k][cursor:cursor + n].view(rows, self.width, 3)
                cursor += n
        return self.frame


def broadcast_bytes(payload, rank, dist, device="cpu"):
    """One broadcast of a byte buffer from rank 0 (e.g. the prepared scene of rt_scene_serialize): rank 0
    passes a numpy uint8 array, the other ranks None; every rank gets the bytes back as a numpy array.
    `device` is where the collective runs ("cuda" for RCCL, "cpu" for gloo)."""
    import torch
    size = torch.tensor([payload.size if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(size, src=0)
    n = int(size.item())
    if rank == 0:
        data = torch.from_numpy(np.ascontiguousarray(payload, np.uint8)).to(device)
    else:
        data = torch.empty(n, dtype=torch.uint8, device=device)
    dist.broadcast(data, src=0)
    return payload if rank == 0 else data.cpu().numpy()
