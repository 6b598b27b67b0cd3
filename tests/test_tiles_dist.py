"""Multi-rank path on CPU: band assignment and the gather of finished bands to rank 0 over
torch.distributed (gloo, world_size 2).  The renderer is replaced by a function of the global pixel
coordinates so the assembled frame has a known answer; the GPU render itself is covered by -m gpu."""
import os
import socket

import numpy as np
import pytest

from raytracing_amd import tiles


@pytest.mark.parametrize("height,world,bpr", [(1080, 1, 4), (1080, 2, 4), (1080, 8, 4), (2160, 8, 4), (7, 8, 4), (5, 2, 1)])
def test_bands_partition_the_frame(height, world, bpr):
    bands = tiles.band_rows(height, world, bpr)
    assert bands[0][0] == 0 and sum(r for _, r in bands) == height
    for (y0, r), (y1, _) in zip(bands[:-1], bands[1:]):
        assert y0 + r == y1
    seen = []
    for rank in range(world):
        seen += [(y0, r) for _, y0, r in tiles.rank_bands(height, world, rank, bpr)]
    assert sorted(seen) == bands
    rows = [sum(r for _, _, r in tiles.rank_bands(height, world, k, bpr)) for k in range(world)]
    assert max(rows) == tiles.max_rank_rows(height, world, bpr)
    if height >= world * bpr:
        assert max(rows) - min(rows) <= tiles.group_size(height, world, bpr)     # balanced to within one band
    g = tiles.group_size(height, world, bpr)
    assert all(r == g for _, r in bands[:-1]) and 0 < bands[-1][1] <= g           # fixed-size groups


def _expected(width, height):
    y, x = np.mgrid[0:height, 0:width]
    return np.stack([x + 1000.0 * y, x * 0.5, y * 0.25], -1).astype(np.float32)


def _worker(rank, world, port, width, height, bpr, result):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        want = torch.from_numpy(_expected(width, height))
        buf = torch.zeros(tiles.max_rank_rows(height, world, bpr) * width * 3)

        def render_rect(x0, y0, w, h, out):
            out.copy_(want[y0:y0 + h, x0:x0 + w].reshape(-1))
            return {"segments": w * h, "ms_trace": 1.0}

        st = tiles.render_sharded(render_rect, width, height, rank, world, buf, bpr)
        frame = tiles.gather_frame(buf, width, height, rank, world, bpr, dist)
        total = torch.tensor([float(st["segments"])])
        dist.all_reduce(total)
        if rank == 0:
            result["ok"] = bool(torch.equal(frame, want)) and int(total.item()) == width * height
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("width,height,bpr", [(64, 37, 4), (33, 5, 2)])
def test_gather_reassembles_frame_world_size_2(width, height, bpr):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    result = mgr.dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, width, height, bpr, result)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert result.get("ok") is True


def _scene_worker(rank, world, port, width, height, bpr, result):
    """The whole multi-rank frame loop with a stub renderer: rank 0 prepares the scene and broadcasts it as
    bytes, every rank turns the bytes back into a scene description, renders its bands (the stub paints the
    global pixel coordinates plus a checksum of the scene it received) and rank 0 gathers, twice, into the
    same preallocated buffers."""
    import ctypes
    import torch
    import torch.distributed as dist
    from raytracing_amd import _lib, scenes
    from raytracing_amd.render import RayTraceData, serialize_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blob = None
        if rank == 0:
            blob = serialize_scene(RayTraceData(scenes.builtin_scene(3).preprocess()))
        blob = tiles.broadcast_bytes(blob, rank, dist)
        desc = _lib.RtSceneDesc()
        _lib.check(_lib.lib().rt_scene_desc_from_blob(_lib.ptr(blob), ctypes.c_uint64(blob.size), ctypes.byref(desc)))
        assert (desc.inode_count, desc.leaf_count, desc.model_count) == (69, 56, 30)
        checksum = float(int(blob.astype(np.uint64).sum()) % 1000)
        want = torch.from_numpy(_expected(width, height)) + checksum
        gatherer = tiles.FrameGatherer(width, height, rank, world, bpr, dist)
        frames = []
        for step in range(2):
            def render_rect(x0, y0, w, h, out):
                out.copy_((want[y0:y0 + h, x0:x0 + w] + step).reshape(-1))
                return {"segments": w * h}
            tiles.render_sharded(render_rect, width, height, rank, world, gatherer.buffer, bpr)
            frame = gatherer.gather()
            frames.append(None if frame is None else frame.clone())
        sums = torch.tensor([checksum])
        dist.all_reduce(sums)
        if rank == 0:
            result["ok"] = bool(torch.equal(frames[0], want)) and bool(torch.equal(frames[1], want + 1)) \
                and float(sums.item()) == world * checksum and gatherer.frame.data_ptr() == frame.data_ptr()
        else:
            assert frames == [None, None] and gatherer.parts is None
    finally:
        dist.destroy_process_group()


def test_broadcast_scene_render_gather_world_size_2():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    result = mgr.dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_scene_worker, args=(r, 2, port, 48, 29, 3, result)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert result.get("ok") is True


def test_interleaved_call_covers_the_same_rows_as_the_bands():
    import torch
    w, h, world, bpr = 12, 53, 4, 3
    want = torch.from_numpy(_expected(w, h))
    g = tiles.group_size(h, world, bpr)
    for rank in range(world):
        a = torch.zeros(tiles.max_rank_rows(h, world, bpr) * w * 3)
        b = torch.zeros_like(a)

        def render_rect(x0, y0, ww, hh, out):
            out.copy_(want[y0:y0 + hh, x0:x0 + ww].reshape(-1))

        def render_rows(y0, rows, row_group, row_stride, out):
            ys = [y0 + r + (r // row_group) * row_group * (row_stride - 1) for r in range(rows)]     # the device's row map
            assert row_group == g and row_stride == world and max(ys) < h
            out.copy_(want[ys].reshape(-1))

        tiles.render_sharded(render_rect, w, h, rank, world, a, bpr)
        tiles.render_interleaved(render_rows, w, h, rank, world, b, bpr)
        assert torch.equal(a, b)


def test_single_rank_is_a_view():
    import torch
    w, h = 16, 9
    want = torch.from_numpy(_expected(w, h))
    buf = torch.zeros(tiles.max_rank_rows(h, 1) * w * 3)

    def render_rect(x0, y0, ww, hh, out):
        out.copy_(want[y0:y0 + hh, x0:x0 + ww].reshape(-1))
        return {}

    tiles.render_sharded(render_rect, w, h, 0, 1, buf)
    assert torch.equal(tiles.gather_frame(buf, w, h, 0, 1), want)
