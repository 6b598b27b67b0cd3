"""rt_render_device_static, the entry point that allocates nothing, against rt_render_device: identical
radiance bits and work counters on two calls in a row, the second with a smaller rectangle, then the size
limit.  Runs last so that a failure here hides no other test."""
import numpy as np
import pytest

from helpers import same_bits

from raytracing_amd import _lib, scenes
from raytracing_amd.render import DeviceScene, RayTraceData

pytestmark = pytest.mark.gpu


def test_static_workspace_render_matches_allocating_render():
    import torch
    sc = scenes.cornell_scene(wall_grid=8, box_grid=3).preprocess()
    ds = DeviceScene(RayTraceData(sc))
    try:
        w, h = 640, 400
        p2r = sc.pixel_to_ray_transform(w, h)
        stream = torch.cuda.current_stream().cuda_stream
        free_before = None
        for rect, spp in (((0, 0, 640, 400), 5), ((33, 17, 301, 150), 3)):
            n = rect[2] * rect[3] * 3
            p = ds.make_params(w, h, sc.camera.position, p2r, spp=spp, seed=31, rect=rect, flags=_lib.RT_FLAG_COUNT_WORK)
            want = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
            got = torch.full((n,), -2.0, dtype=torch.float32, device="cuda")
            st_want = ds.render_device(p, want, stream=stream)
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]
            st_got = ds.render_device_static(p, got, stream=stream)
            torch.cuda.synchronize()
            # nothing was allocated (or freed) on the device by the static call
            assert torch.cuda.mem_get_info()[0] == free_before
            assert same_bits(got.cpu().numpy(), want.cpu().numpy())
            for key in ("paths", "segments", "shaded_hits", "box_tests", "tris_tested", "leaf_visits", "inode_visits"):
                assert st_got[key] == st_want[key], key
            assert st_got["box_filter_errors"] == 0 and st_got["paths"] == rect[2] * rect[3] * spp
            # 640x400x5 paths do not fit the fixed workspace at once: the static call ran more, smaller batches
            assert st_got["batches"] >= st_want["batches"]
        # without stats the call only enqueues; the result is the same once the stream has drained
        p = ds.make_params(w, h, sc.camera.position, p2r, spp=2, seed=7, rect=(100, 100, 64, 64))
        a = torch.zeros(64 * 64 * 3, dtype=torch.float32, device="cuda")
        b = torch.zeros_like(a)
        ds.render_device(p, a, stream=stream, want_stats=False)
        ds.render_device_static(p, b, stream=stream, want_stats=False)
        torch.cuda.synchronize()
        assert same_bits(a.cpu().numpy(), b.cpu().numpy()) and float(a.max()) > 0
        # beyond the fixed capacity: a clear error, nothing rendered
        big = ds.make_params(1920, 1080, sc.camera.position, sc.pixel_to_ray_transform(1920, 1080), spp=1)
        out = torch.zeros(1920 * 1080 * 3, dtype=torch.float32, device="cuda")
        with pytest.raises(_lib.RtError, match="262144"):
            ds.render_device_static(big, out, stream=stream)
        # the estimator extensions and the alternative accelerator run on the same fixed workspace
        ds.build_bvh(4)
        flags = _lib.RT_FLAG_ACCEL_BVH | _lib.RT_FLAG_LOBE_SAMPLING | _lib.RT_FLAG_RUSSIAN_ROULETTE | _lib.RT_FLAG_COUNT_WORK
        p = ds.make_params(w, h, sc.camera.position, p2r, spp=3, seed=5, rect=(10, 20, 200, 100), flags=flags)
        a = torch.zeros(200 * 100 * 3, dtype=torch.float32, device="cuda")
        b = torch.zeros_like(a)
        st_a = ds.render_device(p, a, stream=stream)
        st_b = ds.render_device_static(p, b, stream=stream)
        torch.cuda.synchronize()
        assert same_bits(a.cpu().numpy(), b.cpu().numpy()) and float(a.max()) > 0
        assert st_a["segments"] == st_b["segments"] and st_a["inode_visits"] == st_b["inode_visits"]
        prof = ds.make_params(w, h, sc.camera.position, p2r, spp=1, rect=(0, 0, 64, 64), flags=_lib.RT_FLAG_PROFILE_KERNELS)
        with pytest.raises(_lib.RtError, match="PROFILE"):
            ds.render_device_static(prof, out, stream=stream)
        sort = ds.make_params(w, h, sc.camera.position, p2r, spp=1, rect=(0, 0, 64, 64), flags=_lib.RT_FLAG_SORT_RAYS)
        with pytest.raises(_lib.RtError, match="SORT_RAYS"):
            ds.render_device_static(sort, out, stream=stream)
    finally:
        ds.close()
