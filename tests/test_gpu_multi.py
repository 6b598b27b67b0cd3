"""rt_render_multi: one frame shared out among several scene handles from one process (the native form of the pixel-band
sharding; the reference's render() is a single call, src/render.hpp:215-234).  This box has one GPU, so the handles
sit on the same device: what is checked is the band arithmetic, the peer-copy gather and the summed counters, against
a single rt_render of the same rectangle, bit for bit."""
import numpy as np
import pytest

from helpers import same_bits

from raytracing_amd import _lib, scenes
from raytracing_amd.render import DeviceScene, RayTraceData, render_multi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("handles", [2, 3])
def test_render_multi_equals_single_render(handles):
    sc = scenes.builtin_scene(3).preprocess()
    rtd = RayTraceData(sc)
    ds = [DeviceScene(rtd) for _ in range(handles)]
    try:
        for (w, h), rect, spp in (((320, 200), (0, 0, 320, 200), 4), ((333, 131), (5, 7, 301, 119), 3), ((64, 5), (0, 0, 64, 5), 2)):
            p2r = sc.pixel_to_ray_transform(w, h)
            p = ds[0].make_params(w, h, sc.camera.position, p2r, spp=spp, seed=11, rect=rect, flags=_lib.RT_FLAG_COUNT_WORK)
            want, st_want = ds[0].render(p)
            got, st_got = render_multi(ds, p)
            assert same_bits(got, want), (w, h, rect)
            for key in ("paths", "segments", "shaded_hits", "box_tests", "tris_tested", "leaf_visits", "inode_visits"):
                assert st_got[key] == st_want[key], key
            assert st_got["box_filter_errors"] == 0
        # one handle: the plain call
        one, _ = render_multi(ds[:1], p)
        assert same_bits(one, want)
        # interleaved rows are the entry point's own device: a caller cannot ask for them on top
        bad = ds[0].make_params(320, 200, sc.camera.position, sc.pixel_to_ray_transform(320, 200), spp=1, row_group=4, row_stride=2,
                                rect=(0, 0, 320, 100))
        with pytest.raises(_lib.RtError, match="plain"):
            render_multi(ds, bad)
    finally:
        for d in ds:
            d.close()
