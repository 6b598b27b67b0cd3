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


def test_residency_probe_shrinks_the_persistent_grid(tmp_path):
    """The persistent traversal kernels deal most of their lines out statically over a grid that must be wholly resident.
    At scene creation a probe kernel with the traversal kernels' footprint counts how many blocks are alive at once; when
    that is fewer than the occupancy query promises the grids shrink.  Here the probe is handicapped with extra LDS
    (RT_AMD_PROBE_LDS), as if the system allowed only two blocks per CU: the library says so under RT_AMD_DEBUG and the
    frame is the same, bit for bit (a child process each: the probe's answer is kept per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "from raytracing_amd import scenes\n"
        "from raytracing_amd.render import DeviceScene, RayTraceData\n"
        "sc = scenes.dragon_scale_scene(30000).preprocess()\n"
        "ds = DeviceScene(RayTraceData(sc))\n"
        "img, st = ds.render(ds.make_params(640, 360, sc.camera.position, sc.pixel_to_ray_transform(640, 360), spp=4, seed=3))\n"
        "np.save(sys.argv[1], img)\n"
        "print('segments', st['segments'])\n")
    outs = []
    for name, extra in (("plain", {}), ("handicapped", {"RT_AMD_PROBE_LDS": "40000"})):
        path = tmp_path / f"{name}.npy"
        run = subprocess.run([sys.executable, "-c", code, str(path)], capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, RT_AMD_DEBUG="1", **extra))
        assert run.returncode == 0, run.stderr[-2000:]
        line = [l for l in run.stderr.splitlines() if "residency probe" in l]
        assert len(line) == 1, run.stderr[-2000:]
        outs.append((np.load(path), line[0], run.stdout))
    assert "-> 8 per CU" in outs[0][1] and "fewer than" not in outs[0][1], outs[0][1]
    assert "-> 2 per CU" in outs[1][1] and "fewer than the occupancy query promises" in outs[1][1], outs[1][1]
    assert same_bits(outs[0][0], outs[1][0]) and outs[0][2] == outs[1][2]
