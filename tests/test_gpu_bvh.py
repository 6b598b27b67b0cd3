"""The alternative accelerator on the GPU (RT_FLAG_ACCEL_BVH, RT_INTERSECT_BVH): the 4-wide bounding volume
hierarchy behind the same nearest-intersection query.  The reference has no such accelerator; what is checked:
  * the GPU traversal equals the oracle's restatement of it bit for bit (hits, rendered radiance, work counters);
  * its answer is the nearest triangle the reference's triangle test accepts (src/geometry.hpp:99-132): bit-equal
    to a scan of all triangles (hit point inside the root box, ties to the lowest triangle index), and bit-equal to
    the BSP path (src/bsp.hpp:50-145) on all but a few rays in a hundred thousand (of the BSP's in-leaf-box rule,
    src/bsp.hpp:82-87, the hierarchy keeps only the root box);
  * rendering through it converges to the reference's image inside the same 1/sqrt(n) envelope as the BSP path."""
import numpy as np
import pytest

from helpers import base_from_scene, golden, rel_rmse, same_bits
from oracle import binding

from raytracing_amd import _lib, render, scenes
from raytracing_amd.render import DeviceScene, RayTraceData

pytestmark = pytest.mark.gpu
NONE = 0xFFFFFFFF


def _check_hits(hits, floats, ids):
    assert np.array_equal(hits["mesh"], ids[:, 0]) and np.array_equal(hits["tri"], ids[:, 1])
    got = np.concatenate([np.stack([hits["t"], hits["u"], hits["v"]], 1), hits["point"]], 1)
    assert same_bits(got, floats)


def _with_bvh(oracle_lib, sc, w, h, max_leaf=4):
    if sc.bsp_tree is None:
        sc.preprocess()
    ds = DeviceScene(RayTraceData(sc))
    osc = oracle_lib.scene_create_with_tree(base_from_scene(sc), w, h, sc.bsp_tree)
    b = render.Bvh4(sc.preprocessed_tris.tris, max_leaf)
    ds.attach_bvh(b.nodes, b.tri_order)
    osc.set_bvh4(b.nodes, b.tri_order)
    return ds, osc, b


@pytest.fixture(scope="module")
def dragon_inputs():
    return scenes.dragon_scale_scene(900_000).preprocess()


def _rays(sc, osc, w, h, n, seed):
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.uniform(0, w, n), rng.uniform(0, h, n)], 1).astype(np.float32)
    d0 = osc.camera_rays(xy)
    o0 = np.tile(np.asarray(sc.camera.position, np.float32), (n, 1))
    f, ids = osc.intersect(o0, d0)
    hit = ids[:, 0] != NONE
    o1 = f[hit, 3:6].copy()
    d1 = rng.standard_normal(o1.shape)
    d1 = (d1 / np.linalg.norm(d1, axis=1, keepdims=True)).astype(np.float32)
    return (o0, d0), (o1, d1)


@pytest.mark.parametrize("max_leaf,shape", [(1, "0"), (4, "0"), (8, "0"), (2, "1"), (4, "1"), (8, "1")])
def test_hits_equal_restated_traversal_brute_force_and_bsp(oracle_lib, monkeypatch, max_leaf, shape):
    """Both shapes of the traversal loop (two-phase for small hierarchies, one step per iteration for large ones;
    forced here through the library's tuning variable, read when the hierarchy is attached)."""
    monkeypatch.setenv("RT_AMD_BVH_MERGED", shape)
    sc = scenes.cornell_scene()
    ds, osc, b = _with_bvh(oracle_lib, sc, 96, 64, max_leaf)
    try:
        for o, d in _rays(sc, osc, 96, 64, 6000, seed=3 + max_leaf):
            for front in (True, False):
                for t_min in (1e-3, 0.5):
                    got = ds.intersect(o, d, t_min, front, use_bvh=True)
                    f, ids = osc.intersect(o, d, t_min, front, use_bvh=True)
                    _check_hits(got, f, ids)                                   # GPU == restatement
                    fb, idb = osc.intersect_brute(o, d, t_min, front)
                    _check_hits(got, fb, idb)                                  # == scan of all triangles
                    bsp = ds.intersect(o, d, t_min, front)
                    same = (bsp["tri"] == got["tri"]) & (bsp["mesh"] == got["mesh"]) & (bsp["t"].view(np.uint32) == got["t"].view(np.uint32))
                    assert same.mean() > 0.999
    finally:
        ds.close(); osc.close()


def test_axis_parallel_lines_on_the_device(oracle_lib):
    """Direction components that are exactly zero: same answer as the restatement and as the scan of all triangles."""
    from test_bvh_host import axis_parallel_rays
    sc = scenes.cornell_scene()
    ds, osc, _ = _with_bvh(oracle_lib, sc, 64, 64)
    try:
        o, d = axis_parallel_rays(np.random.default_rng(23), 5000)
        for front in (True, False):
            got = ds.intersect(o, d, 1e-3, front, use_bvh=True)
            f, ids = osc.intersect(o, d, 1e-3, front, use_bvh=True)
            _check_hits(got, f, ids)
            fb, idb = osc.intersect_brute(o, d, 1e-3, front)
            _check_hits(got, fb, idb)
        # directions of extreme length (squares overflow; below 1e-6 / |normal| nothing can be hit at all)
        (o, d), _ = _rays(sc, osc, 64, 64, 3000, seed=5)
        for length in (1e-12, 1e12, 1e19, 3e19, 1e25):
            dl = (d.astype(np.float64) * length).astype(np.float32)
            for front in (True, False):
                got = ds.intersect(o, dl, 0.0, front, use_bvh=True)
                f, ids = osc.intersect(o, dl, 0.0, front, use_bvh=True)
                _check_hits(got, f, ids)
                fb, idb = osc.intersect_brute(o, dl, 0.0, front)
                _check_hits(got, fb, idb)
    finally:
        ds.close(); osc.close()


def _assert_same_image_and_work(img, st, oimg, ost):
    assert same_bits(img, oimg), f"{(img != oimg).sum()} of {img.size} values differ"
    assert st["segments"] == ost["segments"] and st["paths"] == ost["paths"] and st["shaded_hits"] == ost["shadedHits"]
    assert st["box_tests"] == ost["boxTests"] and st["tris_tested"] == ost["trisTested"]
    assert st["leaf_visits"] == ost["leafVisits"] and st["inode_visits"] == ost["inodeVisits"]


@pytest.mark.parametrize("bounce_limit,flags,shape", [(0, 0, "0"), (8, 0, "0"), (8, 0, "1"), (12, _lib.RT_FLAG_RUSSIAN_ROULETTE, "1")])
def test_render_bit_exact_against_restatement(oracle_lib, monkeypatch, bounce_limit, flags, shape):
    monkeypatch.setenv("RT_AMD_BVH_MERGED", shape)
    sc = scenes.cornell_scene(wall_grid=8, box_grid=3)
    ds, osc, _ = _with_bvh(oracle_lib, sc, 96, 64)
    try:
        w, h = 96, 64
        p2r = sc.pixel_to_ray_transform(w, h)
        for count in (_lib.RT_FLAG_COUNT_WORK, 0):
            p = ds.make_params(w, h, sc.camera.position, p2r, spp=4, bounce_limit=bounce_limit, seed=31,
                               flags=flags | count | _lib.RT_FLAG_ACCEL_BVH)
            img, st = ds.render(p)
            oflags = binding.FLAG_ACCEL_BVH | (binding.FLAG_RUSSIAN_ROULETTE if flags & _lib.RT_FLAG_RUSSIAN_ROULETTE else 0)
            oimg, ost = osc.render_device_mode(osc.render_params(4, seed=31, bounce_limit=bounce_limit, flags=oflags), threads=16)
            if count:
                _assert_same_image_and_work(img, st, oimg, ost)
            else:
                assert same_bits(img, oimg) and st["segments"] == ost["segments"]
        # against the BSP path: the same image except for the few paths whose hit differs
        bsp, sb = ds.render(ds.make_params(w, h, sc.camera.position, p2r, spp=4, bounce_limit=bounce_limit, seed=31, flags=flags))
        differing = (bsp != img).any(-1).mean()
        assert differing < 0.01, differing
    finally:
        ds.close(); osc.close()


def test_flag_without_hierarchy_is_an_error_and_attach_validates(oracle_lib):
    sc = scenes.builtin_scene(3).preprocess()
    ds = DeviceScene(RayTraceData(sc))
    try:
        p = ds.make_params(64, 48, sc.camera.position, sc.pixel_to_ray_transform(64, 48), spp=1, flags=_lib.RT_FLAG_ACCEL_BVH)
        with pytest.raises(_lib.RtError, match="hierarchy"):
            ds.render(p)
        with pytest.raises(_lib.RtError, match="hierarchy"):
            ds.intersect(np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32), use_bvh=True)
        b = render.Bvh4(sc.preprocessed_tris.tris, 4)
        bad = b.nodes.copy()
        interior = np.argwhere((bad["child"] & 0x80000000) == 0)
        i, k = interior[0]
        bad["child"][i, k] = i                             # a node that is its own child
        with pytest.raises(_lib.RtError, match="child number"):
            ds.attach_bvh(bad, b.tri_order)
        order = b.tri_order.copy()
        order[0] = order[1]
        with pytest.raises(_lib.RtError, match="permutation"):
            ds.attach_bvh(b.nodes, order)
        with pytest.raises(_lib.RtError):
            ds.attach_bvh(b.nodes[:-1], b.tri_order)       # a child out of range, or a triangle in no leaf
        ds.build_bvh(4)
        img, _ = ds.render(p)
        ref, _ = ds.render(ds.make_params(64, 48, sc.camera.position, sc.pixel_to_ray_transform(64, 48), spp=1))
        assert (img != ref).any(-1).mean() < 0.01
    finally:
        ds.close()


def test_empty_scene_and_single_triangle_through_the_hierarchy():
    from raytracing_amd import scene as S
    m = S.Models()
    empty = S.Scene(S.Camera((0, 0, 5), (0, np.pi, 0), 0.8), S.Meshes([]), [], m, inflate_mode=1)
    im = S.instantiate_meshes(empty.meshes.vertex_positions, empty.meshes.vertex_normals, empty.meshes.vertex_ranges, [], [])
    empty.instantiated_meshes = im
    empty.preprocessed_tris = S.preprocess_tris(im.vertex_positions, im.vertex_begin, empty.meshes.tris,
                                                empty.meshes.tri_ranges, [])
    empty.preprocessed_materials = np.zeros((0, 12), np.float32)
    empty.bounding_box = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    empty.bsp_tree = S.BSPTree(im.vertex_positions, im.vertex_begin, empty.meshes.tris, empty.meshes.tri_ranges, [],
                               empty.bounding_box)
    ds = DeviceScene(RayTraceData(empty))
    try:
        ds.build_bvh(4)
        p = ds.make_params(32, 16, empty.camera.position, empty.pixel_to_ray_transform(32, 16), spp=2, flags=_lib.RT_FLAG_ACCEL_BVH)
        img, st = ds.render(p)
        assert not img.any() and st["segments"] == st["paths"] == 32 * 16 * 2
    finally:
        ds.close()
    tri = (np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32), np.tile(np.array([[0, 0, 1]], np.float32), (3, 1)),
           np.array([[0, 1, 2]], np.uint32))
    m = S.Models()
    m.add(S.MeshTransform(), 0, 0)
    one = S.Scene(S.Camera((0, 0, 5), (0, np.pi, 0), 0.8), S.Meshes([tri]), [S.Material((0.5,) * 3, 0.5, 0.0, (2, 3, 4))],
                  m, inflate_mode=1).preprocess()
    ds = DeviceScene(RayTraceData(one))
    try:
        ds.build_bvh(4)
        p2r = one.pixel_to_ray_transform(64, 64)
        a, _ = ds.render(ds.make_params(64, 64, one.camera.position, p2r, spp=2, flags=_lib.RT_FLAG_ACCEL_BVH))
        b, _ = ds.render(ds.make_params(64, 64, one.camera.position, p2r, spp=2))
        assert same_bits(a, b) and np.allclose(a[32, 32], [2, 3, 4]) and not a[0, 0].any()
    finally:
        ds.close()


def test_converged_image_through_the_hierarchy():
    """Same envelope as the BSP path's convergence test: relative RMSE against the reference's 16384-spp render of the
    built-in scene below 1.25 * 1.17 * sqrt(1/n1 + 1/n2) = 0.026 at 4096 spp, mean radiance within 0.5 %."""
    sc = scenes.builtin_scene(3).preprocess()
    ds = DeviceScene(RayTraceData(sc))
    try:
        ds.build_bvh(4)
        g = golden("builtin_converged_160x120.npz")
        w, h = 160, 120
        p2r = sc.pixel_to_ray_transform(w, h)
        img, _ = ds.render(ds.make_params(w, h, sc.camera.position, p2r, spp=4096, seed=909, flags=_lib.RT_FLAG_ACCEL_BVH))
        ref = g["image"]
        assert rel_rmse(img, ref) < 0.026, rel_rmse(img, ref)
        assert abs(img.mean() / ref.mean() - 1.0) < 0.005
    finally:
        ds.close()


@pytest.mark.parametrize("shape", ["1", "0"])
def test_dragon_scale_scene_through_the_hierarchy(oracle_lib, dragon_inputs, monkeypatch, shape):
    """0.9 M triangles (both loop shapes; the one-step shape is what a hierarchy of this size gets by default): hierarchy deeper than the 20 stack entries kept in LDS (the scratch part of the stack is
    exercised), GPU against the restatement on 60 000 + 60 000 rays and a rendered tile, and against the BSP path."""
    monkeypatch.setenv("RT_AMD_BVH_MERGED", shape)
    sc = dragon_inputs
    ds, osc, b = _with_bvh(oracle_lib, sc, 1920, 1080)
    try:
        assert b.stack_need > 20
        w, h = 1920, 1080
        (o0, d0), (o1, d1) = _rays(sc, osc, w, h, 60_000, seed=8)
        for o, d in ((o0, d0), (o1, d1)):
            got = ds.intersect(o, d, use_bvh=True)
            f, ids = osc.intersect(o, d, use_bvh=True)
            _check_hits(got, f, ids)
            bsp = ds.intersect(o, d)
            same = (bsp["tri"] == got["tri"]) & (bsp["mesh"] == got["mesh"]) & (bsp["t"].view(np.uint32) == got["t"].view(np.uint32))
            assert same.mean() > 0.9999, same.mean()
        rect = (900, 500, 96, 64)
        p2r = sc.pixel_to_ray_transform(w, h)
        tile, st = ds.render(ds.make_params(w, h, sc.camera.position, p2r, spp=2, rect=rect,
                                            flags=_lib.RT_FLAG_COUNT_WORK | _lib.RT_FLAG_ACCEL_BVH))
        otile, ost = osc.render_device_mode(osc.render_params(2, rect=rect, flags=binding.FLAG_ACCEL_BVH), threads=16)
        _assert_same_image_and_work(tile, st, otile, ost)
        # full frame at 1 spp: halves reassemble, and the frame differs from the BSP path's in a few pixels at most
        full, sf = ds.render(ds.make_params(w, h, sc.camera.position, p2r, spp=1, flags=_lib.RT_FLAG_ACCEL_BVH))
        left, _ = ds.render(ds.make_params(w, h, sc.camera.position, p2r, spp=1, rect=(0, 0, 1000, h), flags=_lib.RT_FLAG_ACCEL_BVH))
        right, _ = ds.render(ds.make_params(w, h, sc.camera.position, p2r, spp=1, rect=(1000, 0, w - 1000, h), flags=_lib.RT_FLAG_ACCEL_BVH))
        assert same_bits(np.concatenate([left, right], axis=1), full)
        bsp, sb = ds.render(ds.make_params(w, h, sc.camera.position, p2r, spp=1))
        assert (bsp != full).any(-1).mean() < 1e-3
        print("segments bvh/bsp", sf["segments"], sb["segments"])
    finally:
        ds.close(); osc.close()


@pytest.mark.parametrize("scale,offset", [(1.0, (0, 0, 0)), (1e-3, (0, 0, 0)), (0.05, (0, 0, 0)), (3e3, (0, 0, 0)),
                                          (1.0, (500, -200, 1000)), (0.02, (7, 7, 7))])
def test_random_soups_at_many_scales(oracle_lib, scale, offset):
    """Random triangle soups from millimetres to kilometres and far from the origin: the padding of the stored boxes
    and the widening of the slab interval scale with the coordinates, so the hierarchy must still return exactly what
    a scan of all triangles returns (any box lost to rounding would show up as a missing hit), for lines that start
    inside and outside the scene."""
    from raytracing_amd import scene as S
    rng = np.random.default_rng(int(scale * 1000) + int(offset[0]))
    n = 2500
    centres = rng.uniform(-4, 4, (n, 1, 3))
    pos = ((centres + rng.uniform(-0.3, 0.3, (n, 3, 3))).reshape(-1, 3) * scale).astype(np.float32)
    nrm = rng.standard_normal((len(pos), 3))
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    tris = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    m = S.Models()
    off = np.array(offset, np.float64) * scale
    m.add(S.MeshTransform(tuple(off), S.IDENTITY_QUAT, (1, 1, 1)), 0, 0)
    m.add(S.MeshTransform(tuple(off + np.array([0.3, 0.1, -0.2]) * scale), (0.2, 0.4, 0.1), (1.0, 0.7, 1.2)), 0, 1)
    cam = S.Camera(tuple(off + np.array([0, 0, 14.0]) * scale), (0, np.pi, 0), 0.7)
    mats = [S.Material((0.6, 0.5, 0.4), 0.4, 0.2, (0.5, 0.4, 0.3)), S.Material((0.9, 0.9, 0.9), 0.1, 0.9)]
    sc = S.Scene(cam, S.Meshes([(pos, nrm, tris)]), mats, m, inflate_mode=1)
    ds, osc, _ = _with_bvh(oracle_lib, sc, 96, 64)
    try:
        box = sc.bounding_box
        o = rng.uniform(box[:3] - 0.2 * (box[3:] - box[:3]), box[3:] + 0.2 * (box[3:] - box[:3]), (20000, 3)).astype(np.float32)
        d = rng.standard_normal((20000, 3))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        for front in (True, False):
            got = ds.intersect(o, d, 1e-3 * scale, front, use_bvh=True)
            f, ids = osc.intersect_brute(o, d, 1e-3 * scale, front)
            _check_hits(got, f, ids)
            f, ids = osc.intersect(o, d, 1e-3 * scale, front, use_bvh=True)
            _check_hits(got, f, ids)
        assert (ids[:, 0] != NONE).mean() > (0.05 if scale >= 0.02 else -1)
        w, h = 96, 64
        p = ds.make_params(w, h, sc.camera.position, sc.pixel_to_ray_transform(w, h), spp=3, seed=9,
                           flags=_lib.RT_FLAG_COUNT_WORK | _lib.RT_FLAG_ACCEL_BVH)
        img, st = ds.render(p)
        oimg, ost = osc.render_device_mode(osc.render_params(3, seed=9, flags=binding.FLAG_ACCEL_BVH), threads=16)
        _assert_same_image_and_work(img, st, oimg, ost)
    finally:
        ds.close(); osc.close()


def test_hierarchy_on_a_scene_received_as_bytes(oracle_lib):
    """A process that got the scene as a byte buffer (one process per GPU) builds the hierarchy from those bytes."""
    from raytracing_amd.render import serialize_scene
    sc = scenes.cornell_scene(wall_grid=8, box_grid=3).preprocess()
    rtd = RayTraceData(sc)
    a, b = DeviceScene(rtd), DeviceScene.from_blob(serialize_scene(rtd))
    try:
        a.build_bvh(4); b.build_bvh(4)
        p2r = sc.pixel_to_ray_transform(96, 64)
        ia, _ = a.render(a.make_params(96, 64, sc.camera.position, p2r, spp=2, flags=_lib.RT_FLAG_ACCEL_BVH))
        ib, _ = b.render(b.make_params(96, 64, sc.camera.position, p2r, spp=2, flags=_lib.RT_FLAG_ACCEL_BVH))
        assert same_bits(ia, ib) and ia.max() > 0
    finally:
        a.close(); b.close()
