"""Host-side logic of librt_amd.so (include/rt_host.h) against the oracle and the golden vectors: scene
preprocessing, predicates, and the candidate-list BSP builder.  CPU only; no kernel is launched."""
import numpy as np
import pytest

from helpers import base_from_scene, golden, same_bits, scene_from_golden, tree_arrays

from raytracing_amd import scene as S
from raytracing_amd import scenes
from raytracing_amd.render import RayTraceData


def test_builtin_scene_description_matches_reference():
    """scenes.builtin_scene() restates the data of src/main.cpp:134-187; the golden copy came from the
    reference's own Scene object."""
    g = golden("builtin_scene.npz")
    sc = scenes.builtin_scene(3)
    assert same_bits(sc.meshes.vertex_positions, g["base_positions"])
    assert same_bits(sc.meshes.vertex_normals, g["base_normals"])
    assert np.array_equal(sc.meshes.tris, g["base_tris"])
    assert same_bits(np.stack([m.packed() for m in sc.materials]), g["base_materials"])
    assert same_bits(np.stack([t.packed() for t in sc.models.mesh_transforms]), g["base_transforms"])
    cam = np.concatenate([sc.camera.position, sc.camera.orientation, [sc.camera.fov]]).astype(np.float32)
    assert same_bits(cam, g["base_camera"])
    assert np.array_equal(np.asarray(sc.models.meshes, np.uint32), g["base_model_meshes"])
    assert np.array_equal(np.asarray(sc.models.materials, np.uint32), g["base_model_materials"])


@pytest.mark.parametrize("name,size", [("builtin_scene.npz", (640, 480)), ("builtin_div5_scene.npz", (320, 240))])
def test_preprocessing_and_tree_match_golden(name, size):
    g = golden(name)
    sc = scene_from_golden(g).preprocess()
    assert same_bits(sc.instantiated_meshes.vertex_positions, g["pre_positions"])
    assert same_bits(sc.instantiated_meshes.vertex_normals, g["pre_normals"])
    assert np.array_equal(sc.instantiated_meshes.vertex_begin[:-1], g["pre_vertex_ranges"][:, 0])
    assert same_bits(sc.preprocessed_tris.tris, g["pre_pretris"])
    assert np.array_equal(sc.preprocessed_tris.tri_begin[:-1], g["pre_tri_ranges"][:, 0])
    assert same_bits(RayTraceData(sc).materials, g["pre_model_materials"])
    assert same_bits(sc.bounding_box, g["pre_root_box"])
    assert same_bits(sc.pixel_to_ray_transform(*size), g["pre_pixel_to_ray"])
    for k, v in tree_arrays(sc.bsp_tree).items():
        assert same_bits(v, g["bsp_" + k]), k


def test_predicates_match_golden():
    g = golden("predicates.npz")
    lb = [S.line_intersects_box(o, d, b) for o, d, b in zip(g["line_origins"], g["line_directions"], g["boxes"])]
    assert np.array_equal(np.array(lb, np.uint8), g["line_box"])
    tb = [S.tri_intersects_box(t, b) for t, b in zip(g["tris"], g["boxes"])]
    assert np.array_equal(np.array(tb, np.uint8), g["tri_box"])


def _compare_with_oracle_tree(oracle_lib, sc, w=64, h=48):
    sc.preprocess()
    o = oracle_lib.scene_create(base_from_scene(sc), w, h)
    try:
        pre = o.preprocessed()
        assert same_bits(sc.instantiated_meshes.vertex_positions, pre["positions"])
        assert same_bits(sc.instantiated_meshes.vertex_normals, pre["normals"])
        assert same_bits(sc.preprocessed_tris.tris, pre["pretris"])
        assert same_bits(sc.bounding_box, pre["root_box"])
        assert same_bits(sc.pixel_to_ray_transform(w, h), pre["pixel_to_ray"])
        ob = o.bsp()
        for k, v in tree_arrays(sc.bsp_tree).items():
            assert same_bits(v, ob[k]), k
    finally:
        o.close()


def test_builder_equals_reference_algorithm_on_cornell(oracle_lib):
    """The oracle rescans every triangle at every node like the reference (src/bsp.hpp:188-211); the
    product's builder filters candidate lists.  Same tree, same leaf order."""
    sc = scenes.cornell_scene()
    _compare_with_oracle_tree(oracle_lib, sc)
    assert 3500 < len(sc.preprocessed_tris.tris) < 4500


def test_builder_equals_reference_algorithm_on_displaced_surface(oracle_lib):
    sc = scenes.dragon_scale_scene(target_tris=6000)
    _compare_with_oracle_tree(oracle_lib, sc)
    assert sc.bsp_tree.oversized_leaves == 0


def test_builder_equals_reference_algorithm_on_triangle_soup(oracle_lib):
    """Random small triangles: no structure for the candidate filter to exploit."""
    rng = np.random.default_rng(3)
    n = 3000
    centres = rng.uniform(-4, 4, (n, 1, 3))
    pos = (centres + rng.uniform(-0.25, 0.25, (n, 3, 3))).reshape(-1, 3).astype(np.float32)
    nrm = np.tile(np.array([[0, 1, 0]], np.float32), (len(pos), 1))
    tris = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    m = S.Models()
    m.add(S.MeshTransform(), 0, 0)
    m.add(S.MeshTransform((0.3, 0.1, -0.2), (0.2, 0.4, 0.1), (1.0, 0.7, 1.2)), 0, 0)
    sc = S.Scene(S.Camera((0, 0, 12), (0, np.pi, 0), 0.8), S.Meshes([(pos, nrm, tris)]),
                 [S.Material((0.5, 0.5, 0.5), 0.5, 0.0)], m, inflate_mode=1)
    _compare_with_oracle_tree(oracle_lib, sc)


def test_builder_depth_cap_makes_oversized_leaf():
    """40 coincident triangles can never be separated: the reference would recurse forever
    (src/bsp.hpp:276-284 has no guard); the builder stops at max_depth with one oversized leaf."""
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1]], np.float32)
    pos = np.tile(tri, (40, 1))
    nrm = np.tile(np.array([[0, 1, 0]], np.float32), (len(pos), 1))
    tris = np.arange(120, dtype=np.uint32).reshape(-1, 3)
    m = S.Models()
    m.add(S.MeshTransform(), 0, 0)
    sc = S.Scene(S.Camera((0, 3, 0), (1.5, 0, 0), 0.8), S.Meshes([(pos, nrm, tris)]),
                 [S.Material((0.5, 0.5, 0.5), 0.5, 0.0)], m, inflate_mode=1)
    sc.preprocess(max_depth=12)
    t = sc.bsp_tree
    assert t.depth == 12 and t.oversized_leaves >= 1
    assert np.diff(t.leaf_first_ref).max() == 40


def test_empty_scene_builds_empty_tree():
    m = S.Models()
    sc = S.Scene(S.Camera((0, 0, 0), S.IDENTITY_QUAT, 0.8), S.Meshes([]), [], m, inflate_mode=1)
    im = S.instantiate_meshes(sc.meshes.vertex_positions, sc.meshes.vertex_normals, sc.meshes.vertex_ranges, [], [])
    assert len(im.vertex_positions) == 0 and list(im.vertex_begin) == [0]
    t = S.BSPTree(im.vertex_positions, im.vertex_begin, sc.meshes.tris, sc.meshes.tri_ranges, [],
                  np.array([-1, -1, -1, 1, 1, 1], np.float32))
    assert (t.root_index, t.inode_count, t.leaf_count, t.ref_count) == (0, 0, 0, 0)


def test_material_and_colour_helpers():
    from raytracing_amd import _lib
    out = S.preprocess_material(S.Material((0.25, 0.5, 1.0), 0.5, 0.25, (1, 2, 3)))
    assert out[0] == np.float32(0.0625) and out[1] == np.float32(0.015625)
    assert np.allclose(out[2:5], 0.75 * 0.04 + 0.25 * np.array([0.25, 0.5, 1.0]))
    assert np.allclose(out[5:8], 0.75 * np.array([0.25, 0.5, 1.0]) / np.pi)
    assert list(out[8:11]) == [1, 2, 3]
    L = _lib.lib()
    for v in (0.0, 0.02, 0.04045, 0.2, 0.5, 1.0):
        assert abs(L.rt_linear_to_srgb(L.rt_srgb_to_linear(v)) - v) < 1e-6


def test_oracle_traverses_supplied_tree_like_its_own(oracle_lib):
    """orc_scene_create_with_tree (used for scenes too large for the reference's tree build) gives the
    same hits and radiance as the oracle's own reference-faithful build of the same scene."""
    sc = scenes.cornell_scene(wall_grid=8, box_grid=3).preprocess()
    base = base_from_scene(sc)
    own = oracle_lib.scene_create(base, 48, 32)
    given = oracle_lib.scene_create_with_tree(base, 48, 32, sc.bsp_tree)
    try:
        a, b = own.bsp(), given.bsp()
        for k in a:
            assert same_bits(a[k], b[k]), k
        rng = np.random.default_rng(11)
        o = rng.uniform(-4.5, 4.5, (4000, 3)).astype(np.float32) + np.array([0, 5, 0], np.float32)
        d = rng.standard_normal((4000, 3))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        fa, ia = own.intersect(o, d)
        fb, ib = given.intersect(o, d)
        assert same_bits(fa, fb) and np.array_equal(ia, ib)
        pa, pb = own.render_params(2), given.render_params(2)
        assert same_bits(own.render_device_mode(pa, threads=4)[0], given.render_device_mode(pb, threads=4)[0])
    finally:
        own.close(); given.close()


def test_ppm_writer_matches_reference_file_layout(tmp_path):
    """src/main.cpp:272-280: 'P6', width and height separated by a space, '255', then the raw bytes."""
    from raytracing_amd import image
    rgb = (np.arange(5 * 7 * 3) % 251).astype(np.uint8).reshape(5, 7, 3)
    path = tmp_path / "out.ppm"
    image.write_ppm(path, rgb)
    assert path.read_bytes() == b"P6\n7 5\n255\n" + rgb.tobytes()
    with pytest.raises(OSError):
        image.write_ppm(tmp_path / "missing_dir" / "x.ppm", rgb)


@pytest.mark.parametrize("threads", [2, 5, 8])
def test_threaded_build_restores_sequential_order(threads):
    """Subtrees built in parallel and spliced back must give the arrays of the single-threaded build, bit
    for bit: interior nodes in pre-order, leaves in creation order, references in ascending order."""
    sc = scenes.dragon_scale_scene(target_tris=260_000)
    one = sc.preprocess(threads=1).bsp_tree
    a = tree_arrays(one)
    many = sc.preprocess(threads=threads).bsp_tree
    b = tree_arrays(many)
    assert (one.depth, one.inode_count, one.leaf_count, one.ref_count) == (many.depth, many.inode_count, many.leaf_count, many.ref_count)
    for k in a:
        assert same_bits(a[k], b[k]), k
