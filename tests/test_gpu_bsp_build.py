"""rt_bsp_build_device: the tree of the sequential builder (and so of the reference's BSPTree constructor,
src/bsp.hpp:35-48, 174-285), array for array, with the per-level work on the GPU."""
import time

import numpy as np
import pytest

from raytracing_amd import scene as S, scenes

pytestmark = pytest.mark.gpu


def _trees(sc, **kw):
    """(host tree, device tree, seconds host, seconds device) of a scene whose meshes are instantiated."""
    sc.preprocess(**kw)
    host = sc.bsp_tree
    im = sc.instantiated_meshes
    args = (im.vertex_positions, im.vertex_begin, sc.meshes.tris, sc.meshes.tri_ranges, sc.models.meshes, sc.bounding_box,
            kw.get("max_leaf_tris", 32), kw.get("max_depth", 60))
    t0 = time.perf_counter()
    S.BSPTree(*args, 0)
    t_host = time.perf_counter() - t0
    S.BSPTree(*args, 0, 0)                     # first call on the device: module load, allocations
    t0 = time.perf_counter()
    dev = S.BSPTree(*args, 0, 0)
    return host, dev, t_host, time.perf_counter() - t0


def _same(a, b):
    assert (a.inode_count, a.leaf_count, a.ref_count, a.depth, a.oversized_leaves) == \
           (b.inode_count, b.leaf_count, b.ref_count, b.depth, b.oversized_leaves)
    assert a.root_index == b.root_index and a.root_box.tobytes() == b.root_box.tobytes()
    assert a.inodes.tobytes() == b.inodes.tobytes()
    assert np.array_equal(a.leaf_first_ref, b.leaf_first_ref)
    assert a.leaf_refs.tobytes() == b.leaf_refs.tobytes()


@pytest.mark.parametrize("name", ["builtin", "builtin_div5", "cornell", "surface100k"])
def test_device_build_equals_host_build(name):
    sc = {"builtin": lambda: scenes.builtin_scene(3), "builtin_div5": lambda: scenes.builtin_scene(5),
          "cornell": scenes.cornell_scene, "surface100k": lambda: scenes.dragon_scale_scene(100_000)}[name]()
    host, dev, _, _ = _trees(sc)
    assert host.inode_count > 0
    _same(host, dev)


def test_device_build_with_depth_cap_and_small_leaves():
    """A depth cap low enough to force oversized leaves, and a leaf size of 4 (many levels, many small leaves)."""
    sc = scenes.dragon_scale_scene(100_000)
    host, dev, _, _ = _trees(sc, max_depth=9)
    assert host.oversized_leaves > 0
    _same(host, dev)
    host, dev, _, _ = _trees(scenes.cornell_scene(), max_leaf_tris=4)
    _same(host, dev)


def test_device_build_of_the_benchmark_scene_and_of_instances():
    sc = scenes.dragon_scale_scene(900_000)
    host, dev, t_host, t_dev = _trees(sc)
    _same(host, dev)
    print(f"0.9 M triangles: host build {t_host:.2f} s, device build {t_dev:.2f} s, depth {host.depth}, "
          f"{host.ref_count} leaf references")
    sc = scenes.dragon_scale_scene(200_000, copies=3)
    host, dev, _, _ = _trees(sc)
    _same(host, dev)


def test_device_build_of_an_empty_scene_and_bad_arguments():
    from raytracing_amd import _lib
    empty = S.Scene(S.Camera((0, 0, 5), (0, np.pi, 0), 0.8), S.Meshes([]), [], S.Models(), inflate_mode=1)
    im = S.instantiate_meshes(empty.meshes.vertex_positions, empty.meshes.vertex_normals, empty.meshes.vertex_ranges, [], [])
    box = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    t = S.BSPTree(im.vertex_positions, im.vertex_begin, empty.meshes.tris, empty.meshes.tri_ranges, [], box, 32, 60, 0, 0)
    assert (t.inode_count, t.leaf_count, t.ref_count, t.root_index) == (0, 0, 0, 0)
    L = _lib.lib()
    assert L.rt_bsp_build_device(None, None, None, None, None, 0, None, 32, 60, 0, None) == -1
    assert b"rt_bsp_build_device" in L.rt_last_error()


def _check_tree_against_every_triangle(sc, max_leaf_tris=32, sample=None):
    """The reference's constructor tests every triangle of the scene against every node box (src/bsp.hpp:188-211) and
    divides a node on its 33rd triangle.  The same numbers, computed on the GPU with no candidate list, must describe
    the builder's tree: a leaf's box holds exactly the leaf's triangles, an empty child's box none, a divided node's
    box more than 32.  (The builder's lists are subsets of the true ones by construction, so equal counts mean equal
    lists.)"""
    t = sc.bsp_tree
    im = sc.instantiated_meshes
    boxes = np.concatenate([t.root_box[None], t.inodes["neg_box"], t.inodes["pos_box"]])
    index = np.concatenate([[t.root_index], t.inodes["neg_index"], t.inodes["pos_index"]]).astype(np.int64)
    if sample is not None and sample < len(boxes):          # a random subset of the node boxes
        pick = np.sort(np.random.default_rng(17).choice(len(boxes), sample, replace=False))
        boxes, index = boxes[pick], index[pick]
    t0 = time.perf_counter()
    counts = S.exhaustive_box_counts(im.vertex_positions, im.vertex_begin, sc.meshes.tris, sc.meshes.tri_ranges,
                                     sc.models.meshes, boxes).astype(np.int64)
    dt = time.perf_counter() - t0
    sizes = np.diff(t.leaf_first_ref.astype(np.int64))
    leaf = index < 0
    assert np.array_equal(counts[leaf], sizes[-index[leaf] - 1])
    assert (counts[index == 0] == 0).all()
    assert (counts[index > 0] > max_leaf_tris).all()
    return len(boxes), dt


def test_trees_hold_what_a_scan_of_every_triangle_finds():
    n, _ = _check_tree_against_every_triangle(scenes.builtin_scene(3).preprocess())
    assert n > 3
    _check_tree_against_every_triangle(scenes.cornell_scene().preprocess())
    sc = scenes.dragon_scale_scene(900_000).preprocess()
    n, dt = _check_tree_against_every_triangle(sc)
    print(f"0.9 M triangles against {n} node boxes ({n * len(sc.preprocessed_tris.tris) / 1e9:.0f} G predicate calls): {dt:.2f} s")
    # the 9.9 M-triangle scene (eleven instances, tree built on the device): 40 000 of its 2.5 M node boxes
    sc = scenes.multi_mesh_scene().preprocess(device=0, bsp_device=0)
    n, dt = _check_tree_against_every_triangle(sc, sample=40_000)
    print(f"9.9 M triangles against {n} sampled node boxes: {dt:.2f} s")
