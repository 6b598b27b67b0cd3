"""Shared test plumbing: golden-vector loading and conversions between the product's host-side scene
objects and the oracle's flat BaseScene."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def base_from_golden(g, inflate_mode=None):
    from oracle.binding import BaseScene
    im = int(g["base_inflate_mode"][0]) if inflate_mode is None else inflate_mode
    return BaseScene(g["base_positions"], g["base_normals"], g["base_tris"], g["base_mesh_vertex_ranges"],
                     g["base_mesh_tri_ranges"], g["base_materials"], g["base_transforms"], g["base_model_meshes"],
                     g["base_model_materials"], g["base_camera"], inflate_mode=im)


def base_from_scene(scene):
    """Oracle BaseScene with exactly the inputs of a raytracing_amd.scene.Scene."""
    from oracle.binding import BaseScene
    cam = np.concatenate([scene.camera.position, scene.camera.orientation, [scene.camera.fov]]).astype(np.float32)
    return BaseScene(scene.meshes.vertex_positions, scene.meshes.vertex_normals, scene.meshes.tris,
                     scene.meshes.vertex_ranges, scene.meshes.tri_ranges,
                     np.stack([m.packed() for m in scene.materials]),
                     np.stack([t.packed() for t in scene.models.mesh_transforms]),
                     scene.models.meshes, scene.models.materials, cam, inflate_mode=scene.inflate_mode)


def scene_from_golden(g):
    """raytracing_amd Scene from a golden base description (un-preprocessed)."""
    from raytracing_amd.scene import Camera, Material, Meshes, MeshTransform, Models, Scene
    meshes = []
    for (vb, vs), (tb, ts) in zip(g["base_mesh_vertex_ranges"], g["base_mesh_tri_ranges"]):
        meshes.append((g["base_positions"][vb:vb + vs], g["base_normals"][vb:vb + vs], g["base_tris"][tb:tb + ts]))
    mats = [Material(m[0:3], m[3], m[4], m[5:8]) for m in g["base_materials"]]
    models = Models()
    for t, mesh, mat in zip(g["base_transforms"], g["base_model_meshes"], g["base_model_materials"]):
        models.add(MeshTransform(t[0:3], t[3:7], t[7:10]), int(mesh), int(mat))
    c = g["base_camera"]
    return Scene(Camera(c[0:3], c[3:7], c[7]), Meshes(meshes), mats, models,
                 inflate_mode=int(g["base_inflate_mode"][0]))


def tree_arrays(bsp_tree):
    """BSPTree arrays in the layout of the oracle's tree dump."""
    t = bsp_tree
    boxes = np.concatenate([t.inodes["neg_box"], t.inodes["pos_box"]], 1).astype(np.float32).reshape(-1, 12)
    ints = np.stack([t.inodes["neg_index"], t.inodes["pos_index"], t.inodes["axis"].astype(np.int32)], 1) \
        if t.inode_count else np.zeros((0, 3), np.int32)
    counts = np.diff(t.leaf_first_ref).astype(np.uint32)
    refs = np.full((t.leaf_count, 32, 2), 0xFFFFFFFF, np.uint32)
    for leaf in range(t.leaf_count):
        r = t.leaf_refs[t.leaf_first_ref[leaf]:t.leaf_first_ref[leaf + 1]]
        refs[leaf, :len(r), 0] = r["mesh"]
        refs[leaf, :len(r), 1] = r["tri"]
    return dict(root_box=t.root_box, root_index=np.array([t.root_index], np.int32), inode_boxes=boxes,
                inode_ints=ints.astype(np.int32), leaf_counts=counts, leaf_refs=refs)


def rel_rmse(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
