"""Host-side mirror of the reference's scene interface.

Names follow the reference (src/scene.hpp, src/mesh.hpp, src/camera.hpp, src/material.hpp, src/bsp.hpp)
with snake_case attributes; the arithmetic lives in librt_amd.so (include/rt_host.h) so that a
scene prepared here is bit-identical to one prepared by the reference.  Index types are 32-bit
(the reference's are 16-bit, src/index_types.hpp:8-11).
"""
import ctypes
import numpy as np

from . import _lib
from ._lib import c_u32, c_f32, c_int, ptr


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


def _u32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a.reshape(shape) if shape is not None else a


def quat_from_euler(euler):
    """glm::quat{glm::vec3 eulerAngles} as used at src/main.cpp:136,146-147; returns (w, x, y, z)."""
    e = _f32(euler, (3,))
    q = np.zeros(4, np.float32)
    _lib.lib().rt_quat_from_euler(ptr(e), ptr(q))
    return q


IDENTITY_QUAT = np.array([1.0, 0.0, 0.0, 0.0], np.float32)


class Camera:
    """src/camera.hpp:13-29.  `orientation` is a quaternion (w, x, y, z) or 3 Euler angles."""

    def __init__(self, position, orientation, fov):
        self.position = _f32(position, (3,))
        o = np.asarray(orientation, np.float32).ravel()
        self.orientation = quat_from_euler(o) if o.size == 3 else _f32(o, (4,))
        self.fov = np.float32(fov)

    def _basis(self):
        f, d, r = (np.zeros(3, np.float32) for _ in range(3))
        _lib.lib().rt_camera_basis(ptr(self.orientation), ptr(f), ptr(d), ptr(r))
        return f, d, r

    def forward(self):
        return self._basis()[0]

    def down(self):
        return self._basis()[1]

    def right(self):
        return self._basis()[2]


def pixel_to_ray_transform(forward, down, right, fov, image_width, image_height):
    """src/camera.hpp:33-46.  Returns the mat3 as 9 floats, column-major {right, down, o}."""
    out = np.zeros(9, np.float32)
    _lib.lib().rt_pixel_to_ray_transform(ptr(_f32(forward)), ptr(_f32(down)), ptr(_f32(right)), c_f32(float(fov)),
                                         c_u32(image_width), c_u32(image_height), ptr(out))
    return out


class Material:
    """src/material.hpp:14-19."""

    def __init__(self, colour, roughness, metalness, emission=(0.0, 0.0, 0.0)):
        self.colour = _f32(colour, (3,))
        self.roughness = np.float32(roughness)
        self.metalness = np.float32(metalness)
        self.emission = _f32(emission, (3,))

    def packed(self):
        return np.concatenate([self.colour, [self.roughness, self.metalness], self.emission]).astype(np.float32)


def preprocess_material(material):
    """src/material.hpp:32-45 -> 12 floats {ndfAlphaSq, geometryAlphaSq, f0, adjustedColour, emission, 0}."""
    m = material.packed() if isinstance(material, Material) else _f32(material, (8,))
    out = np.zeros(12, np.float32)
    _lib.lib().rt_preprocess_material(ptr(m), ptr(out))
    return out


class MeshTransform:
    """src/mesh.hpp:57-70."""

    def __init__(self, position=(0, 0, 0), orientation=IDENTITY_QUAT, scale=(1, 1, 1)):
        self.position = _f32(position, (3,))
        o = np.asarray(orientation, np.float32).ravel()
        self.orientation = quat_from_euler(o) if o.size == 3 else _f32(o, (4,))
        self.scale = _f32(np.broadcast_to(np.asarray(scale, np.float32), (3,)), (3,))

    def packed(self):
        return np.concatenate([self.position, self.orientation, self.scale]).astype(np.float32)


class Meshes:
    """src/scene.hpp:17-26, src/scene.cpp:8-30: base meshes concatenated as structure-of-arrays."""

    def __init__(self, meshes):
        pos, nrm, tris, vranges, tranges = [], [], [], [], []
        voff = toff = 0
        for positions, normals, mesh_tris in meshes:
            p, n, t = _f32(positions, (-1, 3)), _f32(normals, (-1, 3)), _u32(mesh_tris, (-1, 3))
            assert len(p) == len(n)
            assert t.size == 0 or int(t.max()) < len(p), "triangle index outside the mesh's vertex range"
            pos.append(p); nrm.append(n); tris.append(t)
            vranges.append((voff, len(p))); tranges.append((toff, len(t)))
            voff += len(p); toff += len(t)
        self.vertex_positions = np.concatenate(pos) if pos else np.zeros((0, 3), np.float32)
        self.vertex_normals = np.concatenate(nrm) if nrm else np.zeros((0, 3), np.float32)
        self.tris = np.concatenate(tris) if tris else np.zeros((0, 3), np.uint32)
        self.vertex_ranges = _u32(vranges, (-1, 2))
        self.tri_ranges = _u32(tranges, (-1, 2))


def load_mesh(path):
    """One Meshes entry (positions, normals, triangles) from a Wavefront OBJ or PLY file, read by
    rt_mesh_load in librt_amd.so.  The reference itself only has literal meshes (src/main.cpp:25-121)."""
    L = _lib.lib()
    handle = ctypes.c_void_p()
    if L.rt_mesh_load(str(path).encode(), ctypes.byref(handle)) != 0:
        raise ValueError(L.rt_mesh_last_error().decode())
    try:
        nv, nt = ctypes.c_uint32(), ctypes.c_uint32()
        L.rt_mesh_counts(handle, ctypes.byref(nv), ctypes.byref(nt))
        pp, pn, pt = ctypes.POINTER(ctypes.c_float)(), ctypes.POINTER(ctypes.c_float)(), ctypes.POINTER(ctypes.c_uint32)()
        L.rt_mesh_view(handle, ctypes.byref(pp), ctypes.byref(pn), ctypes.byref(pt))
        def copy(ptr, count, dtype):
            if count == 0:
                return np.zeros((0, 3), dtype)
            return np.ctypeslib.as_array(ptr, shape=(count, 3)).astype(dtype, copy=True)
        return copy(pp, nv.value, np.float32), copy(pn, nv.value, np.float32), copy(pt, nt.value, np.uint32)
    finally:
        L.rt_mesh_destroy(handle)


class Models:
    """Scene::Models, src/scene.hpp:33-37."""

    def __init__(self, mesh_transforms=(), meshes=(), materials=()):
        self.mesh_transforms = list(mesh_transforms)
        self.meshes = list(meshes)
        self.materials = list(materials)

    def add(self, transform, mesh, material):
        self.mesh_transforms.append(transform)
        self.meshes.append(mesh)
        self.materials.append(material)


class InstantiatedMeshes:
    """src/mesh.hpp:79-83; vertex_begin has one extra trailing entry."""

    def __init__(self, positions, normals, vertex_begin):
        self.vertex_positions, self.vertex_normals, self.vertex_begin = positions, normals, vertex_begin


def instantiate_meshes(vertex_positions, vertex_normals, vertex_ranges, instance_transforms, instance_meshes, device=None):
    """src/mesh.hpp:90-138.  With `device` set the work runs on that GPU (rt_instantiate_meshes_device)."""
    tr = _f32(np.stack([t.packed() for t in instance_transforms]) if len(instance_transforms) else
              np.zeros((0, 10)), (-1, 10))
    mm = _u32(instance_meshes)
    vr = _u32(vertex_ranges, (-1, 2))
    total = int(vr[mm, 1].sum()) if len(mm) else 0
    pos = np.zeros((total, 3), np.float32)
    nrm = np.zeros((total, 3), np.float32)
    begin = np.zeros(len(mm) + 1, np.uint32)
    if device is None:
        _lib.lib().rt_instantiate_meshes(ptr(_f32(vertex_positions)), ptr(_f32(vertex_normals)), ptr(vr), ptr(tr),
                                         ptr(mm), c_u32(len(mm)), ptr(pos), ptr(nrm), ptr(begin))
    else:
        _lib.check(_lib.lib().rt_instantiate_meshes_device(c_int(device), ptr(_f32(vertex_positions)), ptr(_f32(vertex_normals)),
                                                           ptr(vr), c_u32(len(vr)), ptr(tr), ptr(mm), c_u32(len(mm)),
                                                           ptr(pos), ptr(nrm), ptr(begin)))
    return InstantiatedMeshes(pos, nrm, begin)


class PreprocessedTris:
    """src/mesh.hpp:141-144; tri_begin has one extra trailing entry."""

    def __init__(self, tris, tri_begin):
        self.tris, self.tri_begin = tris, tri_begin


def preprocess_tris(inst_positions, vertex_begin, tris, tri_ranges, instance_meshes, device=None):
    """src/mesh.hpp:149-188: per model, {normal, v1, v1ToV3, v1ToV2} for each triangle.  With `device` set the
    work runs on that GPU (rt_preprocess_tris_device)."""
    mm = _u32(instance_meshes)
    trr = _u32(tri_ranges, (-1, 2))
    total = int(trr[mm, 1].sum()) if len(mm) else 0
    out = np.zeros((total, 12), np.float32)
    begin = np.zeros(len(mm) + 1, np.uint32)
    if device is None:
        _lib.lib().rt_preprocess_tris(ptr(_f32(inst_positions)), ptr(_u32(vertex_begin)), ptr(_u32(tris)), ptr(trr),
                                      ptr(mm), c_u32(len(mm)), ptr(out), ptr(begin))
    else:
        _lib.check(_lib.lib().rt_preprocess_tris_device(c_int(device), ptr(_f32(inst_positions)), ptr(_u32(vertex_begin)),
                                                        ptr(_u32(tris)), ptr(trr), c_u32(len(trr)), ptr(mm), c_u32(len(mm)),
                                                        ptr(out), ptr(begin)))
    return PreprocessedTris(out, begin)


def scene_bounding_box(inst_positions, inflate_mode=0):
    """computeBoundingBox (src/geometry.hpp:85-92) + the inflation of src/main.cpp:207-208."""
    p = _f32(inst_positions, (-1, 3))
    box = np.zeros(6, np.float32)
    _lib.lib().rt_scene_bounding_box(ptr(p), _lib.c_u64(len(p)), c_int(inflate_mode), ptr(box))
    return box


def line_intersects_box(origin, direction, box):
    return bool(_lib.lib().rt_line_intersects_box(ptr(_f32(origin)), ptr(_f32(direction)), ptr(_f32(box))))


def tri_intersects_box(tri, box):
    return bool(_lib.lib().rt_tri_intersects_box(ptr(_f32(tri)), ptr(_f32(box))))


class BSPTree:
    """src/bsp.hpp:33-48: same constructor inputs, same tree; built by the candidate-list builder of
    librt_amd.so.  Arrays are exposed (the reference keeps them private) because the device upload
    needs them."""

    MAX_TRIS = 32       # Leaf::MAX_TRIS, src/bsp.hpp:162

    def __init__(self, inst_positions, vertex_begin, tris, tri_ranges, instance_meshes, box, max_leaf_tris=32,
                 max_depth=60, threads=0, device=None):
        """device: None builds on the host (rt_bsp_build, `threads` workers); a GPU number builds the same tree
        with the per-level work on that GPU (rt_bsp_build_device)."""
        L = _lib.lib()
        h = _lib.vp()
        mm = _u32(instance_meshes)
        build = L.rt_bsp_build if device is None else L.rt_bsp_build_device
        _lib.check(build(ptr(_f32(inst_positions)), ptr(_u32(vertex_begin)), ptr(_u32(tris)),
                         ptr(_u32(tri_ranges, (-1, 2))), ptr(mm), c_u32(len(mm)), ptr(_f32(box, (6,))),
                         c_u32(max_leaf_tris), c_u32(max_depth), c_int(threads if device is None else device),
                         _lib.ctypes.byref(h)))
        try:
            counts = np.zeros(5, np.uint32)
            L.rt_bsp_counts(h, ptr(counts))
            self.inode_count, self.leaf_count, self.ref_count, self.depth, self.oversized_leaves = map(int, counts)
            root_box = np.zeros(6, np.float32)
            root_index = _lib.c_i32()
            pi, pf, pr = _lib.vp(), _lib.vp(), _lib.vp()
            L.rt_bsp_view(h, ptr(root_box), _lib.ctypes.byref(root_index), _lib.ctypes.byref(pi),
                          _lib.ctypes.byref(pf), _lib.ctypes.byref(pr))
            self.root_box, self.root_index = root_box, int(root_index.value)

            def view(p, dtype, n):
                if n == 0 or not p.value:
                    return np.zeros(0, dtype)
                buf = (_lib.ctypes.c_uint8 * (n * np.dtype(dtype).itemsize)).from_address(p.value)
                return np.frombuffer(buf, dtype=dtype).copy()

            self.inodes = view(pi, _lib.INODE_DTYPE, self.inode_count)
            self.leaf_first_ref = view(pf, np.uint32, self.leaf_count + 1)
            self.leaf_refs = view(pr, _lib.TRI_REF_DTYPE, self.ref_count)
        finally:
            L.rt_bsp_destroy(h)


def exhaustive_box_counts(inst_positions, vertex_begin, tris, tri_ranges, instance_meshes, boxes, device=0):
    """Per box, how many triangles of the whole scene pass the reference's triangle/box predicate
    (src/geometry.hpp:217-428), evaluated on a GPU with no candidate list: the numbers the reference's constructor
    computes at every node (src/bsp.hpp:188-211)."""
    b = _f32(boxes, (-1, 6))
    mm = _u32(instance_meshes)
    counts = np.zeros(len(b), np.uint32)
    _lib.check(_lib.lib().rt_bsp_exhaustive_counts_device(
        ptr(_f32(inst_positions)), ptr(_u32(vertex_begin)), ptr(_u32(tris)), ptr(_u32(tri_ranges, (-1, 2))), ptr(mm),
        c_u32(len(mm)), ptr(b), c_u32(len(b)), c_int(device), ptr(counts)))
    return counts


class Scene:
    """src/scene.hpp:29-41 plus the preprocessing sequence of src/main.cpp:191-216."""

    def __init__(self, camera, meshes, materials, models, inflate_mode=0):
        self.camera, self.meshes, self.materials, self.models = camera, meshes, list(materials), models
        self.inflate_mode = inflate_mode
        self.instantiated_meshes = None
        self.preprocessed_tris = None
        self.preprocessed_materials = None
        self.bounding_box = None
        self.bsp_tree = None

    def preprocess(self, max_leaf_tris=32, max_depth=60, threads=0, device=None, bsp_device=None):
        """`device`: run instancing and triangle preprocessing on that GPU instead of the host (same arrays);
        `bsp_device`: build the tree with rt_bsp_build_device on that GPU (same tree)."""
        self.preprocessed_materials = np.stack([preprocess_material(m) for m in self.materials])
        self.instantiated_meshes = instantiate_meshes(self.meshes.vertex_positions, self.meshes.vertex_normals,
                                                      self.meshes.vertex_ranges, self.models.mesh_transforms,
                                                      self.models.meshes, device)
        im = self.instantiated_meshes
        self.preprocessed_tris = preprocess_tris(im.vertex_positions, im.vertex_begin, self.meshes.tris,
                                                 self.meshes.tri_ranges, self.models.meshes, device)
        self.bounding_box = scene_bounding_box(im.vertex_positions, self.inflate_mode)
        self.bsp_tree = BSPTree(im.vertex_positions, im.vertex_begin, self.meshes.tris, self.meshes.tri_ranges,
                                self.models.meshes, self.bounding_box, max_leaf_tris, max_depth, threads, bsp_device)
        return self

    def pixel_to_ray_transform(self, width, height):
        c = self.camera
        return pixel_to_ray_transform(c.forward(), c.down(), c.right(), c.fov, width, height)
