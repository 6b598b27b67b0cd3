"""Render entry points over the C ABI of librt_amd.so (include/rt_amd.h).

Mirrors src/render.hpp: `RayTraceData`, `RenderData`, `render()`; `BSPTree.line_tri_nearest_intersection`
is `DeviceScene.intersect`.  All computation happens in the HIP kernels; nothing here falls back to
the CPU.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import c_u32, c_u64, c_f32, c_int, ptr

PIXEL_SAMPLE_RATE = 2048            # src/render.hpp:47
RAY_INTERSECTION_T_MIN = 1e-3       # src/render.hpp:48
RAY_BOUNCE_LIMIT = 8                # src/render.hpp:49


class RayTraceData:
    """src/render.hpp:28-35 with the model->mesh and model->material permutations applied."""

    def __init__(self, scene):
        if scene.bsp_tree is None:
            scene.preprocess()
        self.scene = scene
        mm = np.asarray(scene.models.meshes, np.uint32)
        self.bsp_tree = scene.bsp_tree
        self.vertex_normals = scene.instantiated_meshes.vertex_normals
        self.vertex_begin = scene.instantiated_meshes.vertex_begin
        self.tris = scene.meshes.tris
        self.model_tri_begin = np.ascontiguousarray(scene.meshes.tri_ranges[mm, 0], np.uint32)
        self.materials = np.ascontiguousarray(
            scene.preprocessed_materials[np.asarray(scene.models.materials, np.int64)], np.float32)
        self.pretris = scene.preprocessed_tris.tris
        self.pretri_begin = scene.preprocessed_tris.tri_begin


def _describe(d):
    """rt_scene_desc over the arrays of a RayTraceData; returns (desc, arrays to keep alive)."""
    t = d.bsp_tree
    keep = dict(inodes=np.ascontiguousarray(t.inodes), first=np.ascontiguousarray(t.leaf_first_ref, np.uint32),
                refs=np.ascontiguousarray(t.leaf_refs), pretris=np.ascontiguousarray(d.pretris, np.float32),
                ptb=np.ascontiguousarray(d.pretri_begin, np.uint32),
                vn=np.ascontiguousarray(d.vertex_normals, np.float32),
                vb=np.ascontiguousarray(d.vertex_begin, np.uint32), bt=np.ascontiguousarray(d.tris, np.uint32),
                mtb=np.ascontiguousarray(d.model_tri_begin, np.uint32),
                mats=np.ascontiguousarray(d.materials, np.float32))
    desc = _lib.RtSceneDesc()
    desc.root_box[:] = [float(x) for x in t.root_box]
    desc.root_index = t.root_index
    desc.inodes, desc.inode_count = ptr(keep["inodes"]), t.inode_count
    desc.leaf_first_ref, desc.leaf_refs, desc.leaf_count = ptr(keep["first"]), ptr(keep["refs"]), t.leaf_count
    desc.pretris, desc.model_pretri_begin, desc.pretri_count = ptr(keep["pretris"]), ptr(keep["ptb"]), len(keep["pretris"])
    desc.vertex_normals, desc.model_vertex_begin, desc.vertex_count = ptr(keep["vn"]), ptr(keep["vb"]), len(keep["vn"])
    desc.base_tris, desc.model_tri_begin, desc.base_tri_count = ptr(keep["bt"]), ptr(keep["mtb"]), len(keep["bt"])
    desc.model_materials, desc.model_count = ptr(keep["mats"]), len(keep["mats"])
    return desc, keep


def serialize_scene(ray_trace_data):
    """rt_scene_serialize: the prepared scene as one uint8 array (what rank 0 broadcasts to the other GPUs'
    processes)."""
    desc, keep = _describe(ray_trace_data)
    size = c_u64(0)
    _lib.check(_lib.lib().rt_scene_serialize(ctypes.byref(desc), None, ctypes.byref(size)))
    blob = np.zeros(size.value, np.uint8)
    _lib.check(_lib.lib().rt_scene_serialize(ctypes.byref(desc), ptr(blob), ctypes.byref(size)))
    del keep
    return blob


class FlatTree:
    """rt_debug_flatten: the node and leaf bounds records exactly as rt_scene_create uploads them (host only, no GPU).
    `nodes` is a structured view of the node records, `bounds` of the leaf bounds records, both indexed by node slot;
    `leaf_axes` is the form of the bounds records (1: one cone, 3: up to three axes)."""

    NODE_DTYPE = np.dtype([("lo", np.float32, 3), ("hi", np.float32, 3), ("index", np.int32), ("aux", np.uint32)])
    BOUNDS_DTYPE = np.dtype([("lo", np.float32, 3), ("hi", np.float32, 3), ("cone", np.uint32), ("index", np.int32),
                             ("cK", np.float32), ("cE", np.float32), ("cN", np.float32), ("normals", np.uint32, 5)])

    def __init__(self, ray_trace_data):
        L = _lib.lib()
        desc, keep = _describe(ray_trace_data)
        n, slots, axes, amax = c_u64(0), c_u32(0), c_u32(0), c_f32(0)
        _lib.check(L.rt_debug_flatten(ctypes.byref(desc), None, ctypes.byref(n), ctypes.byref(slots), ctypes.byref(axes),
                                      ctypes.byref(amax)))
        self.packed = np.zeros(n.value, np.float32)
        _lib.check(L.rt_debug_flatten(ctypes.byref(desc), ptr(self.packed), ctypes.byref(n), ctypes.byref(slots),
                                      ctypes.byref(axes), ctypes.byref(amax)))
        del keep
        self.node_slots, self.leaf_axes, self.box_abs_max = slots.value, axes.value, float(amax.value)
        self.nodes = self.packed[16 * self.node_slots:].view(self.NODE_DTYPE)
        self.bounds = self.packed[:16 * self.node_slots].view(self.BOUNDS_DTYPE)[::-1]
        self.leaf_slots = np.nonzero(self.nodes["index"] < 0)[0].astype(np.uint32)

    def cone_skip(self, slots, directions):
        """rt_debug_cone_skip: would the traversal skip the leaf at node slot slots[i] for direction i because its normal
        cone says no triangle faces the line?"""
        words = np.ascontiguousarray(self.nodes["aux"][np.asarray(slots, np.int64)], np.uint32)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        out = np.zeros(len(d), np.uint8)
        _lib.check(_lib.lib().rt_debug_cone_skip(ptr(words), ptr(d), c_u64(len(d)), ptr(out)))
        return out.astype(bool)

    def described(self):
        """Per node slot: does the record describe the leaf's normals at all (sin b below 1)?"""
        n = self.bounds["normals"]
        return (n[:, 0].view(np.float32) < 1.0) if self.leaf_axes == 1 else ((n[:, 0] & 0xFFFF) < 0xFFFF)

    def leaf_drop(self, slots, origins, directions):
        """rt_debug_leaf_drop: would the traversal leave the leaf at node slot slots[i] unscanned for line i?"""
        sl = np.ascontiguousarray(slots, np.uint32)
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        assert len(sl) == len(o) == len(d)
        out = np.zeros(len(o), np.uint8)
        _lib.check(_lib.lib().rt_debug_leaf_drop(ptr(self.packed), c_u32(self.node_slots), c_u32(self.leaf_axes),
                                                 c_f32(self.box_abs_max), ptr(sl), ptr(o), ptr(d), c_u64(len(o)), ptr(out)))
        return out.astype(bool)


class Bvh4:
    """rt_bvh4_build: the alternative accelerator, a 4-wide bounding volume hierarchy over preprocessed
    triangles (12 floats each), built on the host.  `nodes` (BVH4_NODE_DTYPE) and `tri_order` are copies."""

    def __init__(self, pretris, max_leaf_tris=4, threads=0):
        L = _lib.lib()
        pt = np.ascontiguousarray(pretris, np.float32).reshape(-1, 12)
        h = _lib.vp()
        _lib.check(L.rt_bvh4_build(ptr(pt), c_u32(len(pt)), c_u32(max_leaf_tris), c_int(threads), ctypes.byref(h)))
        try:
            counts = (c_u32 * 6)()
            L.rt_bvh4_counts(h, counts)
            (self.node_count, self.tri_count, self.depth, self.stack_need, self.leaf_count,
             self.largest_leaf) = [int(x) for x in counts]
            pn, po = _lib.vp(), _lib.vp()
            L.rt_bvh4_view(h, ctypes.byref(pn), ctypes.byref(po))
            raw = ctypes.string_at(pn.value, 64 * self.node_count)
            self.nodes = np.frombuffer(raw, _lib.BVH4_NODE_DTYPE).copy()
            if self.tri_count:
                self.tri_order = np.frombuffer(ctypes.string_at(po.value, 4 * self.tri_count), np.uint32).copy()
            else:
                self.tri_order = np.zeros(0, np.uint32)
        finally:
            L.rt_bvh4_destroy(h)


class DeviceScene:
    """A scene resident in HBM (rt_scene).  One per GPU."""

    def __init__(self, ray_trace_data=None, device=0, blob=None):
        L = _lib.lib()
        self._h = _lib.vp()
        if blob is not None:
            blob = np.ascontiguousarray(blob, np.uint8)
            _lib.check(L.rt_scene_create_from_blob(ptr(blob), c_u64(blob.size), c_int(device), ctypes.byref(self._h)))
        else:
            desc, keep = _describe(ray_trace_data)
            _lib.check(L.rt_scene_create(ctypes.byref(desc), c_int(device), ctypes.byref(self._h)))
            del keep
        self.device = device
        self._source = (ray_trace_data, blob)
        self.has_bvh = False
        self._refresh_info()

    def _refresh_info(self):
        L = _lib.lib()
        nbytes, nodes, ltris, depth = c_u64(), c_u32(), c_u32(), c_u32()
        _lib.check(L.rt_scene_info(self._h, ctypes.byref(nbytes), ctypes.byref(nodes), ctypes.byref(ltris),
                                   ctypes.byref(depth)))
        self.device_bytes, self.node_count, self.leaf_tri_count, self.depth = (nbytes.value, nodes.value,
                                                                               ltris.value, depth.value)

    def _desc(self):
        rtd, blob = self._source
        if rtd is not None:
            return _describe(rtd)
        desc = _lib.RtSceneDesc()
        _lib.check(_lib.lib().rt_scene_desc_from_blob(ptr(blob), c_u64(blob.size), ctypes.byref(desc)))
        return desc, blob

    def build_bvh(self, max_leaf_tris=4, threads=0):
        """rt_scene_build_bvh4: build the 4-wide bounding volume hierarchy over the scene's triangles and
        attach it; RT_FLAG_ACCEL_BVH / intersect(use_bvh=True) then answer queries from it."""
        desc, keep = self._desc()
        _lib.check(_lib.lib().rt_scene_build_bvh4(self._h, ctypes.byref(desc), c_u32(max_leaf_tris), c_int(threads)))
        del keep
        self.has_bvh = True
        self._refresh_info()

    def attach_bvh(self, nodes, tri_order):
        """rt_scene_attach_bvh4 with caller-supplied nodes (BVH4_NODE_DTYPE) and triangle order."""
        desc, keep = self._desc()
        nodes = np.ascontiguousarray(nodes, _lib.BVH4_NODE_DTYPE)
        order = np.ascontiguousarray(tri_order, np.uint32)
        _lib.check(_lib.lib().rt_scene_attach_bvh4(self._h, ctypes.byref(desc), ptr(nodes), c_u32(len(nodes)), ptr(order)))
        del keep
        self.has_bvh = True
        self._refresh_info()

    @classmethod
    def from_blob(cls, blob, device=0):
        """rt_scene_create_from_blob: upload a scene received as bytes (see serialize_scene)."""
        return cls(None, device, blob=blob)

    def close(self):
        if self._h:
            _lib.lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # BSPTree::lineTriNearestIntersection<Surfaces>, src/bsp.hpp:50-145
    def intersect(self, origins, directions, t_min=RAY_INTERSECTION_T_MIN, front_only=True, use_bvh=False):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        out = np.zeros(len(o), _lib.HIT_DTYPE)
        mode = (_lib.RT_INTERSECT_FRONT_ONLY if front_only else 0) | (_lib.RT_INTERSECT_BVH if use_bvh else 0)
        _lib.check(_lib.lib().rt_intersect(self._h, ptr(o), ptr(d), c_u64(len(o)), c_f32(t_min),
                                           c_int(mode), ptr(out)))
        return out

    def intersect_check(self, origins, directions, t_min=RAY_INTERSECTION_T_MIN, front_only=True):
        """rt_intersect_check: the same hits through the kernels that count work and cross-check every conservative
        shortcut against the exact evaluation; returns (hits, stats dict)."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        out = np.zeros(len(o), _lib.HIT_DTYPE)
        st = _lib.RtStats()
        _lib.check(_lib.lib().rt_intersect_check(self._h, ptr(o), ptr(d), c_u64(len(o)), c_f32(t_min),
                                                 c_int(_lib.RT_INTERSECT_FRONT_ONLY if front_only else 0), ptr(out),
                                                 ctypes.byref(st)))
        return out, st.as_dict()

    def intersect_device(self, d_origins, d_directions, n, d_out, t_min=RAY_INTERSECTION_T_MIN, front_only=True,
                         stream=None):
        """Device-pointer form (integers or objects with data_ptr())."""
        def dp(x):
            return ctypes.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        _lib.check(_lib.lib().rt_intersect_device(self._h, dp(d_origins), dp(d_directions), c_u64(n), c_f32(t_min),
                                                  c_int(int(front_only)), dp(d_out),
                                                  ctypes.c_void_p(stream or 0)))

    def make_params(self, width, height, camera_position, pixel_to_ray, spp=PIXEL_SAMPLE_RATE,
                    bounce_limit=RAY_BOUNCE_LIMIT, t_min=RAY_INTERSECTION_T_MIN, seed=0x5EED, rect=None, flags=0,
                    max_batch_paths=0, row_group=0, row_stride=0):
        p = _lib.RtRenderParams()
        p.width, p.height = width, height
        p.x0, p.y0, p.w, p.h = rect if rect is not None else (0, 0, width, height)
        p.spp, p.bounce_limit, p.t_min, p.seed = spp, bounce_limit, t_min, seed
        p.camera_position[:] = [float(x) for x in camera_position]
        p.pixel_to_ray[:] = [float(x) for x in pixel_to_ray]
        p.flags, p.max_batch_paths = flags, max_batch_paths
        p.row_group, p.row_stride = row_group, row_stride
        return p

    def render(self, params):
        """rt_render: returns (image[h, w, 3] float32 on the host, stats dict)."""
        img = np.zeros((params.h, params.w, 3), np.float32)
        st = _lib.RtStats()
        _lib.check(_lib.lib().rt_render(self._h, ctypes.byref(params), ptr(img), ctypes.byref(st)))
        return img, st.as_dict()

    def render_device(self, params, d_out, stream=None, want_stats=True):
        """rt_render_device: writes w*h*3 floats at device pointer d_out; returns stats (syncs) or None."""
        st = _lib.RtStats()
        dp = ctypes.c_void_p(d_out.data_ptr() if hasattr(d_out, "data_ptr") else int(d_out))
        _lib.check(_lib.lib().rt_render_device(self._h, ctypes.byref(params), dp,
                                               ctypes.byref(st) if want_stats else None,
                                               ctypes.c_void_p(stream or 0)))
        return st.as_dict() if want_stats else None

    def render_device_static(self, params, d_out, stream=None, want_stats=True):
        """rt_render_device_static: the same image from the library's fixed workspace, no allocation."""
        st = _lib.RtStats()
        dp = ctypes.c_void_p(d_out.data_ptr() if hasattr(d_out, "data_ptr") else int(d_out))
        _lib.check(_lib.lib().rt_render_device_static(self._h, ctypes.byref(params), dp,
                                                      ctypes.byref(st) if want_stats else None,
                                                      ctypes.c_void_p(stream or 0)))
        return st.as_dict() if want_stats else None


def render_multi(device_scenes, params):
    """rt_render_multi: one frame over several scene handles (normally one per GPU), rows dealt in bands, bands gathered
    GPU-to-GPU on the first handle's GPU.  Returns (image[h, w, 3] float32 on the host, summed stats)."""
    handles = (ctypes.c_void_p * len(device_scenes))(*[d._h.value for d in device_scenes])
    img = np.zeros((params.h, params.w, 3), np.float32)
    st = _lib.RtStats()
    _lib.check(_lib.lib().rt_render_multi(handles, c_u32(len(device_scenes)), ctypes.byref(params), ptr(img), ctypes.byref(st)))
    return img, st.as_dict()


class RenderData:
    """src/render.hpp:38-44."""

    def __init__(self, image_width, image_height, camera_position, pixel_to_ray_transform, ray_trace_data):
        self.image_width, self.image_height = image_width, image_height
        self.camera_position = np.asarray(camera_position, np.float32)
        self.pixel_to_ray_transform = np.asarray(pixel_to_ray_transform, np.float32)
        self.ray_trace_data = ray_trace_data
        self._device_scene = None

    def device_scene(self, device=0):
        if self._device_scene is None:
            self._device_scene = DeviceScene(self.ray_trace_data, device)
        return self._device_scene


def render(data, image, spp=PIXEL_SAMPLE_RATE, bounce_limit=RAY_BOUNCE_LIMIT, seed=0x5EED, device=0):
    """render(RenderData const&, Span<glm::vec3>) of src/render.hpp:215-234: fills `image`
    (width*height x 3 float32, row-major) with the mean radiance per pixel.  Returns the stats."""
    assert image.size == data.image_width * data.image_height * 3        # src/render.hpp:216
    ds = data.device_scene(device)
    p = ds.make_params(data.image_width, data.image_height, data.camera_position, data.pixel_to_ray_transform,
                       spp=spp, bounce_limit=bounce_limit, seed=seed)
    img, stats = ds.render(p)
    np.copyto(image.reshape(data.image_height, data.image_width, 3), img)
    return stats
