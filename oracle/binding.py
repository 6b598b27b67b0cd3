"""TEST INFRASTRUCTURE ONLY: ctypes bindings for the C oracle (liboracle.so) and, in the development
container, for the compiled reference (oracle/_ref/libref.so).  Both libraries use the same array
layouts (see rt_oracle.h / ref_harness.cpp), so one wrapper class serves both.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libref.so")

_c = ctypes
_vp = ctypes.c_void_p


def _p(a):
    return a.ctypes.data_as(_vp)


def build_oracle(force=False):
    """Compile liboracle.so (and libref.so when /root/reference exists)."""
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(HERE, "rt_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"])
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")])


class RenderParams(ctypes.Structure):
    _fields_ = [("width", _c.c_uint32), ("height", _c.c_uint32), ("x0", _c.c_uint32), ("y0", _c.c_uint32),
                ("w", _c.c_uint32), ("h", _c.c_uint32), ("spp", _c.c_uint32), ("bounceLimit", _c.c_uint32),
                ("tMin", _c.c_float), ("seed", _c.c_uint64), ("cameraPosition", _c.c_float * 3),
                ("pixelToRay", _c.c_float * 9), ("flags", _c.c_uint32)]


FLAG_RUSSIAN_ROULETTE = 4
FLAG_ACCEL_BVH = 8
FLAG_LOBE_SAMPLING = 16


class Stats(ctypes.Structure):
    _fields_ = [(n, _c.c_uint64) for n in
                ("segments", "shadedHits", "boxTests", "trisTested", "leafVisits", "inodeVisits", "paths")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class BaseScene:
    """Flat description of a scene before preprocessing (the inputs of src/main.cpp:134-187)."""

    def __init__(self, positions, normals, tris, mesh_vertex_ranges, mesh_tri_ranges, materials, transforms,
                 model_meshes, model_materials, camera, inflate_mode=1):
        self.positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        self.tris = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
        self.mesh_vertex_ranges = np.ascontiguousarray(mesh_vertex_ranges, np.uint32).reshape(-1, 2)
        self.mesh_tri_ranges = np.ascontiguousarray(mesh_tri_ranges, np.uint32).reshape(-1, 2)
        self.materials = np.ascontiguousarray(materials, np.float32).reshape(-1, 8)
        self.transforms = np.ascontiguousarray(transforms, np.float32).reshape(-1, 10)
        self.model_meshes = np.ascontiguousarray(model_meshes, np.uint32).ravel()
        self.model_materials = np.ascontiguousarray(model_materials, np.uint32).ravel()
        self.camera = np.ascontiguousarray(camera, np.float32).ravel()
        self.inflate_mode = int(inflate_mode)
        assert self.camera.size == 8


class _Lib:
    """Common wrapper; `prefix` is 'orc' (C oracle) or 'ref' (compiled reference)."""

    def __init__(self, path, prefix):
        self.lib = ctypes.CDLL(path)
        self.prefix = prefix
        self.is_ref = prefix == "ref"
        f = self._f
        f("scene_create").restype = _vp
        if self.is_ref:
            self.lib.ref_scene_builtin.restype = _vp

    def _f(self, name):
        return getattr(self.lib, f"{self.prefix}_{name}")

    # -- predicates ------------------------------------------------------------------------
    def line_intersects_box(self, origins, directions, boxes):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(directions, np.float32)
        b = np.ascontiguousarray(boxes, np.float32)
        n = o.shape[0]
        out = np.zeros(n, np.uint8)
        self._f("line_intersects_box")(_p(o), _p(d), _p(b), _c.c_size_t(n), _p(out))
        return out

    def tri_intersects_box(self, tris, boxes):
        t = np.ascontiguousarray(tris, np.float32)
        b = np.ascontiguousarray(boxes, np.float32)
        n = t.shape[0]
        out = np.zeros(n, np.uint8)
        self._f("tri_intersects_box")(_p(t), _p(b), _c.c_size_t(n), _p(out))
        return out

    # -- post-processing (src/main.cpp:233-240) -----------------------------------------------
    def tone_srgb_flags(self, hdr):
        h = np.ascontiguousarray(hdr, np.float32)
        out = np.empty_like(h)
        self._f("tone_srgb_flags")(_p(h), _c.c_size_t(h.size // 3), _p(out))
        return out

    def median3(self, image):
        a = np.ascontiguousarray(image, np.float32)
        height, width = a.shape[0], a.shape[1]
        out = np.empty_like(a)
        self._f("median3")(_p(a), _c.c_uint32(width), _c.c_uint32(height), _p(out))
        return out

    def to_u8(self, image):
        a = np.ascontiguousarray(image, np.float32)
        out = np.empty(a.shape, np.uint8)
        n = a.size if not self.is_ref else a.size // 3
        self._f("to_u8")(_p(a), _c.c_size_t(n), _p(out))
        return out

    def postprocess(self, hdr):
        """HDR (h, w, 3) float32 -> (h, w, 3) uint8, the whole of src/main.cpp:233-240."""
        return self.to_u8(self.median3(self.tone_srgb_flags(hdr)))

    # -- scenes ----------------------------------------------------------------------------
    def scene_create(self, base, width, height):
        idx_t = np.uint16 if self.is_ref else np.uint32
        tris = np.ascontiguousarray(base.tris, idx_t)
        mm = np.ascontiguousarray(base.model_meshes, idx_t)
        mmat = np.ascontiguousarray(base.model_materials, idx_t)
        if self.is_ref:
            assert base.tris.max(initial=0) < 65536 and len(base.model_meshes) < 65536
        h = self._f("scene_create")(
            _p(base.positions), _p(base.normals), _c.c_uint32(len(base.positions)),
            _p(tris), _c.c_uint32(len(tris)),
            _p(base.mesh_vertex_ranges), _p(base.mesh_tri_ranges), _c.c_uint32(len(base.mesh_vertex_ranges)),
            _p(base.materials), _c.c_uint32(len(base.materials)),
            _p(base.transforms), _p(mm), _p(mmat), _c.c_uint32(len(mm)),
            _p(base.camera), _c.c_uint32(width), _c.c_uint32(height), _c.c_int(base.inflate_mode))
        if not h:
            raise RuntimeError("scene_create failed (tree deeper than 64 levels)")
        return Scene(self, h, width, height)

    def scene_create_with_tree(self, base, width, height, bsp_tree):
        """C oracle only: preprocess `base` but traverse the supplied tree (a raytracing_amd BSPTree)."""
        assert not self.is_ref
        t = bsp_tree
        inodes = np.ascontiguousarray(t.inodes)
        first = np.ascontiguousarray(t.leaf_first_ref, np.uint32)
        refs = np.ascontiguousarray(np.stack([t.leaf_refs["mesh"], t.leaf_refs["tri"]], 1), np.uint32) \
            if t.ref_count else np.zeros((0, 2), np.uint32)
        f = self.lib.orc_scene_create_with_tree
        f.restype = _vp
        h = f(_p(base.positions), _p(base.normals), _c.c_uint32(len(base.positions)),
              _p(base.tris), _c.c_uint32(len(base.tris)),
              _p(base.mesh_vertex_ranges), _p(base.mesh_tri_ranges), _c.c_uint32(len(base.mesh_vertex_ranges)),
              _p(base.materials), _c.c_uint32(len(base.materials)),
              _p(base.transforms), _p(base.model_meshes), _p(base.model_materials), _c.c_uint32(len(base.model_meshes)),
              _p(base.camera), _c.c_uint32(width), _c.c_uint32(height), _c.c_int(base.inflate_mode),
              _c.c_int32(t.root_index), _p(inodes), _c.c_uint32(t.inode_count), _p(first), _p(refs),
              _c.c_uint32(t.leaf_count))
        if not h:
            raise RuntimeError("scene_create_with_tree failed")
        return Scene(self, h, width, height)

    def scene_builtin(self, divisor, width, height):
        assert self.is_ref
        return Scene(self, self.lib.ref_scene_builtin(_c.c_uint(divisor), _c.c_uint(width), _c.c_uint(height)),
                     width, height)


class Scene:
    def __init__(self, lib, handle, width, height):
        self.L = lib
        self.h = _vp(handle)
        self.width, self.height = width, height
        c = np.zeros(9, np.uint32)
        lib._f("scene_counts")(self.h, _p(c))
        (self.n_models, self.n_vertices, self.n_tris, self.n_inodes, self.n_leaves, self.n_base_vertices,
         self.n_base_tris, self.n_meshes, self.n_materials) = [int(x) for x in c]

    def close(self):
        if self.h:
            self.L._f("scene_destroy")(self.h)
            self.h = None

    def base(self):
        """Only the compiled reference exports its base description (used for the built-in scene)."""
        assert self.L.is_ref
        pos = np.zeros((self.n_base_vertices, 3), np.float32)
        nrm = np.zeros_like(pos)
        tris = np.zeros((self.n_base_tris, 3), np.uint32)
        mvr = np.zeros((self.n_meshes, 2), np.uint32)
        mtr = np.zeros((self.n_meshes, 2), np.uint32)
        mats = np.zeros((self.n_materials, 8), np.float32)
        trs = np.zeros((self.n_models, 10), np.float32)
        mm = np.zeros(self.n_models, np.uint32)
        mmat = np.zeros(self.n_models, np.uint32)
        cam = np.zeros(8, np.float32)
        self.L.lib.ref_get_base(self.h, _p(pos), _p(nrm), _p(tris), _p(mvr), _p(mtr), _p(mats), _p(trs), _p(mm),
                                _p(mmat), _p(cam))
        return BaseScene(pos, nrm, tris, mvr, mtr, mats, trs, mm, mmat, cam, inflate_mode=0)

    def preprocessed(self):
        d = dict(positions=np.zeros((self.n_vertices, 3), np.float32),
                 normals=np.zeros((self.n_vertices, 3), np.float32),
                 vertex_ranges=np.zeros((self.n_models, 2), np.uint32),
                 pretris=np.zeros((self.n_tris, 12), np.float32),
                 tri_ranges=np.zeros((self.n_models, 2), np.uint32),
                 model_materials=np.zeros((self.n_models, 12), np.float32),
                 camera_position=np.zeros(3, np.float32),
                 pixel_to_ray=np.zeros(9, np.float32),
                 root_box=np.zeros(6, np.float32))
        self.L._f("get_preprocessed")(self.h, *[_p(v) for v in d.values()])
        return d

    def bsp(self):
        d = dict(root_box=np.zeros(6, np.float32), root_index=np.zeros(1, np.int32),
                 inode_boxes=np.zeros((self.n_inodes, 12), np.float32),
                 inode_ints=np.zeros((self.n_inodes, 3), np.int32),
                 leaf_counts=np.zeros(self.n_leaves, np.uint32),
                 leaf_refs=np.zeros((self.n_leaves, 32, 2), np.uint32))
        self.L._f("get_bsp")(self.h, *[_p(v) for v in d.values()])
        return d

    def intersect(self, origins, directions, t_min=1e-3, front_only=True, use_bvh=False):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(directions, np.float32)
        n = o.shape[0]
        f = np.zeros((n, 6), np.float32)
        ids = np.zeros((n, 2), np.uint32)
        assert not (use_bvh and self.L.is_ref)
        self.L._f("intersect")(self.h, _p(o), _p(d), _c.c_size_t(n), _c.c_float(t_min),
                               _c.c_int(int(bool(front_only)) | (2 if use_bvh else 0)), _p(f), _p(ids))
        return f, ids

    def intersect_brute(self, origins, directions, t_min=1e-3, front_only=True):
        """C oracle only: nearest accepted triangle by scanning all triangles (no accelerator)."""
        assert not self.L.is_ref
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(directions, np.float32)
        n = o.shape[0]
        f = np.zeros((n, 6), np.float32)
        ids = np.zeros((n, 2), np.uint32)
        self.L.lib.orc_intersect_brute(self.h, _p(o), _p(d), _c.c_size_t(n), _c.c_float(t_min),
                                       _c.c_int(int(bool(front_only))), _p(f), _p(ids))
        return f, ids

    def leaf_hits(self, leaves, boxes, origins, directions, t_min=1e-3, front_only=True):
        """C oracle only: visitLeaf alone (src/bsp.hpp:58-104) for line i against leaf leaves[i] with box boxes[i]."""
        assert not self.L.is_ref
        lv = np.ascontiguousarray(leaves, np.uint32)
        bx = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(directions, np.float32)
        n = o.shape[0]
        assert len(lv) == n and len(bx) == n
        out = np.zeros(n, np.uint8)
        t = np.zeros(n, np.float32)
        self.L.lib.orc_leaf_hits(self.h, _p(lv), _p(bx), _p(o), _p(d), _c.c_size_t(n), _c.c_float(t_min),
                                 _c.c_int(int(bool(front_only))), _p(out), _p(t))
        return out.astype(bool), t

    def leaf_facing_counts(self, leaves, directions):
        """C oracle only: per line, how many triangles of leaf leaves[i] pass the FRONT_ONLY facing test."""
        assert not self.L.is_ref
        lv = np.ascontiguousarray(leaves, np.uint32)
        d = np.ascontiguousarray(directions, np.float32)
        out = np.zeros(len(lv), np.uint32)
        self.L.lib.orc_leaf_facing_counts(self.h, _p(lv), _p(d), _c.c_size_t(len(lv)), _p(out))
        return out

    def set_bvh4(self, nodes, tri_order):
        """C oracle only: hand over the product's bounding volume hierarchy (64-byte nodes, triangle order) for the
        restated traversal (intersect(use_bvh=True), render_params(flags=FLAG_ACCEL_BVH))."""
        assert not self.L.is_ref
        nodes = np.ascontiguousarray(nodes)
        assert nodes.dtype.itemsize == 64
        order = np.ascontiguousarray(tri_order, np.uint32)
        f = self.L.lib.orc_scene_set_bvh4
        f.restype = _c.c_int
        rc = f(self.h, _p(nodes), _c.c_uint32(len(nodes)), _p(order), _c.c_uint32(len(order)))
        assert rc == 0, "orc_scene_set_bvh4: size mismatch"

    def camera_rays(self, sample_xy):
        xy = np.ascontiguousarray(sample_xy, np.float32)
        out = np.zeros((xy.shape[0], 3), np.float32)
        self.L._f("camera_rays")(self.h, _p(xy), _c.c_size_t(xy.shape[0]), _p(out))
        return out

    def ray_trace(self, origins, directions, seeds):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(directions, np.float32)
        s = np.ascontiguousarray(seeds, np.uint64)
        out = np.zeros((o.shape[0], 3), np.float32)
        self.L._f("ray_trace")(self.h, _p(o), _p(d), _p(s), _c.c_size_t(o.shape[0]), _p(out))
        return out

    def render(self, spp, seed):
        img = np.zeros((self.height, self.width, 3), np.float32)
        self.L._f("render")(self.h, _c.c_uint32(spp), _c.c_uint64(seed), _p(img))
        return img

    # -- device-mode (C oracle only) ----------------------------------------------------------
    def set_simd_leaves(self, enable=True):
        """C oracle only: leaf test 8 triangles per AVX2 step (src/geometry.hpp:117-132) instead of one at a time."""
        self.L.lib.orc_scene_set_simd_leaves(self.h, _c.c_int(int(enable)))

    def render_params(self, spp, seed=0x5EED, bounce_limit=8, t_min=1e-3, rect=None, flags=0):
        pre = dict(camera_position=np.zeros(3, np.float32), pixel_to_ray=np.zeros(9, np.float32))
        self.L.lib.orc_get_camera(self.h, _p(pre["camera_position"]), _p(pre["pixel_to_ray"]))
        p = RenderParams()
        p.flags = flags
        p.width, p.height = self.width, self.height
        x0, y0, w, h = rect if rect is not None else (0, 0, self.width, self.height)
        p.x0, p.y0, p.w, p.h = x0, y0, w, h
        p.spp, p.bounceLimit, p.tMin, p.seed = spp, bounce_limit, t_min, seed
        p.cameraPosition[:] = [float(x) for x in pre["camera_position"]]
        p.pixelToRay[:] = [float(x) for x in pre["pixel_to_ray"]]
        return p

    def render_device_mode(self, params, threads=0, with_stats=True):
        assert not self.L.is_ref
        img = np.zeros((params.h, params.w, 3), np.float32)
        st = Stats()
        self.L.lib.orc_render_device_mode(self.h, ctypes.byref(params), _p(img),
                                          ctypes.byref(st) if with_stats else None,
                                          _c.c_int(threads or (os.cpu_count() or 1)))
        return img, st.as_dict()


_cache = {}


def oracle():
    if "orc" not in _cache:
        build_oracle()
        _cache["orc"] = _Lib(ORACLE_SO, "orc")
    return _cache["orc"]


def reference():
    """The compiled reference, or None where it is not available (e.g. the GPU box without the .so)."""
    if "ref" not in _cache:
        _cache["ref"] = _Lib(REF_SO, "ref") if os.path.exists(REF_SO) else None
    return _cache["ref"]
