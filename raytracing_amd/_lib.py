"""ctypes view of librt_amd.so (include/rt_amd.h, include/rt_host.h).

The library is the product: there is no Python or CPU fallback.  Importing this module fails loudly
when the shared object has not been built (run `python -c "import __graft_entry__ as g; g.build()"`
or `make -C raytracing_amd/csrc`).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB selects another build of the same library (e.g. the diagnostic build with phase stamps).
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")

c_u32, c_u64, c_i32, c_f32, c_int = ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int32, ctypes.c_float, ctypes.c_int
vp = ctypes.c_void_p


class RtInode(ctypes.Structure):
    """rt_inode: the reference's BSPTree::INode layout (src/bsp.hpp:148-159), 60 bytes."""
    _fields_ = [("neg_box", c_f32 * 6), ("neg_index", c_i32), ("pos_box", c_f32 * 6), ("pos_index", c_i32),
                ("axis", ctypes.c_uint8), ("pad_", ctypes.c_uint8 * 3)]


INODE_DTYPE = np.dtype([("neg_box", np.float32, 6), ("neg_index", np.int32), ("pos_box", np.float32, 6),
                        ("pos_index", np.int32), ("axis", np.uint8), ("pad_", np.uint8, 3)])
TRI_REF_DTYPE = np.dtype([("mesh", np.uint32), ("tri", np.uint32)])
HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32), ("point", np.float32, 3),
                      ("mesh", np.uint32), ("tri", np.uint32)])
# rt_bvh4_node: one 64-byte node of the alternative accelerator (include/rt_amd.h)
BVH4_NODE_DTYPE = np.dtype([("origin", np.float32, 3), ("exp", np.uint8, 3), ("pad0_", np.uint8),
                            ("qlo_x", np.uint8, 4), ("qlo_y", np.uint8, 4), ("qlo_z", np.uint8, 4),
                            ("qhi_x", np.uint8, 4), ("qhi_y", np.uint8, 4), ("qhi_z", np.uint8, 4),
                            ("pad1_", np.uint32, 2), ("child", np.uint32, 4)])
assert INODE_DTYPE.itemsize == 60 and ctypes.sizeof(RtInode) == 60 and HIT_DTYPE.itemsize == 32
assert BVH4_NODE_DTYPE.itemsize == 64


class RtSceneDesc(ctypes.Structure):
    _fields_ = [("root_box", c_f32 * 6), ("root_index", c_i32), ("inodes", vp), ("inode_count", c_u32),
                ("leaf_first_ref", vp), ("leaf_refs", vp), ("leaf_count", c_u32),
                ("pretris", vp), ("model_pretri_begin", vp), ("pretri_count", c_u32),
                ("vertex_normals", vp), ("model_vertex_begin", vp), ("vertex_count", c_u32),
                ("base_tris", vp), ("model_tri_begin", vp), ("base_tri_count", c_u32),
                ("model_materials", vp), ("model_count", c_u32), ("leaf_pretris", vp)]


class RtRenderParams(ctypes.Structure):
    _fields_ = [("width", c_u32), ("height", c_u32), ("x0", c_u32), ("y0", c_u32), ("w", c_u32), ("h", c_u32),
                ("spp", c_u32), ("bounce_limit", c_u32), ("t_min", c_f32), ("seed", c_u64),
                ("camera_position", c_f32 * 3), ("pixel_to_ray", c_f32 * 9), ("flags", c_u32),
                ("max_batch_paths", c_u32), ("row_group", c_u32), ("row_stride", c_u32)]


class RtStats(ctypes.Structure):
    _fields_ = [("paths", c_u64), ("segments", c_u64), ("shaded_hits", c_u64), ("box_tests", c_u64),
                ("tris_tested", c_u64), ("leaf_visits", c_u64), ("inode_visits", c_u64),
                ("ms_total", ctypes.c_double), ("ms_raygen", ctypes.c_double), ("ms_trace", ctypes.c_double),
                ("ms_shade", ctypes.c_double), ("ms_accumulate", ctypes.c_double),
                ("trace_launches", c_u32), ("batches", c_u32),
                ("box_exact_fallbacks", c_u64), ("box_filter_errors", c_u64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


RT_FLAG_COUNT_WORK = 1
RT_FLAG_PROFILE_KERNELS = 2
RT_FLAG_RUSSIAN_ROULETTE = 4
RT_FLAG_ACCEL_BVH = 8
RT_FLAG_LOBE_SAMPLING = 16
RT_FLAG_SORT_RAYS = 32
RT_INTERSECT_FRONT_ONLY = 1
RT_INTERSECT_BVH = 2
RT_STATIC_MAX_PIXELS = 262144

# Every symbol declared by include/rt_amd.h and include/rt_host.h.
DEVICE_SYMBOLS = ["rt_scene_create", "rt_scene_serialize", "rt_scene_desc_from_blob", "rt_scene_create_from_blob", "rt_scene_destroy", "rt_scene_attach_bvh4", "rt_scene_build_bvh4", "rt_scene_info", "rt_intersect", "rt_intersect_device", "rt_intersect_check",
                  "rt_render", "rt_render_device", "rt_render_device_static", "rt_render_multi", "rt_render_multi_device", "rt_postprocess", "rt_postprocess_device", "rt_selftest_math",
                  "rt_debug_flatten", "rt_debug_leaf_drop", "rt_debug_cone_skip", "rt_last_error", "rt_abi_version"]
HOST_SYMBOLS = ["rt_preprocess_material", "rt_quat_from_euler", "rt_camera_basis", "rt_pixel_to_ray_transform",
                "rt_instantiate_meshes", "rt_preprocess_tris", "rt_instantiate_meshes_device", "rt_preprocess_tris_device", "rt_scene_bounding_box", "rt_line_intersects_box",
                "rt_tri_intersects_box", "rt_srgb_to_linear", "rt_linear_to_srgb", "rt_bsp_build", "rt_bsp_build_device", "rt_bsp_exhaustive_counts_device", "rt_bsp_destroy", "rt_bsp_counts", "rt_bsp_view", "rt_bvh4_build", "rt_bvh4_destroy", "rt_bvh4_counts", "rt_bvh4_view", "rt_write_ppm",
                "rt_mesh_load", "rt_mesh_counts", "rt_mesh_view", "rt_mesh_destroy", "rt_mesh_last_error"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
                              "(there is no fallback implementation)")
        # librt_amd.so needs the HIP runtime by soname (libamdhip64.so.7).  PyTorch ships its own copy; two
        # HIP runtimes in one process cannot both open the GPU, so when torch is installed its copy is
        # loaded first and the dynamic loader resolves our dependency to that same object.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        L.rt_last_error.restype = ctypes.c_char_p
        L.rt_abi_version.restype = c_u32
        L.rt_mesh_last_error.restype = ctypes.c_char_p
        L.rt_srgb_to_linear.restype = c_f32
        L.rt_srgb_to_linear.argtypes = [c_f32]
        L.rt_linear_to_srgb.restype = c_f32
        L.rt_linear_to_srgb.argtypes = [c_f32]
        for name in DEVICE_SYMBOLS + HOST_SYMBOLS:
            getattr(L, name)        # AttributeError here means header and library disagree
        _lib = L
    return _lib


class RtError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise RtError(f"librt_amd status {rc}: {lib().rt_last_error().decode()}")


def ptr(a):
    return a.ctypes.data_as(vp)
