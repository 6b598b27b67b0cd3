#!/usr/bin/env python3
"""Build time of the BSP tree on the host (16 threads, 1 thread) and with the per-level work on the GPU
(rt_bsp_build_device), for the 0.9 M- and 9.9 M-triangle scenes.  Run on the GPU box from the repository root."""
import sys, time
sys.path.insert(0, '.')
from raytracing_amd import scene as S, scenes
for name, mk in (("0.9M", lambda: scenes.dragon_scale_scene(900_000)), ("9.9M", scenes.multi_mesh_scene)):
    sc = mk(); sc.preprocess()
    im = sc.instantiated_meshes
    args = (im.vertex_positions, im.vertex_begin, sc.meshes.tris, sc.meshes.tri_ranges, sc.models.meshes, sc.bounding_box, 32, 60)
    for label, extra in (("host 16 threads", (0,)), ("host 1 thread", (1,)), ("device", (0, 0)), ("device again", (0, 0))):
        t0 = time.perf_counter(); t = S.BSPTree(*args, *extra); dt = time.perf_counter() - t0
        print(f"{name} {label}: {dt:.3f} s  inodes {t.inode_count} refs {t.ref_count} depth {t.depth}", flush=True)
