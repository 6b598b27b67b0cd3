#!/usr/bin/env python3
"""Traversal time with and without the leaf triangle bounds over a range of scene sizes (development aid for the
size rule in rt_scene_create).  Run on the GPU box from the repository root."""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from raytracing_amd import scenes
    from raytracing_amd.render import DeviceScene, RayTraceData
    n = int(sys.argv[1])
    sc = scenes.dragon_scale_scene(n).preprocess(device=0, bsp_device=0)
    ds = DeviceScene(RayTraceData(sc))
    p = ds.make_params(1920, 1080, sc.camera.position, sc.pixel_to_ray_transform(1920, 1080), spp=16, flags=2)
    ds.render(p)
    best = min(ds.render(p)[1]["ms_trace"] for _ in range(3))
    print(f"{n} triangles, {sc.bsp_tree.inode_count * 2 * 32 / 1e6:.1f} MB of node records, RT_AMD_LEAF_BOUNDS={os.environ.get('RT_AMD_LEAF_BOUNDS')}: trace {best:.2f} ms", flush=True)
    ds.close()
else:
    for n in [int(a) for a in os.environ.get("SIZES", "50000,200000,300000,450000,650000").split(",")]:
        for v in ("0", "1"):
            subprocess.run([sys.executable, __file__, str(n)], env=dict(os.environ, RT_AMD_LEAF_BOUNDS=v), check=True)
