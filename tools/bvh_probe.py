#!/usr/bin/env python3
"""Development aid: work counters and time of the hierarchy path on the multi-mesh scene, per bounce limit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scenes
from raytracing_amd.render import DeviceScene, RayTraceData
sc = scenes.multi_mesh_scene().preprocess()
ds = DeviceScene(RayTraceData(sc))
ds.build_bvh(4)
w, h = 1920, 1080
p2r = sc.pixel_to_ray_transform(w, h)
import itertools
for spp, seed, merged in itertools.product((2, 4, 8), (0x5EED, 100), ("1", "0")):
    os.environ["RT_AMD_BVH_MERGED"] = merged
    ds.build_bvh(4)
    p = ds.make_params(w, h, sc.camera.position, p2r, spp=spp, bounce_limit=1, flags=8 | 2, seed=seed)
    ds.render(p)
    img, st = ds.render(p)
    print("spp", spp, "seed", seed, "merged", merged, st["segments"], round(st["ms_trace"], 2), round(st["ms_total"], 2), flush=True)
ds.close()
