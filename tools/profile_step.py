#!/usr/bin/env python3
"""One production render per workload at 1920x1080 (or the workload's own size), for rocprofv3 kernel-trace
and PMC passes.  Usage: profile_step.py <workload> <spp> [stats.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scenes
from raytracing_amd.render import DeviceScene, RayTraceData
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bounce = 12 if name == "multi" else 8
sc = {"cornell": scenes.cornell_scene, "builtin": scenes.builtin_scene,
      "dragon100k": lambda: scenes.dragon_scale_scene(100_000), "dragon": scenes.dragon_scale_scene,
      "multi": scenes.multi_mesh_scene}[name]().preprocess()
ds = DeviceScene(RayTraceData(sc))
flags = 0
if os.environ.get("RT_ACCEL") == "bvh":          # the alternative accelerator instead of the BSP tree
    ds.build_bvh(int(os.environ.get("RT_BVH_LEAF", "4")))
    flags = 8
p = ds.make_params(1920, 1080, sc.camera.position, sc.pixel_to_ray_transform(1920, 1080), spp=spp, bounce_limit=bounce, flags=flags)
img, st = ds.render(p)
print(name, st["segments"], st["ms_total"])
if len(sys.argv) > 3:
    json.dump({"workload": name, "spp": spp, "segments": st["segments"], "paths": st["paths"], "trace_launches": st["trace_launches"]},
              open(sys.argv[3], "w"))
ds.close()
