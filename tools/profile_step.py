#!/usr/bin/env python3
"""One short render per workload, for rocprofv3 (kernel trace or PMC passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scenes
from raytracing_amd.render import DeviceScene, RayTraceData
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = {"cornell": scenes.cornell_scene, "builtin": scenes.builtin_scene,
      "dragon100k": lambda: scenes.dragon_scale_scene(100_000), "dragon": scenes.dragon_scale_scene,
      "multi": scenes.multi_mesh_scene}[name]().preprocess()
ds = DeviceScene(RayTraceData(sc))
p = ds.make_params(1920, 1080, sc.camera.position, sc.pixel_to_ray_transform(1920, 1080), spp=spp)
img, st = ds.render(p)
print(name, st["segments"], st["ms_total"])
ds.close()
