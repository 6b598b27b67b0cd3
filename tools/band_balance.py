#!/usr/bin/env python3
"""Load balance of the row-band sharding: per-row cost (kernel time of 8-row strips) of a workload, then the
max/mean rank load for several band counts and world sizes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scenes, tiles
from raytracing_amd.render import DeviceScene, RayTraceData
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
sc = {"cornell": scenes.cornell_scene, "builtin": scenes.builtin_scene, "dragon": scenes.dragon_scale_scene}[name]().preprocess()
ds = DeviceScene(RayTraceData(sc))
w, h, strip = 1920, 1080, 8
p2r = sc.pixel_to_ray_transform(w, h)
cost = np.zeros(h)
for y0 in range(0, h, strip):
    rows = min(strip, h - y0)
    p = ds.make_params(w, h, sc.camera.position, p2r, spp=16, rect=(0, y0, w, rows), flags=2)
    ds.render(p)
    _, st = ds.render(p)
    cost[y0:y0 + rows] = (st["ms_trace"] + st["ms_shade"]) / rows
ds.close()
print(name, "rows cost min/mean/max", cost.min(), cost.mean(), cost.max())
for world in (2, 4, 8):
    for bpr in (1, 2, 4, 8, 16):
        loads = [sum(cost[y0:y0 + r].sum() for _, y0, r in tiles.rank_bands(h, world, k, bpr)) for k in range(world)]
        print(f"world {world} bands/rank {bpr:2d}: max/mean = {max(loads) / np.mean(loads):.3f}")
