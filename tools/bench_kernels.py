#!/usr/bin/env python3
"""Per-kernel timing of the render path for a few workloads (development aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scenes
from raytracing_amd.render import DeviceScene, RayTraceData

def run(name, sc, w, h, spp, reps=3, bounce_limit=8):
    t0 = time.time()
    sc.preprocess()
    t1 = time.time()
    ds = DeviceScene(RayTraceData(sc))
    print(f"{name}: {len(sc.preprocessed_tris.tris)} triangles, preprocess+tree {t1 - t0:.1f} s, upload {time.time() - t1:.1f} s, "
          f"depth {ds.depth}, {ds.device_bytes / 1e9:.2f} GB on device", flush=True)
    flags = 2
    if os.environ.get("RT_ACCEL") == "bvh":          # the alternative accelerator instead of the BSP tree
        t2 = time.time()
        ds.build_bvh(int(os.environ.get("RT_BVH_LEAF", "4")))
        flags |= 8
        print(f"{name}: hierarchy built and attached in {time.time() - t2:.2f} s", flush=True)
    p = ds.make_params(w, h, sc.camera.position, sc.pixel_to_ray_transform(w, h), spp=spp, flags=flags, bounce_limit=bounce_limit)
    ds.render(p)
    best = None
    for _ in range(reps):
        img, st = ds.render(p)
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    st = best
    print(f"{name:12s} {w}x{h}x{spp}: total {st['ms_total']:8.2f} ms  trace {st['ms_trace']:8.2f}  shade {st['ms_shade']:7.2f}  "
          f"raygen {st['ms_raygen']:5.2f} acc {st['ms_accumulate']:5.2f}  seg {st['segments']/1e6:8.1f}M  "
          f"{st['segments']/st['ms_total']/1e3:8.1f} Mrays/s  (trace-only {st['segments']/st['ms_trace']/1e3:8.1f})", flush=True)
    ds.close()

which = sys.argv[1:] or ["builtin", "cornell", "dragon100k"]
if "builtin" in which: run("builtin", scenes.builtin_scene(), 1920, 1080, 16)
if "cornell" in which: run("cornell", scenes.cornell_scene(), 1920, 1080, 16)
if "dragon100k" in which: run("dragon100k", scenes.dragon_scale_scene(100_000), 1920, 1080, 16)
if "dragon" in which: run("dragon900k", scenes.dragon_scale_scene(900_000), 1920, 1080, 16)
# the multi-mesh benchmark scene: eleven instances of the 0.9 M-triangle surface, low-roughness GGX, 12 bounces
if "multi10m" in which: run("multi10m", scenes.dragon_scale_scene(900_000, copies=11), 1920, 1080, 8, reps=2, bounce_limit=12)
