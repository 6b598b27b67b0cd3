#!/usr/bin/env python3
"""Share of box tests the conservative filter leaves to the exact test, per workload (1080p; samples per pixel as first argument, default 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scenes
from raytracing_amd.render import DeviceScene, RayTraceData
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for name, f in (("builtin", lambda: scenes.builtin_scene(3)), ("cornell", scenes.cornell_scene), ("dragon100k", lambda: scenes.dragon_scale_scene(100_000)),
                ("dragon900k", scenes.dragon_scale_scene), ("multi10m", scenes.multi_mesh_scene)):
    sc = f().preprocess(); ds = DeviceScene(RayTraceData(sc))
    _, st = ds.render(ds.make_params(1920, 1080, sc.camera.position, sc.pixel_to_ray_transform(1920, 1080), spp=spp, flags=1))
    print(f"{name}: {st['box_tests']} box tests, {100.0 * st['box_exact_fallbacks'] / st['box_tests']:.4f} % exact, {st['box_filter_errors']} filter errors", flush=True)
    ds.close()
