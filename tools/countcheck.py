#!/usr/bin/env python3
"""One full benchmark frame through the kernels that count work and cross-check every conservative shortcut of the
traversal (slab filter against the six-face box test, leaf normal cones against every triangle's facing test, leaf
triangle bounds against every accepted hit): the frames of bench.py hold a thousand times as many segments as the
test suite, which is what it takes to meet events of probability 1e-8 per segment.
Usage (GPU box): python3 tools/countcheck.py <workload> [spp] [seed] > profiles/r03_countcheck_<workload>.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracing_amd import _lib
from raytracing_amd.render import DeviceScene, RayTraceData

name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
_, width, height, spp, _, bounce_limit, description = bench.WORKLOADS[name]
if len(sys.argv) > 2:
    spp = int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 100
scene = bench.build_scene(name, 0)
ds = DeviceScene(RayTraceData(scene))
p = ds.make_params(width, height, scene.camera.position, scene.pixel_to_ray_transform(width, height), spp=spp,
                   bounce_limit=bounce_limit, seed=seed, flags=_lib.RT_FLAG_COUNT_WORK)
t0 = time.perf_counter()
img, st = ds.render(p)
dt = time.perf_counter() - t0
q = ds.make_params(width, height, scene.camera.position, scene.pixel_to_ray_transform(width, height), spp=spp,
                   bounce_limit=bounce_limit, seed=seed)
img2, st2 = ds.render(q)
out = {"workload": description, "name": name, "width": width, "height": height, "spp": spp, "seed": seed, "bounce_limit": bounce_limit,
       "segments": st["segments"], "box_tests": st["box_tests"], "tris_tested": st["tris_tested"], "leaf_visits": st["leaf_visits"],
       "box_exact_fallbacks": st["box_exact_fallbacks"], "box_filter_errors": st["box_filter_errors"],
       "seconds_counting_render": dt,
       "image_equals_production_render_bit_for_bit": bool((img.view("uint32") == img2.view("uint32")).all()),
       "segments_of_production_render": st2["segments"],
       "note": "box_filter_errors counts disagreements of any shortcut (slab filter, normal cones, triangle bounds) with the "
               "exact evaluation over the whole frame; the production render of the same frame (shortcuts active) is compared "
               "with the counting render (every node and triangle visited as the reference does) bit for bit"}
print(json.dumps(out, indent=1))
ds.close()
