#!/usr/bin/env python3
"""Where the traversal kernel's wave time goes, from the diagnostic build (make -C raytracing_amd/csrc diag).
Usage: RT_AMD_LIB=raytracing_amd/librt_amd_diag.so python3 tools/diag.py [workload] [spp] [bounce limit]
(bounce limit 1: camera rays only, the coherent case)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import _lib, scenes
from raytracing_amd.render import DeviceScene, RayTraceData
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bounces = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sc = {"cornell": scenes.cornell_scene, "dragon": scenes.dragon_scale_scene, "multi": scenes.multi_mesh_scene}[name]().preprocess()
ds = DeviceScene(RayTraceData(sc))
p = ds.make_params(1920, 1080, sc.camera.position, sc.pixel_to_ray_transform(1920, 1080), spp=spp, bounce_limit=bounces)
ds.render(p)
d = (ctypes.c_ulonglong * 16)()
L = _lib.lib()
L.rt_diag_read(d, 1)
_, st = ds.render(p)
L.rt_diag_read(d, 0)
v = list(d)
tot = sum(v[0:5])
v[5] -= 0
print(f"{name} {spp} spp, bounce limit {bounces}: {st['segments']} segments")
for k, label in enumerate(["refill", "node walk", "leaf fetch", "triangle scan", "store"]):
    print(f"  {label:14s} {100.0 * v[k] / tot:5.1f} % of wave cycles")
print(f"  walk loop: {v[6] / st['segments'] * 64:8.1f} wave iterations per 64 segments, {v[7] / max(v[6], 1):5.1f} lanes active")
print(f"  tri pass 1: {v[8] / st['segments'] * 64:8.1f} wave iterations per 64 segments, {v[9] / max(v[8], 1):5.1f} lanes active")
print(f"  tri pass 2: {v[10] / st['segments'] * 64:8.1f} wave iterations per 64 segments, {v[11] / max(v[10], 1):5.1f} lanes active")
print(f"  walk iterations in which some lane needs the origin-inside rule: {100.0 * v[14] / max(v[5], 1):5.1f} %, the exact box test: {100.0 * v[15] / max(v[5], 1):5.1f} % of {v[5] / st['segments'] * 64:.0f} wave iterations per 64 segments")
print(f"  outer loop: {v[12] / st['segments'] * 64:8.1f} iterations per 64 segments, {v[13] / max(v[12], 1):5.1f} lanes hold a line")
ds.close()
