#!/usr/bin/env python3
"""Frame time against the batch size (paths per batch) for one workload at a fixed sample count."""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
from raytracing_amd import scenes
from raytracing_amd.render import DeviceScene, RayTraceData
sc = scenes.dragon_scale_scene().preprocess()
ds = DeviceScene(RayTraceData(sc))
p = ds.make_params(1920, 1080, sc.camera.position, sc.pixel_to_ray_transform(1920, 1080), spp=int(sys.argv[1]), flags=2)
ds.render(p)
best = min((ds.render(p)[1] for _ in range(2)), key=lambda s: s["ms_total"])
print(f"batch {os.environ.get('RT_AMD_BATCH_PATHS')}: {best['batches']} batches, total {best['ms_total']:.1f} ms, trace {best['ms_trace']:.1f}, {best['segments'] / best['ms_total'] / 1e3:.1f} Mrays/s", flush=True)
''' % root
for mib in (16, 32, 64, 128, 256):
    env = dict(os.environ, RT_AMD_BATCH_PATHS=str(mib << 20))
    subprocess.run([sys.executable, "-c", code, sys.argv[1] if len(sys.argv) > 1 else "128"], env=env, check=True)
