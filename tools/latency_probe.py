#!/usr/bin/env python3
"""Time of rt_intersect_device for very small batches (launch and critical-path latency of the persistent
traversal kernel).  Usage: latency_probe.py [workload]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from raytracing_amd import scenes
from raytracing_amd.render import DeviceScene, RayTraceData
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
sc = {"cornell": scenes.cornell_scene, "dragon": scenes.dragon_scale_scene, "dragon100k": lambda: scenes.dragon_scale_scene(100_000)}[name]().preprocess()
ds = DeviceScene(RayTraceData(sc))
rng = np.random.default_rng(0)
box = sc.bounding_box
for n in (1, 64, 1024, 16384, 262144, 4194304):
    o = rng.uniform(box[:3], box[3:], (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    out = torch.zeros(n * 8, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): ds.intersect_device(to, td, n, out, stream=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ds.intersect_device(to, td, n, out, stream=st)
    e1.record(); torch.cuda.synchronize()
    print(f"{name} depth {ds.depth}: n = {n:8d}: {e0.elapsed_time(e1) / 10 * 1e3:9.1f} us per call", flush=True)
ds.close()
