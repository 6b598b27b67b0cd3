#!/usr/bin/env python3
"""How much does ordering secondary rays by (origin cell, direction octant) help the trace kernel?
Builds bounce-1 style rays (origins = primary hit points, random hemisphere directions), then times
rt_intersect_device on them in pixel order and in several sorted orders."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scenes, _lib
from raytracing_amd.render import DeviceScene, RayTraceData

def morton3(x, y, z, bits):
    k = np.zeros_like(x, dtype=np.uint64)
    for b in range(bits):
        k |= ((x >> b) & 1).astype(np.uint64) << (3 * b) | ((y >> b) & 1).astype(np.uint64) << (3 * b + 1) | ((z >> b) & 1).astype(np.uint64) << (3 * b + 2)
    return k

def run(name, sc, w=1920, h=1080):
    sc.preprocess()
    ds = DeviceScene(RayTraceData(sc))
    rng = np.random.default_rng(0)
    spp = 8
    ys, xs = np.mgrid[0:h, 0:w]
    ys, xs = np.tile(ys.ravel(), spp), np.tile(xs.ravel(), spp)
    m = sc.pixel_to_ray_transform(w, h).reshape(3, 3)
    sx, sy = xs + rng.random(len(xs)), ys + rng.random(len(ys))
    v = (m[0][None] * sx[:, None] + m[1][None] * sy[:, None] + m[2][None]).astype(np.float32)
    d0 = v / np.linalg.norm(v, axis=1, keepdims=True)
    o0 = np.tile(sc.camera.position, (len(xs), 1)).astype(np.float32)
    hits = ds.intersect(o0, d0)
    ok = hits["mesh"] != 0xFFFFFFFF
    o1 = hits["point"][ok]
    d1 = rng.standard_normal((len(o1), 3)).astype(np.float32)
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    # keep directions pointing away from the incoming ray side (rough hemisphere)
    flip = (d1 * d0[ok]).sum(1) > 0
    d1[flip] *= -1
    n = len(o1)
    box = sc.bounding_box
    out = torch.zeros(n * 8, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def timeit(o, d, label):
        to, td = torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda()
        ds.intersect_device(to, td, n, out, stream=st); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): ds.intersect_device(to, td, n, out, stream=st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"{name:10s} {label:34s} {n/1e6:.2f}M rays  {ms:8.2f} ms  {n/ms/1e3:8.1f} Mrays/s", flush=True)
    timeit(o0, d0, "primary (pixel order)")
    timeit(o1, d1, "secondary, pixel order")
    perm = rng.permutation(n)
    timeit(o1[perm], d1[perm], "secondary, random order")
    octant = ((d1[:, 0] > 0).astype(np.uint64) | (d1[:, 1] > 0).astype(np.uint64) << 1 | (d1[:, 2] > 0).astype(np.uint64) << 2)
    for bits in (4, 6, 8):
        G = 1 << bits
        q = np.clip(((o1 - box[:3]) / (box[3:] - box[:3]) * G).astype(np.int64), 0, G - 1)
        mort = morton3(q[:, 0], q[:, 1], q[:, 2], bits)
        for label, key in ((f"cell{bits}b|octant", (mort << 3) | octant), (f"octant|cell{bits}b", (octant << (3 * bits)) | mort)):
            order = np.argsort(key, kind="stable")
            timeit(o1[order], d1[order], "secondary, " + label)
    # finer direction: 6 faces x 4x4 grid
    a = np.abs(d1); major = a.argmax(1)
    ds.close()

which = sys.argv[1:] or ["cornell", "builtin", "dragon100k"]
if "cornell" in which: run("cornell", scenes.cornell_scene())
if "builtin" in which: run("builtin", scenes.builtin_scene())
if "dragon100k" in which: run("dragon100k", scenes.dragon_scale_scene(100_000))
if "dragon" in which: run("dragon900k", scenes.dragon_scale_scene(900_000))
if "multi" in which: run("multi10m", scenes.multi_mesh_scene())
