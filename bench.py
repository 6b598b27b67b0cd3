#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cornell|builtin|dragon]

One step = one render() of the workload frame: camera rays, BSP traversal, ray/triangle tests, GGX
shading and per-pixel accumulation for every sample of every pixel, with the scene already resident
in HBM and the image written to HBM (no PCIe in the timed region).  Metric: Mrays/s = path segments
(closest-hit queries issued by the path loop, src/render.hpp:104) per second, whole job.

N > 1 (launched by torch.distributed.run, one rank per GPU): the frame is cut into horizontal bands
dealt round-robin to the ranks and gathered to rank 0 with one RCCL all-gather per frame, inside the
timed region.  Scaling is weak: the sample count per pixel is 64*N, so every GPU traces the same number
of paths as the single-GPU run.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
FP32_VALU_PEAK_TFLOPS = 157.3

WORKLOADS = {
    # name: (scene factory, width, height, spp per GPU, bounce limit, description)
    "cornell": ("cornell_scene", 1920, 1080, 64, 8,
                "Cornell-box-style room, ~4k triangles, mixed GGX roughness, 1920x1080, 64 spp, 8 bounces"),
    "builtin": ("builtin_scene", 640, 480, 16, 8,
                "reference built-in scene (src/main.cpp:134-187), 640x480, 16 spp, 8 bounces"),
    "dragon": ("dragon_scale_scene", 1920, 1080, 256, 8,
               "displaced closed surface, ~0.9M triangles (dragon-scale stand-in), 1920x1080, 256 spp, 8 bounces"),
    "dragon4k": ("dragon_scale_scene", 3840, 2160, 128, 8,
                 "displaced closed surface, ~0.9M triangles, 3840x2160, 128 spp per GPU (1024 spp on 8 GPUs), 8 bounces"),
    "multi": ("multi_mesh_scene", 1920, 1080, 64, 12,
              "eleven instances of the 0.9M-triangle surface (~9.9M triangles), GGX roughness 0.04-0.1, 1920x1080, "
              "64 spp, 12 bounces"),
}


def build_scene(name):
    from raytracing_amd import scenes
    factory = getattr(scenes, WORKLOADS[name][0])
    return factory().preprocess()


def cpu_baseline(scene, width, height, bounce_limit, target_seconds=12.0):
    """The oracle's CPU restatement of the same path (same scene, resolution, bounce limit, sampler),
    threaded over pixels, on a bounded sample: full frame at 1 spp to calibrate, then as many spp as fit
    in about target_seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import binding
    from helpers import base_from_scene
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    # The oracle traverses the tree the product's builder made (identical to the reference's own build,
    # which is O(nodes x triangles) and would take hours on the large workloads).
    osc = binding.oracle().scene_create_with_tree(base_from_scene(scene), width, height, scene.bsp_tree)
    try:
        t0 = time.perf_counter()
        _, st = osc.render_device_mode(osc.render_params(1, bounce_limit=bounce_limit), threads=threads)
        t1 = time.perf_counter() - t0
        spp = int(max(1, min(64, target_seconds / max(t1, 1e-3))))
        t0 = time.perf_counter()
        _, st = osc.render_device_mode(osc.render_params(spp, bounce_limit=bounce_limit, seed=99), threads=threads)
        dt = time.perf_counter() - t0
    finally:
        osc.close()
    out = {"value": st["segments"] / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"full {width}x{height} frame at {spp} spp ({st['segments']} segments in {dt:.2f} s)",
           "segments_per_path": st["segments"] / st["paths"]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracing_amd import _lib, tiles
    from raytracing_amd.render import DeviceScene, RayTraceData

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # Rehearsal on a box with fewer GPUs than ranks (RT_BENCH_SHARE_GPU=1): every rank renders on GPU 0 and
    # the bands are gathered over gloo through host memory.  The numbers of such a run mean nothing; it
    # exists to exercise the multi-rank code path.
    share_gpu = os.environ.get("RT_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    _, width, height, spp1, bounce_limit, description = WORKLOADS[args.workload]
    spp1 = args.spp or spp1
    spp = spp1 * world                      # weak scaling: constant paths per GPU
    scene = build_scene(args.workload)
    ds = DeviceScene(RayTraceData(scene), device=local_rank)
    p2r = scene.pixel_to_ray_transform(width, height)
    stream = torch.cuda.current_stream().cuda_stream
    bands_per_rank = 8          # tools/band_balance.py: max/mean rank load 1.03 at 8 ranks on the 0.9M-triangle scene (1.07 with 4)
    buf = torch.zeros(tiles.max_rank_rows(height, world, bands_per_rank) * width * 3, dtype=torch.float32, device="cuda")

    def step(seed, flags):
        def render_rect(x0, y0, w, h, out):
            p = ds.make_params(width, height, scene.camera.position, p2r, spp=spp, bounce_limit=bounce_limit,
                               seed=seed, rect=(x0, y0, w, h), flags=flags)
            return ds.render_device(p, out, stream=stream)
        st = tiles.render_sharded(render_rect, width, height, rank, world, buf, bands_per_rank)
        if share_gpu and world > 1:
            torch.cuda.synchronize()
            frame = tiles.gather_frame(buf.cpu(), width, height, rank, world, bands_per_rank, dist)
        else:
            frame = tiles.gather_frame(buf, width, height, rank, world, bands_per_rank, dist)
        return st, frame

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed: algorithmic work per segment for the roofline (box tests B, triangles tested T, shaded hits H).
    work, _ = step(1000, _lib.RT_FLAG_COUNT_WORK) if rank == 0 or world > 1 else ({}, None)
    for i in range(args.warmup):
        step(i, 0)
    fence()
    t0 = time.perf_counter()
    seg = 0
    trace_ms = shade_ms = 0.0
    launches = 0
    for i in range(args.steps):
        st, frame = step(100 + i, _lib.RT_FLAG_PROFILE_KERNELS)
        seg += st["segments"]
        trace_ms += st["ms_trace"]
        shade_ms += st["ms_shade"]
        launches += st["trace_launches"]
    fence()
    dt = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([dt, float(seg)], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, seg = float(tmax[0]), float(t[1])
    if rank == 0:
        finite = bool(torch.isfinite(frame).all())
        # Algorithmic bytes of the traversal kernel: 28 B per node box test (box + index) and 52 B per
        # triangle tested (48 B preprocessed triangle + 4 B id); shading adds 84 B per shaded hit.
        bpt = (28.0 * work["box_tests"] + 52.0 * work["tris_tested"]) / work["segments"]
        fps = (30.0 * work["box_tests"] + 38.0 * work["tris_tested"]) / work["segments"]
        seg_rank0 = seg / world
        trace_s = trace_ms / 1e3
        achieved = bpt * seg_rank0 / trace_s / 1e9 if trace_s > 0 else 0.0
        traffic = None
        traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(traffic_file) and args.workload == "cornell" and world == 1 and not args.spp:
            # HBM bytes per k_trace launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
            # this same command (a PMC pass cannot run inside the timed region).
            traffic = json.load(open(traffic_file)).get("k_trace", {}).get("hbm_bytes_per_launch")
        line = {
            "metric": "Mrays/s (path segments/s) at 1080p" if height == 1080 else "Mrays/s (path segments/s)",
            "value": seg / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (procedural scene, random-free geometry; counter-based sampler)",
            "config": {"workload": description, "name": args.workload, "width": width, "height": height,
                       "spp_total": spp, "spp_per_gpu": spp1, "bounce_limit": bounce_limit,
                       "triangles": int(len(scene.preprocessed_tris.tris)), "tree_depth": ds.depth,
                       "parallelism": f"pixel bands over {world} GPU(s), scene replicated, 1 all-gather per frame",
                       "segments_per_step": seg / args.steps, "image_finite": finite},
            "roofline": {"bound": "hbm", "kernel": "k_trace", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": "HBM bytes per launch (2*FETCH_SIZE+WRITE_SIZE)*1024 from profiles/hbm_traffic.json; "
                                         "the algorithmic bytes are served by L1/L2, so achieved exceeds what HBM moves",
                         "algorithmic_bytes_per_segment": bpt, "segments_per_launch": seg_rank0 / max(launches, 1),
                         "avg_launch_ms": trace_ms / max(launches, 1),
                         "box_tests_per_segment": work["box_tests"] / work["segments"],
                         "tris_tested_per_segment": work["tris_tested"] / work["segments"],
                         "shaded_hits_per_segment": work["shaded_hits"] / work["segments"],
                         "valu_tflops": fps * seg_rank0 / trace_s / 1e12 if trace_s > 0 else 0.0,
                         "valu_frac": (fps * seg_rank0 / trace_s / 1e12 / FP32_VALU_PEAK_TFLOPS) if trace_s > 0 else 0.0,
                         "trace_share_of_kernel_time": trace_ms / max(trace_ms + shade_ms, 1e-9)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene, width, height, bounce_limit)
        print(json.dumps(line), flush=True)
    ds.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
