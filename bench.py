#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload dragon|cornell|builtin|dragon4k|multi]
                    [--scaling weak|strong] [--spp S] [--no-cpu-baseline]

One step = one render() of the workload frame: camera rays, BSP traversal, ray/triangle tests, GGX
shading and per-pixel accumulation for every sample of every pixel, with the scene already resident
in HBM and the image written to HBM (no PCIe in the timed region).  Metric: Mrays/s = path segments
(closest-hit queries issued by the path loop, src/render.hpp:104) per second, whole job.

Default workload: the ~0.9 M-triangle scene at 1920x1080, 256 samples per pixel, 8 bounces.

N > 1 (launched by torch.distributed.run, one rank per GPU): rank 0 prepares the scene and broadcasts it
once as a byte buffer; the frame is cut into horizontal bands dealt round-robin to the ranks and gathered
to rank 0 (RCCL) once per frame, inside the timed region.  --scaling weak (default): the sample count per
pixel is spp*N, so every GPU traces as many paths as the single-GPU run; --scaling strong: the frame's
sample count is fixed and the ranks share it (e.g. dragon4k: 3840x2160 at 1024 spp over 2/4/8 GPUs).

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

HBM_PEAK_GBS = 8000.0                      # MI355X HBM3E
FP32_VALU_PEAK_TFLOPS = 157.3              # = 2 flop x 64 lanes x VALU_PEAK_WAVE_INSTR_PER_S
VALU_PEAK_WAVE_INSTR_PER_S = 157.3e12 / 128.0

WORKLOADS = {
    # name: (scene factory, width, height, spp per GPU (weak), spp of the frame (strong), bounce limit, description)
    "dragon": ("dragon_scale_scene", 1920, 1080, 256, 256, 8,
               "displaced closed surface, ~0.9M triangles (dragon-scale stand-in), 1920x1080, 256 spp, 8 bounces"),
    "cornell": ("cornell_scene", 1920, 1080, 64, 64, 8,
                "Cornell-box-style room, ~4k triangles, mixed GGX roughness, 1920x1080, 64 spp, 8 bounces"),
    "builtin": ("builtin_scene", 640, 480, 16, 16, 8,
                "reference built-in scene (src/main.cpp:134-187), 640x480, 16 spp, 8 bounces"),
    "dragon4k": ("dragon_scale_scene", 3840, 2160, 128, 1024, 8,
                 "displaced closed surface, ~0.9M triangles, 3840x2160, 1024 spp per frame (128 per GPU when weak), 8 bounces"),
    "multi": ("multi_mesh_scene", 1920, 1080, 64, 64, 12,
              "eleven instances of the 0.9M-triangle surface (~9.9M triangles), GGX roughness 0.04-0.1, 1920x1080, "
              "64 spp, 12 bounces"),
}


def build_scene(name, device=None):
    """Instancing, triangle preprocessing and the per-level work of the BSP build run on the GPU."""
    from raytracing_amd import scenes
    factory = getattr(scenes, WORKLOADS[name][0])
    return factory().preprocess(device=device, bsp_device=device)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(scene, width, height, bounce_limit, target_seconds=12.0):
    """The reference's algorithm and parallel decomposition on the host: the oracle's restatement of the same
    path with the 8-triangle AVX2/FMA leaf test (src/geometry.hpp:117-132), threads over pixels
    (src/render.hpp:218) on this process's share of the host cores, same scene, resolution, bounce limit and
    sampler, on a bounded sample: full frame at 1 spp to calibrate, then as many spp as fit in about
    target_seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import binding
    from helpers import base_from_scene
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    # The oracle traverses the tree the product's builder made (identical to the reference's own build,
    # which is O(nodes x triangles) and would take hours on the large workloads).
    osc = binding.oracle().scene_create_with_tree(base_from_scene(scene), width, height, scene.bsp_tree)
    try:
        osc.set_simd_leaves(True)
        t0 = time.perf_counter()
        _, st = osc.render_device_mode(osc.render_params(1, bounce_limit=bounce_limit), threads=threads)
        t1 = time.perf_counter() - t0
        spp = int(max(1, min(64, target_seconds / max(t1, 1e-3))))
        t0 = time.perf_counter()
        _, st = osc.render_device_mode(osc.render_params(spp, bounce_limit=bounce_limit, seed=99), threads=threads)
        dt = time.perf_counter() - t0
        # One thread, a quarter of the rows at 1 spp: the rate of the reference as g++ builds it where TBB is absent (its
        # std::execution::par then runs serially, src/render.hpp:218), for the record beside the threaded number.
        rect = (0, height // 2 - height // 8, width, max(1, height // 4))
        t0 = time.perf_counter()
        _, st1 = osc.render_device_mode(osc.render_params(1, bounce_limit=bounce_limit, seed=98, rect=rect), threads=1)
        dt1 = time.perf_counter() - t0
    finally:
        osc.close()
    return {"value": st["segments"] / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "single_thread": {"value": st1["segments"] / dt1 / 1e6, "unit": "Mrays/s", "cores": 1,
                              "sample": f"rows {rect[1]}..{rect[1] + rect[3] - 1} at 1 spp ({st1['segments']} segments in {dt1:.2f} s)"},
            "port_variant": "the reference's 8-triangle AVX2+FMA leaf test (src/geometry.hpp:117-132), OpenMP threads over pixels",
            "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(),
            "sample": f"full {width}x{height} frame at {spp} spp ({st['segments']} segments in {dt:.2f} s), "
                      f"8-triangle AVX2+FMA leaf test, {threads} threads over pixels",
            "segments_per_path": st["segments"] / st["paths"]}


def image_rel_rmse(device, spp=4096):
    """The metric's second half: the image against the reference's converged render.  The reference's own 16384-spp
    render of its built-in scene at 160x120 (tests/golden/builtin_converged_160x120.npz, written from the compiled
    reference: render(), src/render.hpp:215-234) against a GPU render of the same scene and size at `spp` samples
    (another sampler, so the two are independent estimates of one image): relative RMSE ||a - b|| / ||b|| over the linear
    HDR frame, beside the value two unbiased estimates of this scene are expected to show, 1.17 sqrt(1/n1 + 1/n2)
    (the constant was measured with the reference's own sampler at 64 to 1024 spp).  Untimed."""
    from raytracing_amd import scenes
    from raytracing_amd.render import DeviceScene, RayTraceData
    path = os.path.join(ROOT, "tests", "golden", "builtin_converged_160x120.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    ref = g["image"].astype(np.float64)
    sc = scenes.builtin_scene(3).preprocess()
    ds = DeviceScene(RayTraceData(sc), device=device)
    try:
        img, _ = ds.render(ds.make_params(160, 120, sc.camera.position, sc.pixel_to_ray_transform(160, 120), spp=spp, seed=2024))
    finally:
        ds.close()
    img = img.astype(np.float64)
    ref_spp = int(g["spp"][0])
    return {"value": float(np.sqrt(((img - ref) ** 2).sum() / (ref ** 2).sum())),
            "expected_for_unbiased_estimates": 1.17 * float(np.sqrt(1.0 / spp + 1.0 / ref_spp)),
            "mean_radiance_ratio": float(img.mean() / ref.mean()), "spp": spp, "reference_spp": ref_spp,
            "scene": "reference built-in scene (src/main.cpp:134-187), 160x120, linear HDR",
            "reference": "compiled reference's render() at 16384 spp (tests/golden/builtin_converged_160x120.npz)"}


def roofline(workload, work, seg_per_launch, launch_ms, trace_share, accel="bsp"):
    """Ceilings for k_trace, the dominant kernel.  Live inputs: segments per launch and the average launch
    duration (HIP events around every k_trace launch of the timed steps) and the work counters of an untimed
    counting step.  Counter inputs: per-segment HBM bytes, VALU instructions and cache accesses of k_trace from
    the committed rocprofv3 --pmc passes of this workload (profiles/roofline_<workload>.json, made by
    tools/roofline_passes.sh; a PMC pass cannot run inside the timed region), and the ceilings measured by the
    calibration kernels of tools/ubench/peaks.hip in the same passes."""
    seg_per_s = seg_per_launch / (launch_ms * 1e-3) if launch_ms > 0 else 0.0
    if accel == "bvh":
        # a node visit reads 112 bytes (four child boxes and references) and costs about 20 flops per child box
        bpt = (112.0 * work["inode_visits"] + 52.0 * work["tris_tested"]) / work["segments"]
        fps = (20.0 * work["box_tests"] + 38.0 * work["tris_tested"]) / work["segments"]
    else:
        bpt = (28.0 * work["box_tests"] + 52.0 * work["tris_tested"]) / work["segments"]
        fps = (30.0 * work["box_tests"] + 38.0 * work["tris_tested"]) / work["segments"]
    ceilings = {
        "algorithmic_zero_reuse": {
            "note": ("112 B per node visit" if accel == "bvh" else "28 B per node box test")
                    + " + 52 B per triangle tested, every byte counted as if it came from HBM; "
                    "informational only: these bytes are served by L1/L2, so this is not a bound",
            "bytes_per_segment": bpt, "gbytes_per_s": bpt * seg_per_s / 1e9, "flops_per_segment": fps,
            "tflops": fps * seg_per_s / 1e12, "frac_of_fp32_valu_peak": fps * seg_per_s / 1e12 / FP32_VALU_PEAK_TFLOPS},
    }
    out = {"kernel": "k_trace_bvh" if accel == "bvh" else "k_trace", "avg_launch_ms": launch_ms, "segments_per_launch": seg_per_launch,
           "segments_per_s_in_kernel": seg_per_s, "trace_share_of_kernel_time": trace_share,
           "box_tests_per_segment": work["box_tests"] / work["segments"],
           "tris_tested_per_segment": work["tris_tested"] / work["segments"],
           "node_visits_per_segment": work["inode_visits"] / work["segments"],
           "shaded_hits_per_segment": work["shaded_hits"] / work["segments"],
           # the counting step evaluates every conservative shortcut of the traversal beside the exact test: anything but 0 is a bug
           "shortcut_disagreements_in_counting_step": work.get("box_filter_errors", 0),
           "counting_step_segments": work["segments"]}
    # counter passes are per scene: the 4K workload shares the 1080p one's
    pmc_file = os.path.join(ROOT, "profiles", f"roofline_{ {'dragon4k': 'dragon'}.get(workload, workload) }{'_bvh' if accel == 'bvh' else ''}.json")
    traffic = None
    if os.path.exists(pmc_file):
        pmc = json.load(open(pmc_file))
        d = pmc.get("k_trace_per_segment", {})
        peaks = pmc.get("peaks_unprofiled", {})
        out["counters_file"] = os.path.relpath(pmc_file, ROOT)
        out["counters_spp"] = pmc.get("spp")          # the counter passes render the workload's frame at this many samples
        sq = next((v.get("sq", {}) for k, v in pmc.get("kernels", {}).items() if k.startswith(d.get("kernel", "k_trace"))), {})
        if sq.get("SQ_BUSY_CU_CYCLES"):
            # both counters tick once per four clocks; their ratio is the mean number of waves resident per SIMD
            out["occupancy_waves_per_simd"] = sq["SQ_WAVE_CYCLES"] / sq["SQ_BUSY_CU_CYCLES"]
        if "hbm_bytes_per_segment" in d:
            traffic = d["hbm_bytes_per_segment"] * seg_per_launch
            a = d["hbm_bytes_per_segment"] * seg_per_s / 1e9
            ceilings["hbm"] = {"achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                               "bytes_per_segment": d["hbm_bytes_per_segment"],
                               "note": "measured HBM bytes: (2*FETCH_SIZE + WRITE_SIZE) KiB of k_trace, separate --pmc passes"}
        if "valu_wave_instr_per_segment" in d:
            a = d["valu_wave_instr_per_segment"] * seg_per_s
            lanes = d["active_lanes_per_valu_instr"] / 64.0
            ceilings["valu"] = {"achieved": a / 1e9, "peak": VALU_PEAK_WAVE_INSTR_PER_S / 1e9, "unit": "G wave-instructions/s",
                                "frac": a / VALU_PEAK_WAVE_INSTR_PER_S, "active_lane_fraction": lanes,
                                "lane_slot_frac": a / VALU_PEAK_WAVE_INSTR_PER_S * lanes,
                                "measured_peak_fma_stream": peaks.get("valu_wave_instr_per_s", 0) / 1e9,
                                "note": "SQ_INSTS_VALU per segment x segments/s against 2 wave64 instructions per cycle and CU "
                                        "at 2.4 GHz (the 157.3 TFLOP/s FP32 vector peak); lanes = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU"}
        if "l1_accesses_per_segment" in d and peaks.get("gather16_lane_loads_per_s_l1"):
            a = d["l1_accesses_per_segment"] * seg_per_s
            miss1, hit2 = d["l1_miss_rate"], d["l2_hit_rate"]
            p1, p2, p3 = (peaks[f"gather16_lane_loads_per_s_{k}"] for k in ("l1", "l2", "mall"))
            mix_rate = 1.0 / ((1.0 - miss1) / p1 + miss1 * hit2 / p2 + miss1 * (1.0 - hit2) / p3)
            units = 256.0 / 8.0          # TA / TD units per XCD; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            ceilings["texture_path"] = {
                "achieved": a / 1e9, "peak": p1 / 1e9, "unit": "G lane-loads/s", "frac": a / p1,
                "l1_miss_rate": miss1, "l2_hit_rate": hit2,
                "rate_of_calibration_loads_at_this_hit_mix": mix_rate / 1e9,
                "td_busy": d.get("td_busy_per_gui_cycle", 0.0) / units, "ta_busy": d.get("ta_busy_per_gui_cycle", 0.0) / units,
                "td_busy_of_calibration_kernel": pmc.get("peak_gather_td_busy_per_gui_cycle", 0.0) / units,
                "ta_busy_of_calibration_kernel": pmc.get("peak_gather_ta_busy_per_gui_cycle", 0.0) / units,
                "note": "TCP_TOTAL_CACHE_ACCESSES (one per lane of a scattered 16-byte load) x segments/s against the highest "
                        "rate at which the chip serves scattered 16-byte loads: tools/ubench/peaks.hip, all 64 lanes active, "
                        "4 loads in flight per lane, 16 KiB table (every load an L1 hit).  The same loads from a 1 MiB (L2) or "
                        "64 MiB (Infinity Cache) table run 3.7x and 17x slower; weighted by this kernel's L1 miss and L2 hit "
                        "rates that gives rate_of_calibration_loads_at_this_hit_mix, an estimate, not a bound.  td_busy / "
                        "ta_busy: fraction of cycles the texture data / address units are busy, beside the calibration kernel's"}
    bound = max((k for k in ceilings if "frac" in ceilings[k]), key=lambda k: ceilings[k]["frac"], default=None)
    if bound is None:
        # no counter file for this workload: only the FP32 VALU fraction of the algorithmic flops can be stated
        z = ceilings["algorithmic_zero_reuse"]
        out.update({"bound": "valu", "achieved": z["tflops"], "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": z["frac_of_fp32_valu_peak"], "traffic": None})
    else:
        b = ceilings[bound]
        out.update({"bound": {"hbm": "hbm", "valu": "valu", "texture_path": "texture-path"}[bound], "achieved": b["achieved"],
                    "peak": b["peak"], "unit": b["unit"], "frac": b["frac"], "traffic": traffic,
                    "traffic_note": "measured HBM bytes per k_trace launch (PMC), against "
                                    f"{bpt * seg_per_launch:.3e} algorithmic zero-reuse bytes per launch"})
    out["ceilings"] = ceilings
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (per GPU when weak, per frame when strong)")
    ap.add_argument("--roulette", action="store_true", help="render with RT_FLAG_RUSSIAN_ROULETTE")
    ap.add_argument("--lobes", action="store_true", help="render with RT_FLAG_LOBE_SAMPLING (per-lobe sampling)")
    ap.add_argument("--sort", action="store_true", help="render with RT_FLAG_SORT_RAYS (live paths re-ordered per bounce)")
    ap.add_argument("--accel", default="bsp", choices=["bsp", "bvh"],
                    help="bsp: the reference's tree (results bit-identical to the reference's traversal); bvh: the "
                         "alternative accelerator, a 4-wide bounding volume hierarchy (RT_FLAG_ACCEL_BVH)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra measurement of the alternative accelerator")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracing_amd import _lib, tiles
    from raytracing_amd.render import DeviceScene, RayTraceData, serialize_scene

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
    # scene bytes and bands travel over gloo through host memory.  The numbers of such a run mean nothing; it
    # exists to exercise the multi-rank code path.
    share_gpu = os.environ.get("RT_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    comm_device = "cpu" if share_gpu else "cuda"
    # RT_BENCH_FORCE_DIST=1: go through the multi-rank code (process group, scene broadcast, band gather, reductions) even
    # with one rank, so that the RCCL path can be executed on a box with a single GPU.
    distributed = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")          # only used when not started by torch.distributed.run
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    _, width, height, spp_weak, spp_strong, bounce_limit, description = WORKLOADS[args.workload]
    if args.scaling == "weak":
        spp1 = args.spp or spp_weak
        spp = spp1 * world                  # constant paths per GPU
    else:
        spp = args.spp or spp_strong        # fixed frame; each GPU traces 1/world of it
        spp1 = spp / world

    # ---- scene: prepared once on rank 0, broadcast as bytes, uploaded by every rank ---------------------------
    t_setup = time.perf_counter()
    scene = None
    if rank == 0:
        scene = build_scene(args.workload, local_rank)
        rtd = RayTraceData(scene)
        header = np.concatenate([scene.camera.position, scene.pixel_to_ray_transform(width, height),
                                 [len(scene.preprocessed_tris.tris)]]).astype(np.float64)
    if distributed:
        blob = tiles.broadcast_bytes(serialize_scene(rtd) if rank == 0 else None, rank, dist, comm_device)
        head = torch.from_numpy(header).to(comm_device) if rank == 0 else torch.zeros(13, dtype=torch.float64, device=comm_device)
        dist.broadcast(head, src=0)
        header = head.cpu().numpy()
        # rank 0 keeps its own description; every other rank (and rank 0 of a forced single-rank run) uploads from the bytes
        ds = DeviceScene(rtd, device=local_rank) if rank == 0 and world > 1 else DeviceScene.from_blob(blob, device=local_rank)
        scene_bytes = int(blob.size)
        del blob
    else:
        ds = DeviceScene(rtd, device=local_rank)
        scene_bytes = 0
    cam_pos, p2r, triangles = header[0:3].astype(np.float32), header[3:12].astype(np.float32), int(header[12])
    bvh_build_s = 0.0
    if args.accel == "bvh":
        # every rank builds the hierarchy from the scene bytes it holds (deterministic, so all ranks hold the same one)
        t_bvh = time.perf_counter()
        ds.build_bvh(4)
        bvh_build_s = time.perf_counter() - t_bvh
    setup_s = time.perf_counter() - t_setup

    stream = torch.cuda.current_stream().cuda_stream
    # One rectangle per GPU when alone; otherwise 8 bands per rank, dealt round-robin and rendered by one call
    # per frame (tools/band_balance.py: max/mean rank load 1.03 at 8 ranks on the 0.9M-triangle scene, 1.07 with 4).
    bands_per_rank = 8 if distributed else 1
    render_buf = torch.zeros(tiles.max_rank_rows(height, world, bands_per_rank) * width * 3, dtype=torch.float32, device="cuda")
    gatherer = tiles.FrameGatherer(width, height, rank, world, bands_per_rank, dist,
                                   buffer=(torch.zeros(render_buf.numel()) if share_gpu and world > 1 else render_buf),
                                   always_gather=distributed)
    extra = ((_lib.RT_FLAG_RUSSIAN_ROULETTE if args.roulette else 0) | (_lib.RT_FLAG_ACCEL_BVH if args.accel == "bvh" else 0)
             | (_lib.RT_FLAG_LOBE_SAMPLING if args.lobes else 0) | (_lib.RT_FLAG_SORT_RAYS if args.sort else 0))

    def step(seed, flags, spp_step):
        def render_rows(y0, rows, row_group, row_stride, out):
            p = ds.make_params(width, height, cam_pos, p2r, spp=spp_step, bounce_limit=bounce_limit, seed=seed,
                               rect=(0, y0, width, rows), flags=flags | extra, row_group=row_group, row_stride=row_stride)
            return ds.render_device(p, out, stream=stream)
        # all of this rank's bands in one call (interleaved row groups)
        st = tiles.render_interleaved(render_rows, width, height, rank, world, render_buf, bands_per_rank)
        if share_gpu and world > 1:
            torch.cuda.synchronize()
            gatherer.buffer.copy_(render_buf.cpu())
        return st, gatherer.gather()

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed: work per segment for the roofline (box tests, triangles tested, shaded hits), on one batch.
    work, _ = step(1000, _lib.RT_FLAG_COUNT_WORK, min(spp, 8))
    for i in range(args.warmup):
        step(i, 0, spp)
    fence()
    t0 = time.perf_counter()
    seg = 0
    trace_ms = shade_ms = 0.0
    launches = 0
    for i in range(args.steps):
        st, frame = step(100 + i, _lib.RT_FLAG_PROFILE_KERNELS, spp)
        seg += st["segments"]
        trace_ms += st["ms_trace"]
        shade_ms += st["ms_shade"]
        launches += st["trace_launches"]
    fence()
    dt = time.perf_counter() - t0

    seg_rank = seg
    if distributed:
        t = torch.tensor([dt, float(seg)], dtype=torch.float64, device=comm_device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, seg = float(tmax[0]), float(t[1])
    if rank == 0:
        finite = bool(torch.isfinite(frame).all())
        line = {
            "metric": "Mrays/s (path segments/s) at 1080p" if height == 1080 else "Mrays/s (path segments/s)",
            "value": seg / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (procedural scene, random-free geometry; counter-based sampler)",
            "config": {"workload": description, "name": args.workload, "width": width, "height": height,
                       "spp_total": spp, "spp_per_gpu": spp1, "bounce_limit": bounce_limit,
                       "russian_roulette": bool(args.roulette), "per_lobe_sampling": bool(args.lobes), "sorted_rays": bool(args.sort), "triangles": triangles, "tree_depth": ds.depth,
                       "accelerator": "reference BSP tree" if args.accel == "bsp" else "4-wide bounding volume hierarchy (not the reference's accelerator)",
                       "bvh_build_s": bvh_build_s,
                       "parallelism": (f"pixel bands over {world} GPUs, scene broadcast once from rank 0 "
                                       f"({scene_bytes / 1e6:.1f} MB), bands gathered to rank 0 every frame") if distributed
                       else "one GPU, one rectangle",
                       "scene_setup_s": setup_s, "segments_per_step": seg / args.steps, "image_finite": finite},
            "roofline": roofline(args.workload, work, seg_rank / max(launches, 1), trace_ms / max(launches, 1),
                                 trace_ms / max(trace_ms + shade_ms, 1e-9), args.accel),
        }
        if world == 1 and args.accel == "bsp" and not args.no_alt:
            # Beside the headline (the reference's own tree, results bit-identical to the reference's traversal): the
            # same frames through the alternative accelerator, which returns the same nearest triangle on all but a few
            # rays in a hundred thousand.  Reported separately; `value` above is never taken from it.
            t_bvh = time.perf_counter()
            ds.build_bvh(4)
            build_s = time.perf_counter() - t_bvh
            extra |= _lib.RT_FLAG_ACCEL_BVH
            step(7, 0, spp)
            fence()
            t1 = time.perf_counter()
            seg_alt = 0
            for i in range(args.steps):
                st, frame = step(100 + i, 0, spp)
                seg_alt += st["segments"]
            fence()
            dt_alt = time.perf_counter() - t1
            line["alt_accelerator"] = {
                "accelerator": "4-wide bounding volume hierarchy, 64-byte quantised nodes (RT_FLAG_ACCEL_BVH); not the reference's",
                "value": seg_alt / dt_alt / 1e6, "unit": "Mrays/s", "steps": args.steps, "ms_per_step": dt_alt / args.steps * 1e3,
                "build_and_upload_s": build_s, "image_finite": bool(torch.isfinite(frame).all())}
        if world == 1:
            rmse = image_rel_rmse(local_rank)
            if rmse is not None:
                line["image_rel_rmse"] = rmse["value"]
                line["image_rel_rmse_detail"] = rmse
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene, width, height, bounce_limit)
        print(json.dumps(line), flush=True)
    ds.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
