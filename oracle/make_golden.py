"""TEST INFRASTRUCTURE ONLY: writes the golden vectors under tests/golden/ from the compiled reference
(oracle/_ref/libref.so, built by oracle/build_ref.sh from /root/reference).  Run in the development
container only:  python oracle/make_golden.py

Every file holds inputs and the reference's outputs for them -- data, no reference code.
Ray directions are stored as inputs because they depend on the generating machine's libm.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as B  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def base_dict(base):
    return dict(positions=base.positions, normals=base.normals, tris=base.tris,
                mesh_vertex_ranges=base.mesh_vertex_ranges, mesh_tri_ranges=base.mesh_tri_ranges,
                materials=base.materials, transforms=base.transforms, model_meshes=base.model_meshes,
                model_materials=base.model_materials, camera=base.camera,
                inflate_mode=np.array([base.inflate_mode], np.int32))


def scene_dump(scene, base, prefix=""):
    d = {}
    d.update({prefix + "base_" + k: v for k, v in base_dict(base).items()})
    d.update({prefix + "pre_" + k: v for k, v in scene.preprocessed().items()})
    d.update({prefix + "bsp_" + k: v for k, v in scene.bsp().items()})
    return d


def random_rays(rng, box, n):
    lo, hi = box[:3], box[3:]
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.standard_normal((n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # a few axis-parallel and grazing directions: the six-face box test treats them specially
    d[:6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    d[6:12] = d[:6] + np.float32(1e-7)
    return o, d


def main():
    R = B.reference()
    assert R is not None, "build oracle/_ref/libref.so first (oracle/build_ref.sh)"
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20240611)

    # ---- built-in scene (src/main.cpp:134-187), 640x480 ------------------------------------------
    W, H = 640, 480
    rs = R.scene_builtin(3, W, H)
    base = rs.base()
    np.savez_compressed(os.path.join(OUT, "builtin_scene.npz"), **scene_dump(rs, base))
    pre = rs.preprocessed()

    # (a) primary rays through pixel centres: a 64x48 grid plus the nine pixels recorded in the survey
    xs, ys = np.meshgrid(np.arange(5, W, 10), np.arange(5, H, 10))
    pix = np.stack([xs.ravel(), ys.ravel()], 1)
    known = np.array([[0, 0], [320, 240], [100, 400], [500, 150], [350, 200], [40, 150], [600, 300], [320, 470],
                      [200, 90]])
    pix = np.concatenate([known, pix])
    xy = (pix + 0.5).astype(np.float32)
    dirs = rs.camera_rays(xy)
    org = np.tile(pre["camera_position"], (len(xy), 1)).astype(np.float32)
    f, ids = rs.intersect(org, dirs, 1e-3, True)
    # full-frame statistics (all 307200 pixel centres)
    ax, ay = np.meshgrid(np.arange(W), np.arange(H))
    axy = np.stack([ax.ravel() + 0.5, ay.ravel() + 0.5], 1).astype(np.float32)
    adirs = rs.camera_rays(axy)
    af, aids = rs.intersect(np.tile(pre["camera_position"], (len(axy), 1)).astype(np.float32), adirs, 1e-3, True)
    mesh_hist = np.bincount(aids[:, 0][aids[:, 0] != 0xFFFFFFFF], minlength=30)
    np.savez_compressed(os.path.join(OUT, "builtin_primary_hits.npz"), pixels=pix, sample_xy=xy, directions=dirs,
                        origins=org, hit_floats=f, hit_ids=ids, full_frame_mesh_histogram=mesh_hist,
                        full_frame_t_sum=np.array([af[:, 0].astype(np.float64).sum()]))

    # (b) secondary rays from random origins inside the scene box, both surface modes, two tMin values
    o2, d2 = random_rays(rng, pre["root_box"], 4096)
    out = dict(origins=o2, directions=d2)
    for name, front, tmin in (("front", True, 1e-3), ("all", False, 1e-3), ("front_tmin1", True, 1.0)):
        ff, ii = rs.intersect(o2, d2, tmin, front)
        out["floats_" + name], out["ids_" + name] = ff, ii
    np.savez_compressed(os.path.join(OUT, "builtin_secondary_hits.npz"), **out)

    # (c) predicate truth tables
    n = 6000
    bo, bd = random_rays(rng, np.array([-3, -3, -3, 3, 3, 3], np.float32), n)
    blo = rng.uniform(-2, 1, (n, 3)).astype(np.float32)
    bhi = (blo + rng.uniform(0, 2.5, (n, 3))).astype(np.float32)
    boxes = np.concatenate([blo, bhi], 1)
    bo[:200] = blo[:200]                                  # origins on a box corner / face
    bo[200:400, 0] = bhi[200:400, 0]
    lb = R.line_intersects_box(bo, bd, boxes)
    tris = rng.uniform(-2, 2, (n, 9)).astype(np.float32)
    ctr, ext = (blo + bhi) / 2, bhi - blo
    tris[:3000] = (np.tile(ctr[:3000], (1, 3)) + rng.uniform(-0.8, 0.8, (3000, 9)) * np.tile(ext[:3000], (1, 3))).astype(np.float32)
    tris[3000:3500, 0::3] = blo[3000:3500, 0:1]           # a vertex coordinate exactly on a box plane
    tb = R.tri_intersects_box(tris, boxes)
    np.savez_compressed(os.path.join(OUT, "predicates.npz"), line_origins=bo, line_directions=bd, boxes=boxes,
                        line_box=lb, tris=tris, tri_box=tb)

    # (d) rayTrace radiance for explicit FastRNG seeds
    sel = rng.integers(0, len(axy), 3000)
    seeds = rng.integers(0, 2 ** 63, 3000, dtype=np.uint64)
    rad = rs.ray_trace(np.tile(pre["camera_position"], (3000, 1)).astype(np.float32), adirs[sel], seeds)
    np.savez_compressed(os.path.join(OUT, "builtin_ray_trace.npz"), directions=adirs[sel], seeds=seeds, radiance=rad,
                        origins=np.tile(pre["camera_position"], (3000, 1)).astype(np.float32))
    rs.close()

    # (e) converged render of the built-in scene at 160x120 (its own pixelToRay matrix)
    rs2 = R.scene_builtin(3, 160, 120)
    img = rs2.render(16384, 12345)
    small = rs2.render(2, 777)
    np.savez_compressed(os.path.join(OUT, "builtin_converged_160x120.npz"), image=img, spp=np.array([16384]),
                        image_2spp_seed777=small, pixel_to_ray=rs2.preprocessed()["pixel_to_ray"])
    rs2.close()

    # (f) a denser variant of the built-in scene (DIVISOR=5: 128 models, 1506 triangles) for the tree builder
    rs3 = R.scene_builtin(5, 320, 240)
    base3 = rs3.base()
    d3 = scene_dump(rs3, base3)
    o3, dd3 = random_rays(rng, rs3.preprocessed()["root_box"], 2048)
    f3, i3 = rs3.intersect(o3, dd3, 1e-3, True)
    d3.update(ray_origins=o3, ray_directions=dd3, hit_floats=f3, hit_ids=i3)
    np.savez_compressed(os.path.join(OUT, "builtin_div5_scene.npz"), **d3)
    rs3.close()
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


def postprocess_golden():
    """(g) src/main.cpp:233-240 applied by the compiled reference to the converged frame of (e), with a
    few special pixels injected (NaN, +-inf, negative, zero, tiny)."""
    R = B.reference()
    assert R is not None, "build oracle/_ref/libref.so first (oracle/build_ref.sh)"
    hdr = np.load(os.path.join(OUT, "builtin_converged_160x120.npz"))["image"].reshape(120, 160, 3).copy()
    hdr[3, 4, 1] = np.nan
    hdr[10, 10, 0] = np.inf
    hdr[11, 12, 2] = -1.0
    hdr[20, 20, 1] = -np.inf
    hdr[5, 5] = 0.0
    hdr[6, 6, 0] = -0.5
    hdr[7, 7] = 1e-4
    hdr[8, 8] = 0.0031308 / (1 - 0.0031308)
    srgb = R.tone_srgb_flags(hdr)
    median = R.median3(srgb)
    rgb8 = R.to_u8(median)
    np.savez_compressed(os.path.join(OUT, "postprocess_160x120.npz"), hdr=hdr, ref_srgb=srgb, ref_median=median,
                        ref_rgb8=rgb8)
    print("postprocess_160x120.npz", os.path.getsize(os.path.join(OUT, "postprocess_160x120.npz")))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "postprocess":
        postprocess_golden()
    else:
        main()
        postprocess_golden()
