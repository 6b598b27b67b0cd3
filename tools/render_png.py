#!/usr/bin/env python3
"""Render a workload on the GPU and write a tone-mapped PNG (Reinhard + sRGB as src/main.cpp:233-235)."""
import argparse, os, struct, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_png(path, rgb8):
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))
    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cornell")
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--height", type=int, default=270)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--tris", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/render.png")
    a = ap.parse_args()
    from raytracing_amd import scenes
    from raytracing_amd.render import DeviceScene, RayTraceData
    import time
    t0 = time.time()
    if a.workload == "cornell":
        sc = scenes.cornell_scene()
    elif a.workload == "builtin":
        sc = scenes.builtin_scene()
    else:
        sc = scenes.dragon_scale_scene(a.tris or 900_000)
    t1 = time.time(); sc.preprocess(); t2 = time.time()
    print(f"scene gen {t1-t0:.2f}s preprocess+bsp {t2-t1:.2f}s tris {len(sc.preprocessed_tris.tris)} "
          f"inodes {sc.bsp_tree.inode_count} leaves {sc.bsp_tree.leaf_count} refs {sc.bsp_tree.ref_count} "
          f"depth {sc.bsp_tree.depth} oversized {sc.bsp_tree.oversized_leaves}", flush=True)
    ds = DeviceScene(RayTraceData(sc))
    print(f"device bytes {ds.device_bytes/1e6:.1f} MB", flush=True)
    p = ds.make_params(a.width, a.height, sc.camera.position, sc.pixel_to_ray_transform(a.width, a.height), spp=a.spp, flags=2)
    img, st = ds.render(p)
    print(st, f"Mrays/s {st['segments']/st['ms_total']/1e3:.1f}", flush=True)
    ldr = img / (1.0 + img)
    srgb = np.where(ldr <= 0.0031308, ldr * 12.92, 1.055 * np.power(np.maximum(ldr, 1e-12), 1 / 2.4) - 0.055)
    write_png(a.out, np.clip(255.0 * srgb, 0, 255).astype(np.uint8))
    ds.close()


if __name__ == "__main__":
    main()
