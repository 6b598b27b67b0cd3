#!/bin/bash
# The reference's whole main() (src/main.cpp:124-281: built-in scene, 1920x1080, PIXEL_SAMPLE_RATE = 2048, 8 bounces,
# post-processing, output.ppm) on the GPU through the C++ host layer (examples/render_builtin), and its picture compared
# with the picture the reference ships (tests/golden/example_render_480x270.npz).  Run on the GPU box from the repository
# root: bash tools/reference_main_run.sh > profiles/r03_reference_main_run.txt
set -e
mkdir -p gpurun_out/main_run
timeout -k 10 300 examples/render_builtin 1920 1080 2048 1 gpurun_out/main_run/output.ppm
timeout -k 10 300 examples/render_builtin 1920 1080 2048 1 gpurun_out/main_run/output.ppm      # second run: warm process start
python3 - <<'PY'
import numpy as np
raw = open("gpurun_out/main_run/output.ppm", "rb").read()
assert raw[:2] == b"P6"
parts = raw.split(b"\n", 3)
w, h = map(int, parts[1].split())
img = np.frombuffer(parts[3], np.uint8).reshape(h, w, 3).astype(np.float64)
g = np.load("tests/golden/example_render_480x270.npz")
ref = g["image"].astype(np.float64)
ours = img.reshape(h // 4, 4, w // 4, 4, 3).mean(axis=(1, 3))
d = ours - ref
lum = ref.mean(axis=2)
bright, dark = lum >= 80, lum < 10
floor = ~bright & ~dark
print(f"against the reference's example_render.png (4x4 block means, 8-bit scale): mean rgb ours {img.mean(axis=(0,1)).round(2)} "
      f"theirs {g['full_mean_rgb'].round(2)}")
print(f"  cubes and mirror images: mean difference {d[bright].mean():+.2f}, rmse {np.sqrt((d[bright]**2).mean()):.2f}")
print(f"  background: mean difference {d[dark].mean():+.2f}")
print(f"  floor (the reference's picture comes from a contracted build that drops floor hits): mean difference {d[floor].mean():+.2f}")
print(f"  block-for-block correlation {np.corrcoef(ours.ravel(), ref.ravel())[0,1]:.4f}")
PY
rm -f gpurun_out/main_run/output.ppm
