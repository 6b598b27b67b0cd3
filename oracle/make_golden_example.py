"""TEST INFRASTRUCTURE ONLY: writes tests/golden/example_render_480x270.npz from the one rendered image the reference
ships, /root/reference/example_render.png (1920x1080, 8 bits per channel: the built-in scene of src/main.cpp:132-187 at
2048 samples per pixel after tone mapping, sRGB transfer and the 3x3 median, written by the author's MSVC /fp:fast build
with seeds from std::random_device; README.txt shows it as the program's output).  Run in the development container only:
    python oracle/make_golden_example.py

The file holds data, no code: the image averaged over 4x4 pixel blocks (480x270x3, float16 on the 8-bit scale) -- enough
for what the image can pin, a loose whole-pipeline comparison (scene, camera, materials, estimator, post-processing), since
neither its sampler nor its floating-point mode can be reproduced bit for bit.
"""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/example_render.png"
OUT = os.path.join(ROOT, "tests", "golden", "example_render_480x270.npz")


def main():
    a = np.asarray(Image.open(SRC).convert("RGB"), np.float64)
    assert a.shape == (1080, 1920, 3)
    small = a.reshape(270, 4, 480, 4, 3).mean(axis=(1, 3))
    np.savez_compressed(OUT, image=small.astype(np.float16), full_width=np.array([1920]), full_height=np.array([1080]),
                        block=np.array([4]), full_mean_rgb=a.mean(axis=(0, 1)))
    print("wrote", OUT, os.path.getsize(OUT), "bytes; mean rgb", a.mean(axis=(0, 1)))


if __name__ == "__main__":
    main()
