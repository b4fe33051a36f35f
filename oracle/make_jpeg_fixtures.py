#!/usr/bin/env python3
"""Writes tests/golden/jpeg/: small JPEG files covering the decoder's cases (sampling
4:4:4 / 4:2:2 / 4:2:0 / 4:4:0, grayscale, restart intervals, odd sizes, 1-pixel-wide chroma) and,
for each, the texels the UNMODIFIED reference produces for it (its rtw_image / stb_image path,
through `oracle/_ref/rt_ref image`).  Runs in the authoring container only.

The JPEGs are encoded by Pillow from a synthetic picture; the expected outputs come from the
reference."""
import os
import subprocess

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "rt_ref")
OUT = os.path.join(ROOT, "tests", "golden", "jpeg")


def picture(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(x / 5.0 + seed), 127 + 120 * np.cos(y / 3.0), (x * 7 + y * 13) % 256], -1)
    img += rng.normal(0, 12, img.shape)          # texture, so that the AC coefficients are busy
    img[h // 3: h // 3 + 4, :, :] = 255          # hard edges: clamping in the IDCT and colour conversion
    img[:, w // 2: w // 2 + 2, :] = 0
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [  # name, width, height, mode, Pillow save options
    ("s444", 67, 45, "RGB", dict(quality=90, subsampling=0)),
    ("s422", 67, 45, "RGB", dict(quality=85, subsampling=1)),
    ("s420", 67, 45, "RGB", dict(quality=75, subsampling=2)),
    ("s420_even", 64, 48, "RGB", dict(quality=95, subsampling=2)),
    ("s420_q20", 50, 33, "RGB", dict(quality=20, subsampling=2)),
    ("gray", 41, 29, "L", dict(quality=80)),
    ("restart", 70, 40, "RGB", dict(quality=85, subsampling=2, restart_marker_blocks=3)),
    ("restart_rows", 40, 70, "RGB", dict(quality=85, subsampling=0, restart_marker_rows=1)),
    ("tiny", 1, 1, "RGB", dict(quality=90, subsampling=2)),
    ("narrow", 2, 37, "RGB", dict(quality=90, subsampling=2)),
    ("optimized", 33, 31, "RGB", dict(quality=88, subsampling=1, optimize=True)),
    # progressive frames (SOF2): Pillow / libjpeg's default script = interleaved DC scan, spectral
    # bands per component, successive-approximation refinements of DC and AC
    ("progressive", 40, 30, "RGB", dict(quality=85, progressive=True)),
    ("prog_444", 67, 45, "RGB", dict(quality=92, subsampling=0, progressive=True)),
    ("prog_422", 35, 50, "RGB", dict(quality=80, subsampling=1, progressive=True)),
    ("prog_420_q30", 83, 61, "RGB", dict(quality=30, subsampling=2, progressive=True)),
    ("prog_gray", 41, 29, "L", dict(quality=80, progressive=True)),
    ("prog_restart", 70, 40, "RGB", dict(quality=85, subsampling=2, progressive=True, restart_marker_blocks=2)),
    ("prog_tiny", 3, 2, "RGB", dict(quality=90, subsampling=2, progressive=True)),
]
SEEDS = {"progressive": 99}

os.makedirs(OUT, exist_ok=True)
for k, (name, w, h, mode, opts) in enumerate(CASES):
    img = Image.fromarray(picture(w, h, SEEDS.get(name, k)), "RGB").convert(mode)
    path = os.path.join(OUT, name + ".jpg")
    img.save(path, "JPEG", **opts)
    r = subprocess.run([REF, "image", path, os.path.join(OUT, name + ".rtimg")], capture_output=True, check=True)
    print(name, os.path.getsize(path), "bytes", r.stdout.decode().strip())
