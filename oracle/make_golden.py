#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/ by running the UNMODIFIED reference
(oracle/_ref/rt_ref, built by oracle/Makefile from /root/reference) in the
authoring container.  The GPU box never runs this: it only reads the fixtures.

  python oracle/make_golden.py scenes   # <scene>.rtscene + <scene>.cam
  python oracle/make_golden.py rays     # <scene>.rays   (ray batch + closest-hit records)
  python oracle/make_golden.py pins     # <scene>.pin.npz (fp32 radiance sums of a row band, low spp)
  python oracle/make_golden.py images [scene ...]   # <scene>.ref.png + <scene>.ref.json (high spp)

Fixtures are data: inputs and the reference's outputs.  No reference source is stored.
"""
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "rt_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
ENV = dict(os.environ, RTW_IMAGES="/root/reference/images")

SCENES = ["bouncing_spheres", "checkered_spheres", "earth", "perlin_spheres", "quads",
          "simple_light", "cornell_box"]
# spp of the converged reference images (the PSNR >= 40 dB gate needs about this much)
IMAGE_SPP = {"bouncing_spheres": 1024, "checkered_spheres": 1024, "earth": 1024,
             "perlin_spheres": 1024, "quads": 1024, "simple_light": 8192, "cornell_box": 16384,
             "synth60_bvh": 1024}
# scenes of the harness itself (not in main.cpp): scene dump, camera and ray records only
# (synth60_*: the synthetic grid scene of the 4K configuration at 14 401 primitives, as a plain list and
# under one bvh_node; `images synth60_bvh` also renders the latter's converged image)
EXTRA_SCENES = ["rotated_in_bvh", "synth60_list", "synth60_bvh"]
RAY_PATHS = 1500
PIN_SPP = 2
PIN_ROWS = 24
BAND_BURN = 50_000_000_000  # draws skipped per extra band: bands use disjoint stretches of the stream


def run(args):
    out = subprocess.run([REF] + [str(a) for a in args], check=True, capture_output=True, env=ENV,
                         cwd=tempfile.gettempdir())
    return json.loads(out.stdout.decode().strip().splitlines()[-1])


def stage_scenes():
    for s in SCENES + EXTRA_SCENES:
        print(run(["dump", s, f"{GOLD}/{s}.rtscene", f"{GOLD}/{s}.cam"]))


def stage_rays():
    for s in SCENES + EXTRA_SCENES:
        print(run(["rays", s, RAY_PATHS, f"{GOLD}/{s}.rays"]))


def stage_pins():
    for s in SCENES:
        info = run(["dump", s, os.devnull, os.path.join(tempfile.gettempdir(), "x.cam")])
        h, w = info["height"], info["width"]
        r0 = h // 2 - PIN_ROWS // 2
        with tempfile.TemporaryDirectory() as td:
            meta = run(["render", s, PIN_SPP, 0, r0, r0 + PIN_ROWS, f"{td}/a.f32", f"{td}/a.u8"])
            sums = np.fromfile(f"{td}/a.f32", np.float32).reshape(h, w, 3)[r0:r0 + PIN_ROWS]
            u8 = np.fromfile(f"{td}/a.u8", np.uint8).reshape(h, w, 3)[r0:r0 + PIN_ROWS]
        np.savez_compressed(f"{GOLD}/{s}.pin.npz", sums=sums, u8=u8, row0=r0, spp=PIN_SPP,
                            stream_pos=meta["stream_pos"], rays=meta["rays"])
        print(s, meta)


# row bands per scene and the stretch of the random stream skipped per band index (bands use
# disjoint stretches); finished bands are kept in BAND_DIR so that an interrupted run resumes
IMAGE_BANDS = {"cornell_box": (30, 8_000_000_000), "simple_light": (24, 4_000_000_000)}
BAND_DIR = os.path.join(tempfile.gettempdir(), "golden_bands")


def stage_images(scenes, workers=6):
    from PIL import Image
    for s in scenes:
        spp = IMAGE_SPP[s]
        info = run(["dump", s, os.devnull, os.path.join(tempfile.gettempdir(), "x.cam")])
        h, w = info["height"], info["width"]
        # split into row bands so that several processes share the work
        n_bands, burn = IMAGE_BANDS.get(s, (workers if h * w * spp > 2e8 else 1, BAND_BURN))
        edges = [round(k * h / n_bands) for k in range(n_bands + 1)]
        td = os.path.join(BAND_DIR, f"{s}_{spp}_{n_bands}")
        os.makedirs(td, exist_ok=True)

        def job(k):
            meta_path = f"{td}/b{k}.json"
            if os.path.exists(meta_path):
                return json.load(open(meta_path))
            m = run(["render", s, spp, k * burn, edges[k], edges[k + 1], f"{td}/b{k}.f32"])
            with open(meta_path + ".tmp", "w") as f:
                json.dump(m, f)
            os.replace(meta_path + ".tmp", meta_path)
            print(s, "band", k, "done", flush=True)
            return m
        with ThreadPoolExecutor(workers) as ex:
            metas = list(ex.map(job, range(n_bands)))
        sums = np.zeros((h, w, 3), np.float32)
        for k in range(n_bands):
            sums += np.fromfile(f"{td}/b{k}.f32", np.float32).reshape(h, w, 3)
        # write_color (color.h:17-32) exactly as the reference quantises: via the harness' own
        # u8 output for a single band, or restated here in fp32 for the merged image
        c = sums * np.float32(1.0 / spp)
        g = np.where(c > 0, np.sqrt(c, dtype=np.float32), np.float32(0))
        g = np.clip(g, np.float32(0.0), np.float32(0.999))
        u8 = (np.float32(256) * g).astype(np.int32).astype(np.uint8)
        Image.fromarray(u8, "RGB").save(f"{GOLD}/{s}.ref.png", optimize=True)
        meta = {"scene": s, "spp": spp, "width": w, "height": h, "bands": n_bands,
                "paths": sum(m["paths"] for m in metas), "rays": sum(m["rays"] for m in metas),
                "cpu_seconds": sum(m["seconds"] for m in metas),
                "mean_linear_rgb": [float(x) for x in c.reshape(-1, 3).mean(0)],
                "generator": "oracle/_ref/rt_ref render (unmodified reference, g++ 11.4 -O2, 1 thread per band)"}
        with open(f"{GOLD}/{s}.ref.json", "w") as f:
            json.dump(meta, f, indent=1)
        print(meta, flush=True)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    stage = sys.argv[1]
    if stage == "scenes":
        stage_scenes()
    elif stage == "rays":
        stage_rays()
    elif stage == "pins":
        stage_pins()
    elif stage == "images":
        stage_images(sys.argv[2:] or SCENES)
    else:
        sys.exit(__doc__)
