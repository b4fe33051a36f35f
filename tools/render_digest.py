#!/usr/bin/env python3
"""GPU box: renders a golden scene with the library RT_AMD_LIB names (default: the in-tree build) and prints a
digest of the accumulation image and the ray count, so that builds can be compared across processes.
usage: python tools/render_digest.py [scene] [spp]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from raytracing_amd import binding  # noqa: E402
from tests.conftest import load_scene  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "bouncing_spheres"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
arrays, cam, _ = load_scene(scene)
sc = binding.Scene(arrays)
img, st = sc.render(cam, spp, seed=5)
print(scene, spp, "image", hashlib.sha256(img.tobytes()).hexdigest()[:16], "rays", int(st.rays), "lds_resident", sc.info().lds_resident)
sc.close()
