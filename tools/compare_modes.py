#!/usr/bin/env python3
"""Times the megakernel against the wavefront variant on the golden scenes (GPU box).
usage: python tools/compare_modes.py [spp] [scene ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import abi, binding  # noqa: E402
from tests.conftest import SCENES, load_scene  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
scenes = sys.argv[2:] or SCENES
for name in scenes:
    arrays, cam, _ = load_scene(name)
    sc = binding.Scene(arrays)
    H, W = cam.image_height(), cam.image_width
    acc = torch.zeros(H, W, 3, device="cuda")
    line = [f"{name:18s} {W}x{H} spp {spp}"]
    for mode, label in ((abi.RT_MODE_MEGAKERNEL, "mega"), (abi.RT_MODE_WAVEFRONT, "wavefront")):
        best = 1e30
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = sc.render_device(cam, spp, acc.data_ptr(), want_stats=True, mode=mode)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if it:
                best = min(best, dt)
        line.append(f"{label}: {best*1e3:8.2f} ms {st.paths/best/1e6:8.1f} Mpaths/s {st.rays/best/1e9:6.2f} Grays/s"
                    + (f" rounds {st.sched[0]}" if mode else ""))
    print(" | ".join(line), flush=True)
    sc.close()
