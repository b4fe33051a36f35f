#!/usr/bin/env python3
"""Times the CPU oracle port with all host cores of the box's share (rows split over threads,
counter-based sampler) next to the 1-thread reference.  usage: python tools/cpu_allcores.py [threads] [spp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import abi
from tests import oracle_lib as O
from tests.conftest import load_scene
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for scene in ("bouncing_spheres", "cornell_box"):
    arrays, cam, _ = load_scene(scene)
    s = spp if scene == "bouncing_spheres" else max(1, spp // 10)
    osc = O.OracleScene(arrays)
    for n in (1, threads):
        t0 = time.perf_counter()
        _, st = osc.render(cam, s, sampler=O.SAMPLER_CB, trace_mode=O.TRACE_REFERENCE, seed=0, n_threads=n)
        dt = time.perf_counter() - t0
        print(f"{scene} {cam.image_width}x{cam.image_height()} {s} spp, oracle port, {n} thread(s): {st.paths/dt/1e6:.2f} Mpaths/s "
              f"{st.rays/dt/1e6:.2f} Mrays/s ({dt:.1f} s)", flush=True)
