#!/usr/bin/env python3
"""Renders the 1 000 001-primitive synthetic scene at 3840x2160 (GPU box): scene build time,
megakernel and wavefront throughput.   usage: python tools/bench_synth.py [spp] [half]"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import abi, binding  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
half = int(sys.argv[2]) if len(sys.argv) > 2 else 500
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = f"/tmp/synth_{half}.rtscene"
t0 = time.perf_counter()
subprocess.run([os.path.join(ROOT, "examples/bin/rt_scenes"), "synth1m", "--half", str(half), "--dump", out],
               check=True, capture_output=True)
t1 = time.perf_counter()
arrays = abi.SceneArrays.load(out)
cam = abi.load_camera_file(out + ".cam")[0]
t2 = time.perf_counter()
sc = binding.Scene(arrays, flags=abi.RT_SCENE_BUILD_DEVICE if "--device-build" in sys.argv else 0)
t3 = time.perf_counter()
info = sc.info()
print(f"prims {len(arrays.prims)}  scene program {t1-t0:.2f} s  load {t2-t1:.2f} s  "
      f"rt_scene_create {t3-t2:.2f} s (flatten + {'device LBVH' if info.built_on_device else 'host SAH'} build "
      f"{info.build_ms/1e3:.2f} s, rest upload)  nodes {info.n_nodes} depth {info.bvh_depth} "
      f"device MB {info.device_bytes/1e6:.1f}", flush=True)
H, W = cam.image_height(), cam.image_width
acc = torch.zeros(H, W, 3, device="cuda")
modes = [(abi.RT_MODE_MEGAKERNEL, "mega")]
if "--wavefront" in sys.argv:
    modes.append((abi.RT_MODE_WAVEFRONT, "wavefront"))
for mode, label in modes:
    best = 1e30
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = sc.render_device(cam, spp, acc.data_ptr(), want_stats=True, mode=mode)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if it:
            best = min(best, dt)
    print(f"{label}: {W}x{H} spp {spp}: {best*1e3:.1f} ms  {st.paths/best/1e6:.1f} Mpaths/s  {st.rays/best/1e9:.2f} Grays/s"
          f"  rays/path {st.rays/st.paths:.2f}", flush=True)
if "--counters" in sys.argv:
    st = sc.render_device(cam, 1, acc.data_ptr(), counters=True)
    print(f"per ray: box tests {st.box_tests/st.rays:.1f} (node visits {st.box_tests/st.rays/2:.1f}), sphere tests "
          f"{st.sphere_tests/st.rays:.2f}, quad tests {st.quad_tests/st.rays:.2f}")
sc.close()
