#!/usr/bin/env python3
"""GPU box: throughput of the batch entry points on a ray batch shaped like a render's (camera rays of
random pixels and three generations of diffuse bounce rays leaving their hit points):
rt_trace_batch_sched_device and rt_trace_batch_device, which return full rt_hit records (t, point,
normal, u, v, face, primitive: 28 bytes in and 44 bytes out per ray), and rt_closest_hits_device, which
returns the primitive and t only: intersection and traversal alone.

usage: python tools/trace_bench.py [scene] [camera rays, default 2^21]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.init()
from raytracing_amd import abi, binding  # noqa: E402
from tests.conftest import load_scene  # noqa: E402


def unit_vectors(rng, n):
    z = rng.uniform(-1, 1, n).astype(np.float32)
    a = rng.uniform(0, 2 * np.pi, n).astype(np.float32)
    r = np.sqrt(np.maximum(0, 1 - z * z))
    return np.stack([r * np.cos(a), r * np.sin(a), z], 1).astype(np.float32)


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "bouncing_spheres"
    n0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
    arrays, cam, _ = load_scene(scene)
    sc = binding.Scene(arrays)
    f = binding.camera_initialize(cam)
    rng = np.random.default_rng(1)
    px = rng.uniform(-0.5, f.image_width - 0.5, n0).astype(np.float32)
    py = rng.uniform(-0.5, f.image_height - 0.5, n0).astype(np.float32)
    tgt = (np.array(f.pixel00_loc, np.float32) + px[:, None] * np.array(f.pixel_delta_u, np.float32)
           + py[:, None] * np.array(f.pixel_delta_v, np.float32))
    rays = np.zeros(n0, abi.RAY_DT)
    rays["orig"] = np.array(f.center, np.float32)
    rays["dir"] = tgt - rays["orig"]
    rays["tm"] = rng.uniform(0, 1, n0)
    gens = [rays]
    for g in range(3):
        hits = sc.trace(gens[-1], sched=True)
        m = hits["prim"] >= 0
        nxt = np.zeros(int(m.sum()), abi.RAY_DT)
        nxt["orig"] = hits["p"][m]
        nxt["dir"] = hits["normal"][m] + unit_vectors(rng, len(nxt))
        nxt["tm"] = gens[-1]["tm"][m]
        gens.append(nxt)
    batch = np.concatenate(gens)
    n = len(batch)
    d_rays = torch.from_numpy(batch.view(np.uint8).reshape(n, -1).copy()).cuda()
    d_hits = torch.zeros((n, abi.HIT_DT.itemsize), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    print(f"{scene}: {n} rays ({', '.join(str(len(g)) for g in gens)} by generation)")
    for sched, name in ((True, "trace_sched (the render kernel's traversal)"), (False, "trace_batch (whole ray per lane)")):
        for _ in range(3):
            sc.trace_device(d_rays.data_ptr(), n, d_hits.data_ptr(), stream, sched=sched)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            sc.trace_device(d_rays.data_ptr(), n, d_hits.data_ptr(), stream, sched=sched)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        hb = (abi.RAY_DT.itemsize + abi.HIT_DT.itemsize) * n / (ms * 1e-3) / 1e9
        print(f"  {name}: {ms:.3f} ms per batch, {n / ms * 1e-6:.2f} Grays/s, {hb:.0f} GB/s of ray and hit records")
    # closest primitive and parameter only (rt_closest_hits_device): intersection and traversal alone
    d_prim = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_t = torch.zeros(n, dtype=torch.float32, device="cuda")
    for _ in range(3):
        sc.closest_hits_device(d_rays.data_ptr(), n, d_prim.data_ptr(), d_t.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        sc.closest_hits_device(d_rays.data_ptr(), n, d_prim.data_ptr(), d_t.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    hit = (d_prim >= 0).float().mean().item()
    print(f"  rt_closest_hits_device: {ms:.3f} ms per batch, {n / ms * 1e-6:.2f} Grays/s ({100 * hit:.1f} % of the rays hit)")
    # the same answers as the full records
    ref = np.frombuffer(d_hits.cpu().numpy().tobytes(), abi.HIT_DT)
    assert np.array_equal(d_prim.cpu().numpy(), ref["prim"]) and np.array_equal(d_t.cpu().numpy()[ref["prim"] >= 0], ref["t"][ref["prim"] >= 0])
    sc.close()


if __name__ == "__main__":
    main()
