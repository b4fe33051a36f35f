#!/usr/bin/env python3
"""GPU box: per-wave timeline of consecutive render launches (RT_AMD_TIMELINE).

Renders `steps` launches of a scene back to back (pipelined unless RT_AMD_PIPELINE=0), lets
the library dump every wave's {start, out of new paths, exit} wall-clock record, and prints
where the time between launches goes: how long the waves of a launch stay after the pool ran
dry, how long a CU waits between the workgroups of consecutive launches, and what share of the
wave slots is occupied over the steady-state part of the run.

usage: python tools/timeline.py [scene] [spp] [steps]
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    scene_name = sys.argv[1] if len(sys.argv) > 1 else "bouncing_spheres"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    out = os.path.join(tempfile.gettempdir(), "rt_timeline.bin")
    os.environ["RT_AMD_TIMELINE"] = out

    import torch
    from raytracing_amd import abi, binding

    torch.cuda.init()
    gold = os.path.join(ROOT, "tests", "golden")
    arrays = abi.SceneArrays.load(os.path.join(gold, scene_name + ".rtscene"))
    cam, _ = abi.load_camera_file(os.path.join(gold, scene_name + ".cam"))
    H, W = cam.image_height(), cam.image_width
    scene = binding.Scene(arrays, device=0)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(steps):
        scene.render_device(cam, spp, accum.data_ptr(), stream=stream, sample_begin=i * spp, seed=0)
    torch.cuda.synchronize()
    scene.close()

    raw = np.fromfile(out, dtype=np.uint64)
    n, waves, bw, total = (int(v) for v in raw[:4])
    rec = raw[4:4 + n * waves * 4].reshape(n, waves, 4)
    start = rec[:, :, 0].astype(np.int64)
    dry = rec[:, :, 1].astype(np.int64)
    exit_ = rec[:, :, 2].astype(np.int64)
    hw = rec[:, :, 3]
    order = np.argsort(start.min(axis=1))
    start, dry, exit_, hw = start[order], dry[order], exit_[order], hw[order]
    t0 = start.min()
    us = lambda t: (t - t0) * 0.01  # noqa: E731  (100 MHz ticks)
    hw_id = (hw & np.uint64(0xffffffff)).astype(np.int64)
    xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    cu_key = xcc * 65536 + (hw_id & 0xff00)  # CU_ID 11:8, SH_ID 12, SE_ID 15:13
    print(f"{scene_name} {W}x{H} {spp} spp: {total} launches, last {n} kept, {waves} waves "
          f"({bw} per workgroup), {len(np.unique(cu_key))} distinct CUs seen")

    print(" launch   first start  last start | dry: first  median    last | exit: first  median    last  (us)")
    for k in range(n):
        d = dry[k][dry[k] > 0]
        print(f"  {k:3d}   {us(start[k].min()):11.0f} {us(start[k].max()):11.0f} | "
              f"{us(d.min()) if d.size else -1:10.0f} {us(np.median(d)) if d.size else -1:7.0f} "
              f"{us(d.max()) if d.size else -1:7.0f} | {us(exit_[k].min()):10.0f} {us(np.median(exit_[k])):7.0f} "
              f"{us(exit_[k].max()):7.0f}")

    lo, hi = n // 4, n - n // 4  # steady-state launches
    if hi - lo >= 2:
        cadence = (np.median(exit_[hi - 1]) - np.median(exit_[lo])) * 0.01 / (hi - 1 - lo)
        print(f"cadence over launches {lo}..{hi - 1}: {cadence:.0f} us per launch")
        # per wave: time in the launch, and how it splits at the moment the pool ran dry
        life = (exit_[lo:hi] - start[lo:hi]) * 0.01
        has = dry[lo:hi] > 0
        fed = np.where(has, dry[lo:hi] - start[lo:hi], exit_[lo:hi] - start[lo:hi]) * 0.01
        drain = life - fed
        print(f"per wave: in the launch {life.mean():.0f} us (fed {fed.mean():.0f} + draining {drain.mean():.0f}); "
              f"draining p50 {np.percentile(drain, 50):.0f}, p90 {np.percentile(drain, 90):.0f}, "
              f"p99 {np.percentile(drain, 99):.0f}, max {drain.max():.0f} us")
        print(f"wave-slot occupancy: {life.mean() / cadence * 100:.1f} % of the cadence "
              f"(fed part {fed.mean() / cadence * 100:.1f} %)")
        # per workgroup: the spread of its waves' exits (the LDS is held until the last one)
        e = exit_[lo:hi].reshape(hi - lo, waves // bw, bw) * 0.01
        s = start[lo:hi].reshape(hi - lo, waves // bw, bw) * 0.01
        spread = e.max(axis=2) - e.mean(axis=2)
        print(f"per workgroup: last wave exits {spread.mean():.0f} us after the average wave "
              f"(p90 {np.percentile(spread, 90):.0f}, max {spread.max():.0f}); waves start within "
              f"{(s.max(axis=2) - s.min(axis=2)).mean():.1f} us")
        # per CU: gap between the end of a workgroup and the start of the next one on it
        gaps = []
        for c in np.unique(cu_key):
            ev = []
            for k in range(lo - 1 if lo else 0, hi):
                m = cu_key[k] == c
                if not m.any():
                    continue
                blocks = np.unique(np.nonzero(m)[0] // bw)
                for b in blocks:
                    sl = slice(b * bw, (b + 1) * bw)
                    ev.append((start[k][sl].min(), exit_[k][sl].max()))
            ev.sort()
            for a, b in zip(ev[:-1], ev[1:]):
                gaps.append((b[0] - a[1]) * 0.01)
        gaps = np.array(gaps)
        if gaps.size:
            print(f"per CU: next workgroup starts {np.median(gaps):.0f} us (median) after the previous one "
                  f"ends; mean {gaps.mean():.0f}, p90 {np.percentile(gaps, 90):.0f}, max {gaps.max():.0f} us; "
                  f"idle share {gaps.clip(min=0).sum() / max(len(np.unique(cu_key)), 1) / ((hi - lo) * cadence) * 100:.1f} %")


if __name__ == "__main__":
    main()
