#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

One step = one pass of the hot path over one batch of synthetic input: every rank renders
`--spp` samples per pixel of the workload scene into its own fp32 accumulation image in HBM
(rt_render_device through the C ABI); with more than one rank the images are combined by one
RCCL reduce (sum, fp32) to rank 0.  Per-GPU work is fixed as N grows (weak scaling): rank r
renders global samples [r*spp, (r+1)*spp).

Prints ONE JSON line (rank 0).  `value` is whole-job Mpaths/s with the scene already resident
in HBM; `roofline` prices the render kernel against the FP32 vector peak; `cpu_baseline` is
the reference renderer (or the oracle port) timed on this box's host cores on a bounded sample.

Consecutive steps overlap on the GPU (the library alternates two internal streams, so the
long-path tail of one launch runs beside the start of the next): inside the timed region a
kernel's HIP-event duration is the span of a launch that shares the GPU with its neighbours,
reported as `kernel_ms_overlapped_avg`.  The roofline's `kernel_ms_avg` is the duration of the
same launch running alone, measured with HIP events over `--iso-steps` further steps after the
timed region (RT_RENDER_NO_OVERLAP); `--serial` runs the timed region itself that way.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
PEAK_FP32_VALU_TFLOPS = 157.3  # 256 CU x 128 lanes x 2 (FMA) x 2.4 GHz

# Nominal fp32 operation weights of the reference algorithm's steps (1 per add / mul / compare /
# min / max, 1 per divide / sqrt / transcendental): the "algorithmic flops" of one launch are
# these weights times the counts the instrumented kernel reports.
W_BOX, W_SPHERE, W_SPHERE_HIT, W_QUAD = 24, 29, 35, 50
W_SHADE = (30, 45, 70, 5)  # lambertian, metal, dielectric, diffuse_light
W_CHECKER, W_NOISE, W_IMAGE, W_RAYGEN, W_RAY = 9, 7 * 160, 8, 30, 3


def algorithmic_flops(st):
    f = W_BOX * st.box_tests + W_SPHERE * st.sphere_tests + W_SPHERE_HIT * st.sphere_hits
    f += W_QUAD * st.quad_tests
    f += sum(w * n for w, n in zip(W_SHADE, st.shades))
    f += W_CHECKER * st.checker_evals + W_NOISE * st.noise_evals + W_IMAGE * st.image_evals
    f += W_RAYGEN * st.paths + W_RAY * st.rays
    return float(f)


def measured_traffic(scene, width, height, spp, key="traffic_bytes_per_launch"):
    """HBM bytes per launch of the render kernel (or, with key="pmc", its VALU counters) from the
    PMC passes committed under profiles/ (hardware counters cannot be read from inside this
    process); None if no profile matches the workload."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not name.endswith("hbm_traffic.json"):
            continue
        try:
            with open(os.path.join(pdir, name)) as f:
                t = json.load(f)
            w = t["workload"]
            if (w["scene"], w["width"], w["height"], w["spp_per_launch"]) == (scene, width, height, spp):
                best = t[key] if key == "pmc" else float(t[key])
        except (OSError, KeyError, ValueError):
            continue
    return best


def cpu_baseline(scene, budget_paths):
    """Times the reference renderer (oracle/_ref/rt_ref, built from the unmodified reference in
    the authoring container) on one host core; falls back to the oracle port if it is absent."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "rt_ref")
    from raytracing_amd import abi
    cam, _ = abi.load_camera_file(os.path.join(GOLD, scene + ".cam"))
    w, h = cam.image_width, cam.image_height()
    spp = max(1, int(round(budget_paths / (w * h))))
    if os.path.exists(ref_bin):
        try:
            env = dict(os.environ, RTW_IMAGES=os.path.join(GOLD, "images"))  # earth's texture file
            out = subprocess.run([ref_bin, "render", scene, str(spp), "0", "0", "0", "-"], check=True,
                                 capture_output=True, cwd="/tmp", timeout=600, env=env)
            m = json.loads(out.stdout.decode().strip().splitlines()[-1])
            return {"value": m["mpaths_per_s"], "unit": "Mpaths/s", "cores": 1, "kind": "reference",
                    "mrays_per_s": m["mrays_per_s"],
                    "sample": f"{scene} {w}x{h} at {spp} spp, depth {cam.max_depth} "
                              f"({m['paths']} paths, {m['seconds']:.1f} s, unmodified reference, 1 thread)"}
        except Exception as e:  # noqa: BLE001
            print(f"[bench] reference binary failed ({e}); timing the oracle port", file=sys.stderr)
    from tests import oracle_lib as O
    arrays = abi.SceneArrays.load(os.path.join(GOLD, scene + ".rtscene"))
    osc = O.OracleScene(arrays)
    _, st = osc.render(cam, spp, sampler=O.SAMPLER_REF, trace_mode=O.TRACE_REFERENCE)
    return {"value": st.paths / st.seconds * 1e-6, "unit": "Mpaths/s", "cores": 1, "kind": "port",
            "mrays_per_s": st.rays / st.seconds * 1e-6,
            "sample": f"{scene} {w}x{h} at {spp} spp, depth {cam.max_depth} "
                      f"({st.paths} paths, {st.seconds:.1f} s, oracle port, 1 thread)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scene", default="bouncing_spheres")
    ap.add_argument("--spp", type=int, default=100, help="samples per pixel per step per GPU")
    ap.add_argument("--width", type=int, default=0, help="override image width (0: the scene's own)")
    ap.add_argument("--serial", action="store_true", help="no overlap between consecutive launches")
    ap.add_argument("--iso-steps", type=int, default=8, help="isolated launches timed for the roofline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-paths", type=float, default=13.5e6, help="paths of the CPU baseline sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from raytracing_amd import abi, binding, spp_split

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torch.distributed.run the collectives run even with one rank (same code path as N > 1)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}: using {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    arrays = abi.SceneArrays.load(os.path.join(GOLD, args.scene + ".rtscene"))
    cam, _ = abi.load_camera_file(os.path.join(GOLD, args.scene + ".cam"))
    if args.width:
        cam.image_width = args.width
    cam.samples_per_pixel = args.spp * world
    H, W = cam.image_height(), cam.image_width
    scene = binding.Scene(arrays, device=local_rank)
    info = scene.info()
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    render_fn = spp_split.gpu_render_fn(scene, cam, seed=0, overlap=not args.serial)
    render_alone = spp_split.gpu_render_fn(scene, cam, seed=0, overlap=False)

    def step(fn=render_fn):
        # rank r renders global samples [r*spp, (r+1)*spp); one RCCL sum-reduce to rank 0
        spp_split.render_split(fn, accum, args.spp * world)

    # one instrumented, untimed pass: rays and test counts of exactly the work a step does
    st = scene.render_device(cam, args.spp, accum.data_ptr(), stream=stream, sample_begin=rank * args.spp,
                             seed=0, counters=True)
    paths_per_step, rays_per_step = int(st.paths), int(st.rays)
    alg_flops = algorithmic_flops(st)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    scene.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, n_launches = scene.profile_end()
    # the same launches one at a time: the render kernel's own duration
    iso_ms, iso_launches = kernel_ms, n_launches
    if not args.serial and args.iso_steps > 0:
        torch.cuda.synchronize()
        scene.profile_begin()
        for _ in range(args.iso_steps):
            step(render_alone)
        torch.cuda.synchronize()
        iso_ms, iso_launches = scene.profile_end()

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([paths_per_step, rays_per_step], dtype=torch.float64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_paths, total_rays = float(cnt[0].item()), float(cnt[1].item())
    else:
        total_paths, total_rays = float(paths_per_step), float(rays_per_step)

    if rank == 0:
        avg_kernel_s = iso_ms / max(iso_launches, 1) * 1e-3
        launches_per_step = max(n_launches, 1) / args.steps
        achieved = alg_flops / launches_per_step / avg_kernel_s / 1e12 if avg_kernel_s > 0 else 0.0
        out = {
            "metric": "Mpaths/s",
            "value": total_paths * args.steps / elapsed * 1e-6,
            "unit": "Mpaths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "mrays_per_s": total_rays * args.steps / elapsed * 1e-6,
            "rays_per_path": total_rays / total_paths,
            "config": {"workload": f"{args.scene} {W}x{H}, {args.spp} spp per GPU per step, "
                                   f"max_depth {cam.max_depth}, {info.n_prims} primitives "
                                   f"({info.n_nodes} BVH nodes, depth {info.bvh_depth}, "
                                   f"{'LDS' if info.lds_resident else 'HBM'}-resident geometry)",
                       "scene": args.scene, "width": W, "height": H, "spp_per_gpu_per_step": args.spp,
                       "max_depth": cam.max_depth, "parallelism": f"spp-split x{world} + RCCL reduce"
                       if world > 1 else "single GPU", "mode": "megakernel"},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                         "traffic": measured_traffic(args.scene, W, H, args.spp), "traffic_unit": "bytes/launch",
                         "pmc": measured_traffic(args.scene, W, H, args.spp, key="pmc"),
                         "hbm_gb_per_s": (measured_traffic(args.scene, W, H, args.spp) or 0.0) / avg_kernel_s * 1e-9
                         if avg_kernel_s > 0 else None,  # against ~8000: the path is not HBM-bound
                         "kernel": "render_mega_bvh" if info.n_nodes else "render_mega", "kernel_ms_avg": avg_kernel_s * 1e3,
                         "kernel_ms_overlapped_avg": kernel_ms / max(n_launches, 1),
                         "launch_interval_ms": elapsed / max(n_launches, 1) * 1e3,
                         "launches": n_launches, "isolated_launches": iso_launches,
                         "overlap": not args.serial, "algorithmic_flops_per_launch": alg_flops / launches_per_step,
                         "counts_per_launch": {"paths": st.paths, "rays": st.rays, "box_tests": st.box_tests,
                                               "sphere_tests": st.sphere_tests, "sphere_hits": st.sphere_hits,
                                               "quad_tests": st.quad_tests, "shades": list(st.shades)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.scene, args.cpu_paths)
        print(json.dumps(out))
    scene.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
