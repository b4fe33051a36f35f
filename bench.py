#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

One step = one pass of the hot path over one batch of synthetic input: every rank renders its
share of the step's samples per pixel of the workload scene into an fp32 accumulation image in
HBM (rt_render_device through the C ABI); with more than one rank the images are combined by
one RCCL reduce (sum, fp32) to rank 0, issued asynchronously so that it overlaps the next
step's render.  --scaling weak (default): every rank renders --spp samples per step (rank r the
global samples [r*spp, (r+1)*spp)); --scaling strong: --spp is the step's total, split across
the ranks (spp_split.sample_range).

`python bench.py --gpus N` without a launcher starts the N ranks itself (fresh processes, before
anything in the parent touches the GPU) and fails if fewer than N devices are visible; under
torch.distributed.run (RANK / WORLD_SIZE in the environment) it is one of the ranks.

Prints ONE JSON line (rank 0).  `value` is whole-job Mpaths/s with the scene already resident
in HBM, the median of --repeats timed regions of --steps steps each (`ms_per_step` likewise;
`repeat_values` lists them all); `roofline` prices the render kernel against the FP32 vector
peak; `cpu_baseline` is the reference renderer timed on this box's host cores on a bounded
sample; `configs` (one GPU only) holds the same measurements for the other workloads.

Consecutive steps overlap on the GPU (the library alternates two internal streams, so the
long-path tail of one launch runs beside the start of the next): inside the timed region a
kernel's HIP-event duration is the span of a launch that shares the GPU with its neighbours,
reported as `kernel_ms_overlapped_avg`.  The roofline's `kernel_ms_avg` is the duration of the
same launch running alone, measured with HIP events over `--iso-steps` further steps after the
timed region (RT_RENDER_NO_OVERLAP); `--serial` runs the timed region itself that way.
"""
import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")
PEAK_FP32_VALU_TFLOPS = 157.3  # 256 CU x 128 lanes x 2 (FMA) x 2.4 GHz
PEAK_HBM_GBS = 8000.0

# Nominal fp32 operation weights of the reference algorithm's steps (1 per add / mul / compare /
# min / max, 1 per divide / sqrt / transcendental): the "algorithmic flops" of one launch are
# these weights times the counts the instrumented kernel reports.
W_BOX, W_SPHERE, W_SPHERE_HIT, W_QUAD = 24, 29, 35, 50
# A scan tests a quad whose normal, u and v lie along coordinate axes (in its own space) with the
# products by zero components dropped (rt_core.h quad_test_axis): 14 arithmetic operations and 7
# comparisons where quad.h:32-91 spends about 50.  `frac` prices such tests at what is executed;
# `frac_reference_weights` keeps the reference's 50.
W_QUAD_AXIS = 21
PEAK_LANE_OPS = 256 * 128 * 2.4e9  # lane-instructions per second: 256 CUs x 128 lanes x 2.4 GHz
N_SIMD = 256 * 4
W_SHADE = (30, 45, 70, 5)  # lambertian, metal, dielectric, diffuse_light
W_CHECKER, W_NOISE, W_IMAGE, W_RAYGEN, W_RAY = 9, 7 * 160, 8, 30, 3


def axis_quad_fraction(arrays, scanned):
    """Share of the scene's quads that a scan tests with quad_test_axis (0 for scenes walked through a tree:
    their leaves use the general test)."""
    if not scanned:
        return 0.0
    import numpy as np
    from raytracing_amd import abi
    q = arrays.prims[arrays.prims["type"] == abi.RT_PRIM_QUAD]
    if len(q) == 0:
        return 0.0
    u, v = q["b"] != 0, q["c"] != 0
    axis = (u.sum(1) == 1) & (v.sum(1) == 1) & ~(u & v).any(1)
    return float(axis.mean())


def algorithmic_flops(st, axis_quads=0.0):
    f = W_BOX * st.box_tests + W_SPHERE * st.sphere_tests + W_SPHERE_HIT * st.sphere_hits
    f += (W_QUAD_AXIS * axis_quads + W_QUAD * (1.0 - axis_quads)) * st.quad_tests
    f += sum(w * n for w, n in zip(W_SHADE, st.shades))
    f += W_CHECKER * st.checker_evals + W_NOISE * st.noise_evals + W_IMAGE * st.image_evals
    f += W_RAYGEN * st.paths + W_RAY * st.rays
    return float(f)


def measured_traffic(scene, width, height, spp, key="traffic_bytes_per_launch"):
    """HBM bytes per launch of the render kernel (or, with key="pmc", its VALU counters) from the
    PMC passes committed under profiles/ (hardware counters cannot be read from inside this
    process); None if no profile matches the workload.  `spp` is the samples per pixel of ONE
    launch; a profile of the same scene and image taken at another launch size is scaled by the
    number of paths (the traffic is per-path records and per-ray node fetches)."""
    best, best_exact = None, False
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not name.endswith("hbm_traffic.json"):
            continue
        try:
            with open(os.path.join(pdir, name)) as f:
                t = json.load(f)
            w = t["workload"]
            if (w["scene"], w["width"], w["height"]) != (scene, width, height) or spp <= 0:
                continue
            exact = abs(w["spp_per_launch"] - spp) < 1e-9
            if best is not None and best_exact and not exact:
                continue
            best = (dict(t[key], spp_per_launch=w["spp_per_launch"]) if key == "pmc"
                    else float(t[key]) * spp / w["spp_per_launch"])
            best_exact = exact
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    return best


def _sha16(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def cpu_baseline(scene, budget_paths, arrays=None, cam=None, port_too=True, all_cores=False):
    """Times the reference renderer (oracle/_ref/rt_ref: the unmodified reference, compiled by
    __graft_entry__.build() where its sources exist and carried along as a binary) on one host
    core, and the oracle port beside it on a smaller sample; says which binary was timed.
    Scenes the reference does not have (synth1m) are timed with the port only."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "rt_ref")
    from raytracing_amd import abi
    if cam is None:
        cam, _ = abi.load_camera_file(os.path.join(GOLD, scene + ".cam"))
    w, h = cam.image_width, cam.image_height()
    spp = max(1, int(round(budget_paths / (w * h))))
    out = None
    if arrays is None and os.path.exists(ref_bin):
        try:
            env = dict(os.environ, RTW_IMAGES=os.path.join(GOLD, "images"))  # earth's texture file
            res = subprocess.run([ref_bin, "render", scene, str(spp), "0", "0", "0", "-"], check=True,
                                 capture_output=True, cwd="/tmp", timeout=600, env=env)
            m = json.loads(res.stdout.decode().strip().splitlines()[-1])
            out = {"value": m["mpaths_per_s"], "unit": "Mpaths/s", "cores": 1, "kind": "reference",
                   "mrays_per_s": m["mrays_per_s"], "binary": "oracle/_ref/rt_ref", "binary_sha256_16": _sha16(ref_bin),
                   "sample": f"{scene} {w}x{h} at {spp} spp, depth {cam.max_depth} "
                             f"({m['paths']} paths, {m['seconds']:.1f} s, unmodified reference, 1 thread)"}
        except Exception as e:  # noqa: BLE001
            print(f"[bench] reference binary failed ({e}); timing the oracle port", file=sys.stderr)
    if out is not None and not port_too:
        return out
    from tests import oracle_lib as O
    if arrays is None:
        arrays = abi.SceneArrays.load(os.path.join(GOLD, scene + ".rtscene"))
    osc = O.OracleScene(arrays)
    pspp = spp if out is None else max(1, spp // 4)
    _, st = osc.render(cam, pspp, sampler=O.SAMPLER_REF, trace_mode=O.TRACE_REFERENCE)
    port = {"value": st.paths / st.seconds * 1e-6, "unit": "Mpaths/s", "cores": 1, "kind": "port",
            "mrays_per_s": st.rays / st.seconds * 1e-6, "binary": "oracle/liboracle.so",
            "sample": f"{scene} {w}x{h} at {pspp} spp, depth {cam.max_depth} "
                      f"({st.paths} paths, {st.seconds:.1f} s, oracle port, 1 thread)"}
    if all_cores:
        allc = all_cores_leg(osc, cam, max(1, spp // 2), f"{scene} {w}x{h}")
    if out is None:
        out = port
    else:
        out["port"] = {k: port[k] for k in ("value", "mrays_per_s", "sample")}
    if all_cores:
        out["all_cores"] = allc
    return out


def host_threads():
    """Threads for the all-cores leg: the box's CPU share (OMP_NUM_THREADS there), not the whole machine."""
    try:
        n = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        n = 0
    return n if n > 0 else (os.cpu_count() or 1)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def all_cores_leg(osc, cam, spp, label):
    """The port with the image's rows split across threads (counter-based sampler: one stream per path)."""
    from tests import oracle_lib as O
    n = host_threads()
    _, st = osc.render(cam, spp, sampler=O.SAMPLER_CB, trace_mode=O.TRACE_REFERENCE, seed=1, n_threads=n)
    return {"value": st.paths / st.seconds * 1e-6, "unit": "Mpaths/s", "cores": n, "cpu_model": cpu_model(),
            "kind": "port", "mrays_per_s": st.rays / st.seconds * 1e-6,
            "sample": f"{label} at {spp} spp ({st.paths} paths, {st.seconds:.1f} s, oracle port, {n} threads)"}


def cpu_baseline_synth(budget_paths):
    """The 10^6-primitive scene under one bvh_node, 480 x 270, oracle port: one thread and all cores."""
    from raytracing_amd import abi
    from tests import oracle_lib as O
    prog = os.path.join(ROOT, "examples", "bin", "rt_scenes")
    out = "/tmp/bench_synth1m_bvh.rtscene"
    if not (os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(prog)):
        subprocess.run([prog, "synth1m", "--half", "500", "--bvh", "1", "--dump", out], check=True, capture_output=True)
    arrays = abi.SceneArrays.load(out)
    cam = abi.load_camera_file(out + ".cam")[0]
    cam.image_width = 480
    w, h = cam.image_width, cam.image_height()
    spp = max(1, int(round(budget_paths / (w * h))))
    osc = O.OracleScene(arrays)
    _, st = osc.render(cam, spp, sampler=O.SAMPLER_REF, trace_mode=O.TRACE_REFERENCE)
    res = {"value": st.paths / st.seconds * 1e-6, "unit": "Mpaths/s", "cores": 1, "kind": "port",
           "mrays_per_s": st.rays / st.seconds * 1e-6, "binary": "oracle/liboracle.so",
           "sample": f"synth1m (1 000 001 primitives under one bvh_node) {w}x{h} at {spp} spp, depth {cam.max_depth} "
                     f"({st.paths} paths, {st.seconds:.1f} s, oracle port, 1 thread)"}
    res["all_cores"] = all_cores_leg(osc, cam, 4 * spp, f"synth1m {w}x{h}")
    return res


class Workload:
    """A scene resident on this rank's GPU and what it takes to step it."""

    def __init__(self, name, arrays, cam, device_index, serial):
        import torch

        from raytracing_amd import binding, spp_split
        self.torch, self.name, self.cam = torch, name, cam
        self.scene = binding.Scene(arrays, device=device_index)
        self.info = self.scene.info()
        self.axis_quads = axis_quad_fraction(arrays, scanned=self.info.n_nodes == 0)
        self.H, self.W = cam.image_height(), cam.image_width
        self.dev = torch.device("cuda", device_index)
        self.render = spp_split.gpu_render_fn(self.scene, cam, seed=0, overlap=not serial)
        self.render_alone = spp_split.gpu_render_fn(self.scene, cam, seed=0, overlap=False)
        self.pipe = spp_split.SplitPipeline(self.render, (self.H, self.W, 3), self.dev)
        self.pipe_alone = spp_split.SplitPipeline(self.render_alone, (self.H, self.W, 3), self.dev)

    def close(self):
        self.scene.close()

    def measure(self, begin, count, steps, warmup, repeats, iso_steps, serial, use_dist):
        """Times `repeats` regions of `steps` steps (this rank renders samples [begin, begin +
        count) per step).  Returns per-rank results; the caller combines ranks."""
        import torch.distributed as dist
        torch = self.torch
        scratch = torch.zeros((self.H, self.W, 3), dtype=torch.float32, device=self.dev)
        st = self.scene.render_device(self.cam, count, scratch.data_ptr(), stream=torch.cuda.current_stream().cuda_stream,
                                      sample_begin=begin, seed=0, counters=True)
        del scratch

        def step(pipe=self.pipe):
            pipe.frame(0, begin_count=(begin, count))

        for _ in range(warmup):
            step()
        self.pipe.finish()
        elapsed, kernel_ms, n_launches = [], 0.0, 0
        for _ in range(repeats):
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            self.scene.profile_begin()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            self.pipe.finish()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            elapsed.append(time.perf_counter() - t0)
            ms, n = self.scene.profile_end()
            kernel_ms += ms
            n_launches += n
        iso_ms, iso_launches = kernel_ms, n_launches
        if not serial and iso_steps > 0:
            torch.cuda.synchronize()
            self.scene.profile_begin()
            for _ in range(iso_steps):
                step(self.pipe_alone)
            self.pipe_alone.finish()
            torch.cuda.synchronize()
            iso_ms, iso_launches = self.scene.profile_end()
        return {"stats": st, "elapsed": elapsed, "kernel_ms": kernel_ms, "n_launches": n_launches,
                "iso_ms": iso_ms, "iso_launches": iso_launches}

    def kernel_name(self):
        return "render_mega_bvh" if self.info.n_nodes else "render_mega"

    def describe(self, spp_text):
        i = self.info
        tree = (f"{i.n_nodes} nodes of a 4-wide BVH, {i.bvh_depth} levels, "
                f"{'LDS' if i.lds_resident else 'HBM'}-resident geometry") if i.n_nodes else "scanned, no tree"
        return f"{self.name} {self.W}x{self.H}, {spp_text}, max_depth {self.cam.max_depth}, {i.n_prims} primitives ({tree})"


def roofline_of(wl, m, steps_total, spp_per_launch_hint, interval_s=None):
    """interval_s: time between the starts of consecutive launches in the timed region (launches
    overlap there, so it is shorter than a launch on its own)."""
    st = m["stats"]
    alg = algorithmic_flops(st, wl.axis_quads)
    alg_ref = algorithmic_flops(st)
    avg_kernel_s = m["iso_ms"] / max(m["iso_launches"], 1) * 1e-3
    launches_per_step = max(m["n_launches"], 1) / steps_total
    per_launch = alg / launches_per_step
    achieved = per_launch / avg_kernel_s / 1e12 if avg_kernel_s > 0 else 0.0
    spp_launch = spp_per_launch_hint / launches_per_step
    traffic = measured_traffic(wl.name, wl.W, wl.H, spp_launch)
    pmc = measured_traffic(wl.name, wl.W, wl.H, spp_launch, key="pmc")
    r = {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
         # frac = frac_alone: the launch on its own (HIP events, no neighbour on the GPU); frac_by_interval:
         # the same work over the interval at which launches complete while three are in flight
         "frac": achieved / PEAK_FP32_VALU_TFLOPS, "frac_alone": achieved / PEAK_FP32_VALU_TFLOPS,
         "frac_by_interval": (per_launch / interval_s / 1e12 / PEAK_FP32_VALU_TFLOPS) if interval_s else None,
         "frac_reference_weights": (alg_ref / launches_per_step / avg_kernel_s / 1e12 / PEAK_FP32_VALU_TFLOPS)
         if avg_kernel_s > 0 else 0.0,
         "traffic": traffic, "traffic_unit": "bytes/launch", "pmc": pmc,
         "hbm_gb_per_s": (traffic / avg_kernel_s * 1e-9) if traffic and avg_kernel_s > 0 else None,
         "hbm_frac_of_peak": (traffic / avg_kernel_s * 1e-9 / PEAK_HBM_GBS) if traffic and avg_kernel_s > 0 else None,
         "kernel": wl.kernel_name(), "kernel_ms_avg": avg_kernel_s * 1e3,
         "kernel_ms_overlapped_avg": m["kernel_ms"] / max(m["n_launches"], 1),
         "launches": m["n_launches"], "isolated_launches": m["iso_launches"],
         "launches_per_step": launches_per_step,
         "algorithmic_flops_per_launch": per_launch,
         "counts_per_step": {"paths": st.paths, "rays": st.rays, "box_tests": st.box_tests,
                             "sphere_tests": st.sphere_tests, "sphere_hits": st.sphere_hits,
                             "quad_tests": st.quad_tests, "axis_aligned_quad_share": wl.axis_quads,
                             "shades": list(st.shades),
                             "noise_evals": st.noise_evals, "image_evals": st.image_evals}}
    # Executed work, from the hardware counters of the committed profile of this workload (the
    # instruction counts per launch scale with the paths; the time is this run's own):
    #   cycles_per_valu_inst = SIMD cycles available during the launch / VALU wave-instructions per SIMD
    #   executed_frac = active lane-instructions per second / (256 CUs x 128 lanes x 2.4 GHz)
    for k in ("valu_busy_frac", "active_lane_frac", "cycles_per_valu_inst", "executed_frac"):
        r[k] = None
    if pmc and avg_kernel_s > 0:
        scale = spp_launch / pmc.get("spp_per_launch", spp_launch)
        insts = pmc["valu_wave_insts_per_launch"] * scale
        r["valu_busy_frac"] = pmc["valu_busy_frac"]
        r["active_lane_frac"] = pmc["active_lane_frac"]
        r["cycles_per_valu_inst"] = avg_kernel_s * 2.4e9 * N_SIMD / insts
        r["executed_frac"] = insts * 64 * pmc["active_lane_frac"] / avg_kernel_s / PEAK_LANE_OPS
    return r


def intersection_alone(wl, st, n_camera=1 << 24, reps=10):
    """The closest-hit query on its own (rt_closest_hits_device: the render kernel's traversal and the exact
    primitive tests, no shading, no sampling) on a ray batch shaped like this workload's: camera rays of
    random pixels and three generations of diffuse bounce rays leaving their hit points, built on the
    device.  The flops per ray are this workload's own traversal and intersection counts per ray as the
    instrumented render reported them (the batch is not a render's exact ray set)."""
    torch, sc, cam = wl.torch, wl.scene, wl.cam
    from raytracing_amd import binding
    f = binding.camera_initialize(cam)
    dev = wl.dev
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    stream = torch.cuda.current_stream().cuda_stream
    v3 = lambda a: torch.tensor([a[0], a[1], a[2]], dtype=torch.float32, device=dev)
    px = torch.rand(n_camera, generator=g, device=dev) * f.image_width - 0.5
    py = torch.rand(n_camera, generator=g, device=dev) * f.image_height - 0.5
    rays = torch.empty((n_camera, 7), dtype=torch.float32, device=dev)  # rt_ray: origin, direction, time
    rays[:, 0:3] = v3(f.center)
    rays[:, 3:6] = v3(f.pixel00_loc) + px[:, None] * v3(f.pixel_delta_u) + py[:, None] * v3(f.pixel_delta_v) - v3(f.center)
    rays[:, 6] = torch.rand(n_camera, generator=g, device=dev)
    gens = [rays]
    for _ in range(3):
        cur = gens[-1]
        hits = torch.empty((len(cur), 11), dtype=torch.float32, device=dev)  # rt_hit: prim, face, t, p, normal, u, v
        sc.trace_device(cur.data_ptr(), len(cur), hits.data_ptr(), stream, sched=True)
        torch.cuda.synchronize()
        m = hits.view(torch.int32)[:, 0] >= 0
        n = int(m.sum().item())
        if n == 0:
            break
        z = torch.rand(n, generator=g, device=dev) * 2 - 1
        a = torch.rand(n, generator=g, device=dev) * 6.2831853
        r = torch.sqrt(torch.clamp(1 - z * z, min=0))
        nxt = torch.empty((n, 7), dtype=torch.float32, device=dev)
        nxt[:, 0:3] = hits[m, 3:6]
        nxt[:, 3:6] = hits[m, 6:9] + torch.stack([r * torch.cos(a), r * torch.sin(a), z], 1)
        nxt[:, 6] = cur[m, 6]
        gens.append(nxt)
        del hits
    batch = torch.cat(gens).contiguous()
    n = len(batch)
    prim = torch.empty(n, dtype=torch.int32, device=dev)
    t = torch.empty(n, dtype=torch.float32, device=dev)
    for _ in range(2):
        sc.closest_hits_device(batch.data_ptr(), n, prim.data_ptr(), t.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sc.closest_hits_device(batch.data_ptr(), n, prim.data_ptr(), t.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    seconds = e0.elapsed_time(e1) * 1e-3 / reps
    flops_per_ray = (W_BOX * st.box_tests + W_SPHERE * st.sphere_tests + W_SPHERE_HIT * st.sphere_hits
                     + (W_QUAD_AXIS * wl.axis_quads + W_QUAD * (1.0 - wl.axis_quads)) * st.quad_tests
                     + W_RAY * st.rays) / max(st.rays, 1)
    achieved = n * flops_per_ray / seconds / 1e12
    return {"entry": "rt_closest_hits_device", "rays": n, "rays_by_generation": [len(x) for x in gens],
            "hit_fraction": float((prim >= 0).float().mean().item()), "ms_per_batch": seconds * 1e3,
            "mrays_per_s": n / seconds * 1e-6, "flops_per_ray": flops_per_ray, "achieved": achieved,
            "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS,
            "bytes_per_ray": 28 + 8, "hbm_gb_per_s": n * 36 / seconds * 1e-9,
            "note": "traversal and exact primitive tests only; flops per ray = this workload's box / sphere / quad test "
                    "counts per ray from the instrumented render; rays: camera rays of random pixels and three "
                    "generations of diffuse bounce rays from their hit points"}


def load_workload(name, width=0):
    """(arrays, cam) of a golden scene, or of synth1m from the scene program (None if it is absent)."""
    from raytracing_amd import abi
    if name == "synth1m":
        prog = os.path.join(ROOT, "examples", "bin", "rt_scenes")
        if not os.path.exists(prog):
            return None
        out = "/tmp/bench_synth1m.rtscene"
        fresh = (os.path.exists(out) and os.path.exists(out + ".cam")
                 and os.path.getmtime(out) >= os.path.getmtime(prog))
        if not fresh:  # (a profiler run generates it beforehand, see tools/pmc_scene.sh)
            subprocess.run([prog, "synth1m", "--half", "500", "--dump", out], check=True, capture_output=True)
        arrays = abi.SceneArrays.load(out)
        cam = abi.load_camera_file(out + ".cam")[0]
    else:
        arrays = abi.SceneArrays.load(os.path.join(GOLD, name + ".rtscene"))
        cam, _ = abi.load_camera_file(os.path.join(GOLD, name + ".cam"))
    if width:
        cam.image_width = width
    return arrays, cam


# the reference's other benchmark configurations, as one GPU runs them: (scene, spp per step, steps, CPU sample in paths)
OTHER_CONFIGS = [
    ("config2", "checkered_spheres", 100, 10, 4.5e6), ("config2", "perlin_spheres", 100, 10, 1.8e6),
    ("config3", "earth", 100, 10, 9e6), ("config3", "quads", 100, 10, 6.4e6), ("config3", "simple_light", 100, 10, 3.6e6),
    ("config4", "cornell_box", 2048, 1, 2.88e6),   # one GPU's share of 16 384 spp over 8
    ("config5", "synth1m", 8, 2, 1.3e6),           # 3840x2160, 8 of the 1024 spp
]


def run_other_configs(args, dev_index):
    out = {}
    for cfg, name, spp, steps, cpu_paths in OTHER_CONFIGS:
        loaded = load_workload(name)
        if loaded is None:
            out[name] = {"config": cfg, "skipped": "examples/bin/rt_scenes (the scene program) is not built"}
            continue
        arrays, cam = loaded
        cam.samples_per_pixel = spp
        wl = Workload(name, arrays, cam, dev_index, serial=False)
        m = wl.measure(0, spp, steps, 1, 3, 2, False, False)
        el = statistics.median(m["elapsed"])
        st = m["stats"]
        r = roofline_of(wl, m, steps * 3, spp, interval_s=el / steps / max(m["n_launches"] / (steps * 3), 1))
        entry = {"config": cfg, "workload": wl.describe(f"{spp} spp per step"), "steps": steps, "repeats": 3,
                 "value": st.paths * steps / el * 1e-6, "unit": "Mpaths/s", "mrays_per_s": st.rays * steps / el * 1e-6,
                 "rays_per_path": st.rays / max(st.paths, 1), "ms_per_step": el / steps * 1e3,
                 "kernel": r["kernel"], "kernel_ms_alone": r["kernel_ms_avg"], "launches_per_step": r["launches_per_step"],
                 "roofline": {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_alone", "frac_by_interval",
                                                "frac_reference_weights", "valu_busy_frac", "active_lane_frac",
                                                "cycles_per_valu_inst", "executed_frac", "traffic", "hbm_gb_per_s",
                                                "hbm_frac_of_peak", "algorithmic_flops_per_launch")}}
        wl.close()
        print(f"[bench] {name}: {entry['value']:.0f} Mpaths/s, kernel alone {entry['kernel_ms_alone']:.3f} ms, "
              f"frac {entry['roofline']['frac']:.4f}", file=sys.stderr, flush=True)
        if not args.no_cpu_baseline:
            if name == "synth1m":
                # the reference's main.cpp has no such scene; the port is timed on the same million
                # primitives held under one bvh_node (bvh.h:48-88), which is how a user of the reference
                # would hold them, at 480 x 270
                entry["cpu_baseline"] = cpu_baseline_synth(cpu_paths)
            else:
                entry["cpu_baseline"] = cpu_baseline(name, cpu_paths, port_too=False)
            print(f"[bench] {name}: CPU {entry['cpu_baseline']['kind']} {entry['cpu_baseline']['value']:.3f} Mpaths/s", file=sys.stderr, flush=True)
        out[name] = entry
    return out


def run_rank(args):
    import torch
    import torch.distributed as dist

    from raytracing_amd import spp_split

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"[bench] --gpus {args.gpus} but {world} rank(s) were launched: refusing to report a mislabelled number",
              file=sys.stderr)
        return 2
    # under torch.distributed.run the collectives run even with one rank (same code path as N > 1)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if local_rank >= torch.cuda.device_count():
        print(f"[bench] rank {rank}: device {local_rank} does not exist ({torch.cuda.device_count()} visible)", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    strong = args.scaling == "strong"
    scene_name = args.scene or ("cornell_box" if strong else "bouncing_spheres")
    spp = args.spp or (16384 if strong else 100)
    steps = args.steps if args.steps is not None else (2 if strong else 200)
    warmup = args.warmup if args.warmup is not None else (1 if strong else 10)
    arrays, cam = load_workload(scene_name, args.width)
    if strong:
        begin, count = spp_split.sample_range(spp, rank, world)
        cam.samples_per_pixel = spp
    else:
        begin, count = rank * spp, spp
        cam.samples_per_pixel = spp * world
    wl = Workload(scene_name, arrays, cam, local_rank, args.serial)
    m = wl.measure(begin, count, steps, warmup, args.repeats, args.iso_steps, args.serial, use_dist)
    st = m["stats"]

    elapsed = m["elapsed"]
    if use_dist:
        t = torch.tensor(elapsed, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # every region: the slowest rank's time
        elapsed = [float(x) for x in t.tolist()]
        cnt = torch.tensor([st.paths, st.rays, 1], dtype=torch.float64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_paths, total_rays, joined = float(cnt[0].item()), float(cnt[1].item()), int(round(float(cnt[2].item())))
    else:
        total_paths, total_rays, joined = float(st.paths), float(st.rays), 1
    if joined != args.gpus:
        print(f"[bench] {joined} rank(s) joined the collective, --gpus {args.gpus}", file=sys.stderr)
        return 2

    if rank == 0:
        values = [total_paths * steps / e * 1e-6 for e in elapsed]
        med = statistics.median(elapsed)
        spp_text = (f"{spp} spp per step split over {world} GPU(s)" if strong else f"{spp} spp per GPU per step")
        out = {
            "metric": "Mpaths/s",
            "value": total_paths * steps / med * 1e-6,
            "unit": "Mpaths/s",
            "n_gpus": joined,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": med / steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "repeats": args.repeats,
            "repeat_values": values,
            "repeat_spread": (max(values) - min(values)) / statistics.median(values) if values else 0.0,
            "mrays_per_s": total_rays * steps / med * 1e-6,
            "rays_per_path": total_rays / total_paths,
            "config": {"workload": wl.describe(spp_text), "scene": scene_name, "width": wl.W, "height": wl.H,
                       "spp_per_gpu_per_step": count, "max_depth": cam.max_depth,
                       "parallelism": f"spp-split x{world} + RCCL reduce (asynchronous, overlapping the next step)"
                       if world > 1 else "single GPU", "mode": "megakernel"},
            "roofline": dict(roofline_of(wl, m, steps * args.repeats, count,
                                         interval_s=med / steps / max(m["n_launches"] / (steps * args.repeats), 1)),
                             overlap=not args.serial,
                             launch_interval_ms=med / steps * 1e3 / max(m["n_launches"] / (steps * args.repeats), 1)),
        }
        print(f"[bench] {scene_name}: {out['value']:.0f} Mpaths/s over {joined} GPU(s)", file=sys.stderr)
        if world == 1 and not args.no_configs and not args.serial:
            try:  # (an addition to the line: a failure here must not cost the line itself)
                out["intersection_alone"] = intersection_alone(wl, st)
                ia = out["intersection_alone"]
                print(f"[bench] {scene_name}: closest hits alone {ia['mrays_per_s']:.0f} Mrays/s, frac {ia['frac']:.4f}",
                      file=sys.stderr)
            except Exception as e:  # noqa: BLE001
                out["intersection_alone"] = None
                print(f"[bench] intersection_alone failed: {e}", file=sys.stderr)
    wl.close()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = (cpu_baseline_synth(1.3e6) if scene_name == "synth1m"
                                   else cpu_baseline(scene_name, args.cpu_paths, all_cores=True))
        if world == 1 and not args.no_configs and not strong and not args.scene:
            out["configs"] = run_other_configs(args, local_rank)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="steps per timed region (default 200; strong scaling 2)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions; value is their median")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--scene", default=None, help="default: bouncing_spheres (weak), cornell_box (strong)")
    ap.add_argument("--spp", type=int, default=0, help="weak: samples per pixel per step per GPU (100); "
                                                       "strong: per step over all GPUs (16384)")
    ap.add_argument("--width", type=int, default=0, help="override image width (0: the scene's own)")
    ap.add_argument("--serial", action="store_true", help="no overlap between consecutive launches")
    ap.add_argument("--iso-steps", type=int, default=8, help="isolated launches timed for the roofline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other workloads (configs key)")
    ap.add_argument("--cpu-paths", type=float, default=13.5e6, help="paths of the CPU baseline sample")
    args = ap.parse_args()

    # Four hardware queues per process (HIP's default) are one too few once RCCL's streams join the
    # library's four: two render streams then share a queue and their launches cannot overlap
    # (7.4 instead of 8.3 Gpaths/s per GPU).  Must be set before anything initialises HIP; the
    # ranks started below inherit it.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the ranks from here.  Nothing in this process has touched the GPU
        # (counting devices does not initialise it).
        import torch

        from raytracing_amd import spp_split
        n = torch.cuda.device_count()
        if n < args.gpus:
            print(f"[bench] --gpus {args.gpus} but only {n} GPU(s) are visible: refusing to run fewer ranks",
                  file=sys.stderr)
            sys.exit(2)
        codes = spp_split.relaunch_as_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
        if any(codes):
            print(f"[bench] rank exit codes {codes}", file=sys.stderr)
            sys.exit(1)
        sys.exit(0)
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
