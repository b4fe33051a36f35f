"""rt_multi_*: the sample loop (camera.h:40-47) split over several GPUs inside the library.
A device may be listed more than once, which is how one GPU rehearses the multi-GPU path: the
replicas render their shares of the samples concurrently and the images are gathered and added
on the first one.  The result must be the one-call image up to the grouping of the fp32
additions (every path is keyed by its global sample index)."""
import os
import subprocess

import numpy as np
import pytest

from raytracing_amd import abi
from tests.conftest import load_scene

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT_SCENES = os.path.join(ROOT, "examples", "bin", "rt_scenes")


@pytest.mark.parametrize("scene,n_replicas,spp", [("bouncing_spheres", 2, 9), ("cornell_box", 3, 7), ("quads", 4, 3)])
def test_replicas_add_up_to_the_single_render(scene, n_replicas, spp):
    from raytracing_amd import binding
    arrays, cam, _ = load_scene(scene)
    cam.image_width = 200
    one = binding.Scene(arrays)
    ref, rst = one.render(cam, spp, sample_begin=5, seed=3)
    multi = binding.MultiScene(arrays, [0] * n_replicas)
    acc, st = multi.render(cam, spp, sample_begin=5, seed=3)
    assert st.paths == rst.paths and st.rays == rst.rays  # the same paths, ray for ray
    assert np.allclose(acc, ref, rtol=2 * spp * 2.0 ** -24, atol=1e-6)
    # the shares themselves, rendered one by one and added in the same order: bit-identical
    parts = None
    base, extra = divmod(spp, n_replicas)
    for k in range(n_replicas):
        begin, count = 5 + k * base + min(k, extra), base + (1 if k < extra else 0)
        part, _ = one.render(cam, count, sample_begin=begin, seed=3)
        parts = part if parts is None else parts + part
    assert np.array_equal(acc, parts)
    # a second call ADDS to the caller's image; no statistics wanted
    acc2, _ = multi.render(cam, spp, sample_begin=5, seed=3, accum=acc.copy(), want_stats=False)
    assert np.array_equal(acc2, acc + acc)
    # fewer samples than replicas: some shares are empty
    tiny, tst = multi.render(cam, 1, seed=3)
    ref1, _ = one.render(cam, 1, seed=3)
    assert np.array_equal(tiny, ref1) and tst.paths == cam.image_width * cam.image_height()
    multi.close()
    one.close()


def test_multi_argument_errors():
    from raytracing_amd import binding
    arrays, cam, _ = load_scene("quads")
    with pytest.raises(binding.RtError):
        binding.MultiScene(arrays, [])
    with pytest.raises(binding.RtError) as e:
        binding.MultiScene(arrays, [0, 99])
    assert e.value.code == abi.RT_ERR_ARG
    multi = binding.MultiScene(arrays, [0, 0])
    with pytest.raises(binding.RtError):
        multi.render(cam, -1)
    acc, st = multi.render(cam, 0)
    assert not acc.any()
    multi.close()


def test_dropin_camera_splits_over_the_listed_devices(tmp_path):
    """RTAMD_DEVICES=0,0: the reference-style scene program renders through rt_multi_render and
    writes the PNG the single-device program writes (8-bit pixels: the rounding of the partial
    sums is far below the quantum; at most a pixel value at a rounding boundary may move)."""
    from PIL import Image
    pngs = []
    for devs in (None, "0,0"):
        png = str(tmp_path / f"quads_{devs}.png")
        env = dict(os.environ)
        if devs:
            env["RTAMD_DEVICES"] = devs
        r = subprocess.run([RT_SCENES, "quads", "--spp", "16", "--out", png], check=True, capture_output=True, env=env)
        assert b"Render took" in r.stderr
        pngs.append(np.asarray(Image.open(png).convert("RGB")).astype(np.int32))
    d = np.abs(pngs[0] - pngs[1])
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_bench_under_the_launcher_reduces_over_rccl():
    """bench.py started the way the driver starts it for several GPUs (torch.distributed.run,
    backend nccl = RCCL), here with the one rank this box has: the process group, the
    asynchronous reduce of the accumulation images and the max-over-ranks timing all run; the
    line must name one GPU and a throughput close to the launcher-less run's."""
    import json
    import sys

    from raytracing_amd.spp_split import free_port
    common = ["--steps", "3", "--warmup", "1", "--repeats", "2", "--no-cpu-baseline", "--no-configs"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    launched = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                               "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                               os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common,
                              capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert launched.returncode == 0, launched.stderr[-2000:]
    line = json.loads([ln for ln in launched.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["value"] > 0
    assert "RCCL" in line["config"]["parallelism"] or line["config"]["parallelism"] == "single GPU"
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common,
                           capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert plain.returncode == 0, plain.stderr[-2000:]
    ref = json.loads([ln for ln in plain.stdout.splitlines() if ln.startswith("{")][-1])
    assert ref["config"]["spp_per_gpu_per_step"] == line["config"]["spp_per_gpu_per_step"]
    assert 0.5 < line["value"] / ref["value"] < 2.0
    # two ranks on this one-GPU box: refused, not mislabelled
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common,
                         capture_output=True, text=True, timeout=120, cwd=ROOT)
    import torch
    if torch.cuda.device_count() < 2:
        assert two.returncode == 2 and "refusing" in two.stderr
