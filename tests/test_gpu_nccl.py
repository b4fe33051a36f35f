"""The multi-GPU path over RCCL with more than one device: needs at least two GPUs, so it skips on
a one-GPU box (tests/test_spp_split.py rehearses the same worker and launcher over gloo on CPU,
tests/test_gpu_multi.py the library's own replicas on one device).

 * two ranks started by spp_split.spawn_ranks (fresh processes from a parent that has not touched
   the GPU, as bench.py --gpus N starts them), SplitPipeline over nccl: the reduced image equals the
   single-GPU image up to the grouping of the fp32 additions;
 * bench.py --gpus 2 --scaling strong reports n_gpus 2."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from raytracing_amd import spp_split
from tests.conftest import ROOT, load_scene

pytestmark = pytest.mark.gpu


def _n_devices():
    import torch
    return torch.cuda.device_count()  # (counting devices does not initialise the GPU)


needs_two = pytest.mark.skipif(_n_devices() < 2, reason="needs at least two GPUs")


def _child_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES=env.get("GPU_MAX_HW_QUEUES", "8"))
    return env


@needs_two
def test_two_ranks_over_rccl_reduce_to_the_single_gpu_image(tmp_path):
    out = str(tmp_path / "acc.npy")
    env = dict(_child_env(), RT_SPLIT_BACKEND="nccl")
    spp = 7
    codes = spp_split.spawn_ranks(2, [sys.executable, "-m", "tests.split_worker", str(spp), out], env=env, timeout=600)
    assert codes == [0, 0]
    got = np.load(out)
    from raytracing_amd import binding
    arrays, cam, _ = load_scene("quads")
    cam.image_width = 48
    sc = binding.Scene(arrays)
    ref, _ = sc.render(cam, spp, seed=4)
    sc.close()
    # the same paths (keyed by pixel and global sample index), summed as two partial sums
    assert got.sum() > 0 and np.allclose(got, ref, rtol=2 * spp * 2.0 ** -24, atol=1e-6)


@needs_two
def test_bench_strong_scaling_on_two_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--scene",
                        "cornell_box", "--spp", "256", "--steps", "1", "--warmup", "1", "--repeats", "1",
                        "--no-cpu-baseline"], capture_output=True, env=_child_env(), timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["spp_per_gpu_per_step"] == 128
