"""The multi-GPU path on CPU: world_size-2 gloo processes, each rendering its share of the
samples (with the oracle standing in for the GPU kernel) and one sum-reduce to rank 0.  The
reduced image must equal the single-process render of all samples."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import spp_split
from tests import oracle_lib as O
from tests.conftest import load_scene


def test_sample_range_partitions():
    for spp in (0, 1, 7, 100, 16384):
        for world in (1, 2, 3, 8):
            ranges = [spp_split.sample_range(spp, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and sum(c for _, c in ranges) == spp
            for (b0, c0), (b1, _) in zip(ranges, ranges[1:]):
                assert b1 == b0 + c0
            assert max(c for _, c in ranges) - min(c for _, c in ranges) <= 1
    with pytest.raises(ValueError):
        spp_split.sample_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, spp_total, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    arrays, cam, _ = load_scene("quads")
    cam.image_width = 48
    osc = O.OracleScene(arrays)

    def render_fn(begin, count, accum):
        a = accum.numpy()
        osc.render(cam, count, sampler=O.SAMPLER_CB, trace_mode=O.TRACE_REFERENCE, sample_begin=begin,
                   seed=4, accum=a)

    accum = torch.zeros((cam.image_height(), cam.image_width, 3), dtype=torch.float32)
    begin, count = spp_split.render_split(render_fn, accum, spp_total)
    assert (begin, count) == spp_split.sample_range(spp_total, rank, world)
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,spp_total", [(2, 8), (2, 5)])
def test_two_ranks_reduce_to_the_single_process_image(tmp_path, world, spp_total):
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, _free_port(), spp_total, out), nprocs=world, join=True)
    got = np.load(out)
    arrays, cam, _ = load_scene("quads")
    cam.image_width = 48
    ref, _ = O.OracleScene(arrays).render(cam, spp_total, sampler=O.SAMPLER_CB, trace_mode=O.TRACE_REFERENCE, seed=4)
    # same samples, summed in two partial sums instead of one running sum: rounding only
    assert np.allclose(got, ref, rtol=1e-6, atol=1e-6)
    assert got.sum() > 0


def test_spawned_ranks_and_pipelined_reduce(tmp_path):
    """The launcher bench.py --gpus N uses (spp_split.spawn_ranks: fresh processes with RANK /
    WORLD_SIZE / MASTER_* set) and the frame pipeline whose reduce overlaps the next render, on
    CPU with gloo: same image as one process; a rank that dies takes the job down with a
    non-zero exit code instead of leaving the others waiting."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "acc.npy")
    env = dict(os.environ, PYTHONPATH=root)
    codes = spp_split.spawn_ranks(2, [sys.executable, "-m", "tests.split_worker", "6", out], env=env, timeout=300)
    assert codes == [0, 0]
    arrays, cam, _ = load_scene("quads")
    cam.image_width = 48
    ref, _ = O.OracleScene(arrays).render(cam, 6, sampler=O.SAMPLER_CB, trace_mode=O.TRACE_REFERENCE, seed=4)
    assert np.allclose(np.load(out), ref, rtol=1e-6, atol=1e-6)
    codes = spp_split.spawn_ranks(2, [sys.executable, "-m", "tests.split_worker", "6", out, "1"], env=env, timeout=300)
    assert codes[1] == 3 and codes[0] != 0


def test_bench_refuses_more_ranks_than_devices():
    """python bench.py --gpus N without a launcher starts N ranks itself, and refuses (exit code 2)
    when fewer than N devices are visible instead of reporting a one-rank number as N."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(max(n, 1) + 1), "--steps", "1"],
                       capture_output=True, env=env, timeout=300)
    assert r.returncode == 2 and b"refusing" in r.stderr and not r.stdout.strip()
    # ... and a rank refuses a --gpus that is not the number of ranks launched
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "1"],
                       capture_output=True, env=env, timeout=300)
    assert r.returncode == 2 and b"refusing" in r.stderr and not r.stdout.strip()


def test_single_process_needs_no_process_group():
    arrays, cam, _ = load_scene("quads")
    cam.image_width = 32
    osc = O.OracleScene(arrays)
    accum = torch.ones((cam.image_height(), cam.image_width, 3), dtype=torch.float32)
    spp_split.render_split(lambda b, c, acc: osc.render(cam, c, sampler=O.SAMPLER_CB, sample_begin=b,
                                                        trace_mode=O.TRACE_REFERENCE, accum=acc.numpy()), accum, 3)
    ref, _ = osc.render(cam, 3, sampler=O.SAMPLER_CB, trace_mode=O.TRACE_REFERENCE)
    assert np.array_equal(accum.numpy(), ref)
