"""One rank of the multi-GPU path, started through spp_split.spawn_ranks (the launcher bench.py --gpus N
uses).  Two forms:
  * CPU rehearsal (tests/test_spp_split.py): gloo, the oracle standing in for the GPU kernel;
  * RT_SPLIT_BACKEND=nccl (tests/test_gpu_nccl.py, boxes with at least two GPUs): RCCL, rank r renders
    on device r through the C ABI.
usage: python -m tests.split_worker <spp_total> <out.npy> [fail_rank]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from raytracing_amd import spp_split  # noqa: E402
from tests.conftest import load_scene  # noqa: E402


def main():
    spp_total, out_path = int(sys.argv[1]), sys.argv[2]
    fail_rank = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("RT_SPLIT_BACKEND", "gloo")
    if rank == fail_rank:
        sys.exit(3)  # a rank that dies before the first collective
    arrays, cam, _ = load_scene("quads")
    cam.image_width = 48
    if backend == "nccl":
        from raytracing_amd import binding
        local = int(os.environ.get("LOCAL_RANK", rank))
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        scene = binding.Scene(arrays, device=local)
        render_fn = spp_split.gpu_render_fn(scene, cam, seed=4)
    else:
        from tests import oracle_lib as O
        device = torch.device("cpu")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        osc = O.OracleScene(arrays)

        def render_fn(begin, count, accum):
            osc.render(cam, count, sampler=O.SAMPLER_CB, trace_mode=O.TRACE_REFERENCE, sample_begin=begin, seed=4,
                       accum=accum.numpy())

    # three frames through the pipeline (the reduce of a frame overlaps the next frame's render)
    pipe = spp_split.SplitPipeline(render_fn, (cam.image_height(), cam.image_width, 3), device)
    for _ in range(3):
        pipe.frame(spp_total)
    img = pipe.finish()
    if backend == "nccl":
        torch.cuda.synchronize()
    if rank == 0:
        np.save(out_path, img.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
    if backend == "nccl":
        scene.close()


if __name__ == "__main__":
    main()
