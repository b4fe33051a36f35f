"""Multi-GPU partition of the sample loop: one process per GPU, samples per pixel split
across ranks, per-rank fp32 accumulation images combined by ONE sum-reduce (RCCL over xGMI
when the tensors live on GPUs; any torch.distributed backend works).

Every (pixel, sample) path is independent and the sampler is keyed by (seed, pixel, global
sample index), so rank r renders global samples [begin, begin + count) of every pixel and
the reduced image equals the single-GPU image up to the order of the fp32 additions.
The scene (at most ~200 MB even for the 10^6-primitive case) is replicated per GPU.
"""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def sample_range(spp_total, rank, world):
    """Contiguous, balanced split of [0, spp_total): returns (begin, count) for `rank`."""
    if world < 1 or not 0 <= rank < world or spp_total < 0:
        raise ValueError("bad rank / world / spp")
    base, extra = divmod(spp_total, world)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def render_split(render_fn, accum, spp_total, dst=0, group=None):
    """Renders this rank's share into `accum` (zeroed first) and sum-reduces it to `dst`.

    render_fn(sample_begin, sample_count, accum) must ADD the radiance sums of that sample
    range to accum (rt_render_device does).  After the call rank `dst` holds the sums of all
    spp_total samples; the caller scales by 1/spp_total and tone-maps.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    begin, count = sample_range(spp_total, rank, world)
    accum.zero_()
    if count > 0:
        render_fn(begin, count, accum)
    if world > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return begin, count


class SplitPipeline:
    """render_split for a sequence of frames with the reduce of frame k overlapping the render
    of the frames after it: three accumulation images used in turn, the collective issued
    asynchronously (with the nccl backend it runs on the communicator's own stream, ordered
    after the render that filled its image; the next frames render into the other images
    meanwhile).  An image is reused only after the reduce that read it has been waited for.
    Three images, like the library's three record sets: the render kernels of the next frame
    fill every CU, so the collective's kernel gets its turn only when that frame begins to
    leave, and the frame after it must not have to wait for that.  `finish()` waits for
    everything and returns the image of the last frame (complete on rank `dst`)."""

    N_IMAGES = 3

    def __init__(self, render_fn, shape, device, dst=0, group=None):
        self.render_fn, self.dst, self.group = render_fn, dst, group
        self.images = [torch.zeros(shape, dtype=torch.float32, device=device) for _ in range(self.N_IMAGES)]
        self.pending = [None] * self.N_IMAGES
        self.k = 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def frame(self, spp_total, begin_count=None):
        """One frame of spp_total samples per pixel over all ranks (or this rank's explicit
        (begin, count)).  Returns the image the frame is accumulated in."""
        b = self.k % self.N_IMAGES
        self.k += 1
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        img = self.images[b]
        begin, count = begin_count if begin_count is not None else sample_range(spp_total, self.rank, self.world)
        img.zero_()
        if count > 0:
            self.render_fn(begin, count, img)
        if self.world > 1:
            self.pending[b] = dist.reduce(img, dst=self.dst, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return img

    def finish(self):
        for b in range(self.N_IMAGES):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None
        return self.images[(self.k - 1) % self.N_IMAGES] if self.k else self.images[0]


def gpu_render_fn(scene, cam, seed=0, overlap=True):
    """render_fn for a raytracing_amd.binding.Scene and a CUDA accumulation tensor.
    overlap=False: every launch runs alone on the current stream (isolated kernel timing)."""
    def fn(sample_begin, sample_count, accum):
        if not accum.is_cuda or accum.dtype != torch.float32 or not accum.is_contiguous():
            raise ValueError("accum must be a contiguous fp32 CUDA tensor")
        scene.render_device(cam, sample_count, accum.data_ptr(),
                            stream=torch.cuda.current_stream(accum.device).cuda_stream,
                            sample_begin=sample_begin, seed=seed, overlap=overlap)
    return fn


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(world, argv, env=None, timeout=None):
    """Starts `world` fresh processes running `argv` (one rank each: RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, as torch.distributed.run would),
    from a parent that must not have touched the GPU.  Rank 0 inherits stdout; every rank
    inherits stderr.  Returns the list of exit codes; a rank that dies takes the others down
    (they would wait for it forever in the first collective)."""
    port = free_port()
    procs = []
    for r in range(world):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(argv, env=e, stdout=None if r == 0 else subprocess.DEVNULL))
    codes = [None] * world
    try:
        import time
        t0 = time.monotonic()
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            if any(c not in (None, 0) for c in codes) or (timeout and time.monotonic() - t0 > timeout):
                break
            time.sleep(0.05)
    finally:
        for r, p in enumerate(procs):  # exactly the processes started here
            if p.poll() is None:
                p.terminate()
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            if codes[r] is None:
                codes[r] = p.returncode if p.returncode is not None else -1
    return codes


def relaunch_as_ranks(world, script, args):
    """bench.py --gpus N started without a launcher: N ranks of the same command line."""
    return spawn_ranks(world, [sys.executable, script] + list(args))
