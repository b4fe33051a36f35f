"""Multi-GPU partition of the sample loop: one process per GPU, samples per pixel split
across ranks, per-rank fp32 accumulation images combined by ONE sum-reduce (RCCL over xGMI
when the tensors live on GPUs; any torch.distributed backend works).

Every (pixel, sample) path is independent and the sampler is keyed by (seed, pixel, global
sample index), so rank r renders global samples [begin, begin + count) of every pixel and
the reduced image equals the single-GPU image up to the order of the fp32 additions.
The scene (at most ~100 MB even for the 10^6-primitive case) is replicated per GPU.
"""
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
