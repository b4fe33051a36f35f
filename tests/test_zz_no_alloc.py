"""rt_render_device_static: the render entry point that allocates nothing (its launch scratch is
the library's statically allocated device block) against rt_render_device, through the C ABI.
No counterpart in the reference; the check is against this library's own entry point, which the
other GPU tests pin to the oracle.  Bit-exact where both calls use one launch (integer ray
counts aside, the image is a sum of fp32 records in sample order); within the rounding of the
additions, 2 (spp - 1) 2^-24 relative, where the static block forces more launches than
rt_render_device needs (the per-launch partial sums are then added in a different grouping)."""
import numpy as np
import pytest

from tests.conftest import load_scene

pytestmark = pytest.mark.gpu


def _same_image(a, b, rtol=0.0):
    """Equal: bit for bit (rtol = 0), or within rtol where the two calls group their fp32 additions
    differently.  The two entry points hand their paths out in different orders; the closest hit of
    a ray does not depend on that (rt_core.h Trav::cull_t)."""
    if rtol == 0.0:
        assert np.array_equal(a, b), np.argwhere((a != b).any(axis=-1))[:4].tolist()
    else:
        bad = (np.abs(a - b) > 1e-6 + rtol * np.abs(b)).any(axis=-1)
        assert not bad.any(), (int(bad.sum()), np.argwhere(bad)[:4].tolist())
    return True


def _pair(sc, cam, spp, seed, sample_begin=0):
    import torch
    H, W = cam.image_height(), cam.image_width
    dev = torch.device("cuda", 0)
    a = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    b = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    sc.render_device_static(cam, spp, a.data_ptr(), stream=stream, sample_begin=sample_begin, seed=seed)
    sc.render_device(cam, spp, b.data_ptr(), stream=stream, sample_begin=sample_begin, seed=seed)
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_static_entry_point_matches_render_device_twice_in_a_row():
    from raytracing_amd import abi, binding
    arrays, cam, _ = load_scene("bouncing_spheres")
    sc = binding.Scene(arrays)
    # first call: the scene's own image size; second call: a smaller image and fewer samples
    # (the block still holds the first call's records and counters)
    a, b = _pair(sc, cam, 8, seed=5)
    assert a.any() and _same_image(a, b)
    small = cam.copy()
    small.image_width = 96
    a2, b2 = _pair(sc, small, 3, seed=6, sample_begin=2)
    assert a2.shape == (small.image_height(), 96, 3) and a2.any() and _same_image(a2, b2)
    # a third, larger again, on a linear-scan scene (another kernel, another scene, same block)
    arrays_q, cam_q, _ = load_scene("cornell_box")
    cam_q.image_width = 128
    sq = binding.Scene(arrays_q)
    a3, b3 = _pair(sq, cam_q, 4, seed=1)
    assert a3.any() and _same_image(a3, b3)
    sq.close()
    # more samples than the block holds records for: several launches, same sums up to grouping
    big = cam.copy()
    big.image_width = 1024
    n_px = big.image_width * big.image_height()
    spp = abi.RT_STATIC_MAX_PIXELS // n_px + 2
    a4, b4 = _pair(sc, big, spp, seed=7)
    # (each grouping of spp fp32 additions is within (spp - 1) * 2^-24 of the exact sum)
    assert _same_image(a4, b4, rtol=2 * (spp - 1) * 2.0 ** -24)
    # accumulation: a second call adds to the image
    import torch
    acc = torch.zeros((small.image_height(), 96, 3), dtype=torch.float32, device="cuda:0")
    sc.render_device_static(small, 2, acc.data_ptr(), sample_begin=2, seed=6)
    sc.render_device_static(small, 1, acc.data_ptr(), sample_begin=4, seed=6)
    ref = torch.zeros_like(acc)
    sc.render_device(small, 2, ref.data_ptr(), sample_begin=2, seed=6)
    sc.render_device(small, 1, ref.data_ptr(), sample_begin=4, seed=6)
    torch.cuda.synchronize()
    assert _same_image(acc.cpu().numpy(), ref.cpu().numpy())
    sc.close()


def test_static_entry_point_rejects_what_it_cannot_hold():
    from raytracing_amd import abi, binding
    import torch
    arrays, cam, _ = load_scene("quads")
    sc = binding.Scene(arrays)
    huge = cam.copy()
    huge.image_width, huge.aspect_ratio = 2100, 1.0  # 4.41 M pixels > RT_STATIC_MAX_PIXELS
    buf = torch.zeros(16, dtype=torch.float32, device="cuda:0")  # never touched: the call must fail first
    with pytest.raises(binding.RtError) as e:
        sc.render_device_static(huge, 1, buf.data_ptr())
    assert e.value.code == abi.RT_ERR_ARG and "static scratch" in str(e.value)
    torch.cuda.synchronize()
    assert not buf.any()
    sc.close()


def test_static_entry_point_allocates_nothing_and_can_be_captured(monkeypatch):
    """What the entry point exists for: no device memory appears or disappears across its calls (free
    memory as the driver reports it is unchanged), and a call can be recorded into a HIP graph (a
    hipMalloc, a stream or an event created during capture would fail it) and replayed.  The scene is
    read from HBM with four stack entries per lane in LDS, so that the spill column of the static
    path (allocated with the scene) is in use."""
    import torch
    from raytracing_amd import abi, binding
    arrays, cam, _ = load_scene("bouncing_spheres")
    cam.image_width = 256
    monkeypatch.setenv("RT_AMD_LDS_STACK", "4")  # read by rt_scene_create
    sc = binding.Scene(arrays, flags=abi.RT_SCENE_FORCE_BVH | abi.RT_SCENE_FORCE_GLOBAL)
    monkeypatch.delenv("RT_AMD_LDS_STACK")
    assert sc.info().lds_resident == 0
    H, W = cam.image_height(), cam.image_width
    dev = torch.device("cuda", 0)
    a = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    b = torch.zeros_like(a)
    c = torch.zeros_like(a)
    stream = torch.cuda.current_stream().cuda_stream
    sc.render_device(cam, 6, b.data_ptr(), stream=stream, seed=9)           # the reference image
    sc.render_device_static(cam, 6, a.data_ptr(), stream=stream, seed=9)   # first static call on this scene
    torch.cuda.synchronize()
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        a.zero_()
        sc.render_device_static(cam, 6, a.data_ptr(), stream=stream, seed=9)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, (free0, free1)
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    # the same call recorded into a graph on a side stream, replayed twice
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        sc.render_device_static(cam, 6, c.data_ptr(), stream=side.cuda_stream, seed=9)
    for _ in range(2):
        c.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(c.cpu().numpy(), b.cpu().numpy())
    del graph
    sc.close()
