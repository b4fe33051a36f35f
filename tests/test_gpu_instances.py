"""GPU parity of two-level scenes (rt_instance: sub-scenes placed several times, hittable.h:51-344
applied to a shared object): the instanced scene must give what the same scene with every
instance written out as primitives gives — closest hits bit for bit against the oracle, images bit
for bit against the written-out scene on the GPU — from one copy of each sub-scene."""
import numpy as np
import pytest

from raytracing_amd import abi
from tests import oracle_lib as O
from tests.test_abi_and_host import check_two_level_hits, two_level_rays, two_level_scene

pytestmark = pytest.mark.gpu


def _binding():
    from raytracing_amd import binding
    return binding


def test_instanced_scene_traces_like_the_written_out_one(monkeypatch):
    B = _binding()
    scene, written_out, origin = two_level_scene()
    rays = two_level_rays(300_000)
    ref, _ = O.OracleScene(written_out).trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    assert (ref["prim"] >= 0).mean() > 0.3
    for flags, lds_stack in ((0, None), (abi.RT_SCENE_FORCE_GLOBAL, None), (abi.RT_SCENE_FORCE_GLOBAL, "4")):
        if lds_stack:
            monkeypatch.setenv("RT_AMD_LDS_STACK", lds_stack)  # read by rt_scene_create: stacks spill to HBM
        sc = B.Scene(scene, flags=flags)
        if lds_stack:
            monkeypatch.delenv("RT_AMD_LDS_STACK")
        info = sc.info()
        assert info.n_prims == len(scene.prims) and 0 < info.n_nodes < 60
        assert info.lds_resident == (0 if flags & abi.RT_SCENE_FORCE_GLOBAL else 1)
        for sched in (False, True):  # the whole-ray kernel and the render kernel's own traversal
            check_two_level_hits(sc.trace(rays, sched=sched), ref, origin)
        full = sc.trace(rays, sched=True)
        prim, t = sc.closest_hits(rays)  # the ids-only entry point: the same primitive and parameter
        assert np.array_equal(prim, full["prim"]) and np.array_equal(t[prim >= 0], full["t"][prim >= 0])
        sc.close()


def test_two_level_scene_built_on_the_device():
    """RT_SCENE_BUILD_DEVICE on a scene with instances: every tree of it (one per distinct sub-scene, one over
    the loose primitives and the instances' boxes) comes from the device-side builder, and the closest
    hits are still those of the written-out scene, bit for bit."""
    B = _binding()
    scene, written_out, origin = two_level_scene()
    rays = two_level_rays(200_000)
    ref, _ = O.OracleScene(written_out).trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    for flags in (abi.RT_SCENE_BUILD_DEVICE, abi.RT_SCENE_BUILD_DEVICE | abi.RT_SCENE_FORCE_GLOBAL):
        sc = B.Scene(scene, flags=flags)
        info = sc.info()
        assert info.built_on_device == 1 and info.n_nodes > 0
        for sched in (False, True):
            check_two_level_hits(sc.trace(rays, sched=sched), ref, origin)
        sc.close()


def test_instanced_scene_renders_like_the_written_out_one():
    B = _binding()
    scene, written_out, _ = two_level_scene()
    cam = abi.CameraDesc(16 / 9, 320, 8, 12, (0.6, 0.7, 0.9), 45.0, (30, 18, 42), (0, 2, 0), (0, 1, 0), 0.0, 10.0)
    a = B.Scene(scene)
    b = B.Scene(written_out)
    assert a.info().device_bytes < 0.4 * b.info().device_bytes  # 58 primitives kept against 10 + 15 x 48 + 1 + 3
    img_a, st_a = a.render(cam, 8, seed=3)
    img_b, st_b = b.render(cam, 8, seed=3)
    assert img_a.std() > 0 and st_a.paths == st_b.paths
    # the same paths: equal images and ray counts (a hit that rounding alone decides may fall either
    # way with the order boxes are visited in, which differs between the two trees: a few pixels)
    differ = (img_a != img_b).any(axis=2).mean()
    assert differ < 1e-3 and abs(int(st_a.rays) - int(st_b.rays)) <= 1e-4 * st_b.rays, (differ, st_a.rays, st_b.rays)
    # the whole-ray scheduler and a second render of the same thing
    img_c, st_c = a.render(cam, 8, seed=3)
    assert np.array_equal(img_a, img_c) and st_a.rays == st_c.rays
    # the wavefront variant carries the instance of a hit in its path records: the same image
    img_w, st_w = a.render(cam, 8, seed=3, mode=abi.RT_MODE_WAVEFRONT)
    assert np.array_equal(img_a, img_w) and st_a.rays == st_w.rays
    a.close()
    b.close()


def test_instanced_scene_through_the_other_entry_points():
    """The same two-level scene through the entry point that allocates nothing and through the
    multi-GPU replicas (device 0 listed twice): the image of the plain render (the replicas add
    their halves of the samples, so theirs agrees up to the order of fp32 additions)."""
    import torch
    B = _binding()
    scene, _, _ = two_level_scene()
    cam = abi.CameraDesc(16 / 9, 256, 8, 12, (0.6, 0.7, 0.9), 45.0, (30, 18, 42), (0, 2, 0), (0, 1, 0), 0.0, 10.0)
    sc = B.Scene(scene)
    ref, st = sc.render(cam, 6, seed=9)
    acc = torch.zeros((cam.image_height(), cam.image_width, 3), dtype=torch.float32, device="cuda:0")
    sc.render_device_static(cam, 6, acc.data_ptr(), seed=9)
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy(), ref)
    sc.close()
    multi = B.MultiScene(scene, [0, 0])
    img, mst = multi.render(cam, 6, seed=9)
    multi.close()
    assert mst.rays == st.rays
    assert (np.abs(img - ref) > 1e-5 * np.maximum(1.0, np.abs(ref))).any(axis=2).mean() < 1e-4
