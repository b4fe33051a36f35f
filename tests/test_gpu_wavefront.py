"""GPU: the wavefront variant (RT_MODE_WAVEFRONT) against the megakernel.

Both variants run the same per-ray code with the same counter-based sampler keyed by
(seed, pixel, sample), and the per-path records are summed in sample order, so the fp32
accumulation images must be identical bit for bit, and so must the ray counts."""
import numpy as np
import pytest

from raytracing_amd import abi
from tests.conftest import SCENES, load_scene

pytestmark = pytest.mark.gpu


def _binding():
    from raytracing_amd import binding
    return binding


@pytest.mark.parametrize("scene", SCENES)
def test_wavefront_equals_megakernel(scene):
    B = _binding()
    arrays, cam, _ = load_scene(scene)
    cam.image_width = 200 if scene != "cornell_box" else 150
    spp = 8
    sc = B.Scene(arrays)
    mega, st_m = sc.render(cam, spp, seed=11)
    wave, st_w = sc.render(cam, spp, seed=11, mode=abi.RT_MODE_WAVEFRONT)
    assert st_w.paths == st_m.paths
    assert st_w.rays == st_m.rays
    assert np.array_equal(mega, wave), f"max diff {np.abs(mega - wave).max()}"
    assert 1 <= st_w.sched[0] <= 8 * (cam.max_depth + 8)  # rounds: a few more than the longest path
    sc.close()


def test_wavefront_few_slots_and_sample_ranges(monkeypatch):
    """Fewer path slots than paths (slots are recycled) and a split sample range."""
    B = _binding()
    arrays, cam, _ = load_scene("bouncing_spheres")
    cam.image_width = 128
    sc = B.Scene(arrays)
    ref, st = sc.render(cam, 6, seed=5)
    monkeypatch.setenv("RT_AMD_WF_SLOTS", "4096")
    acc, s1 = sc.render(cam, 2, sample_begin=0, seed=5, mode=abi.RT_MODE_WAVEFRONT)
    acc2, s2 = sc.render(cam, 4, sample_begin=2, seed=5, mode=abi.RT_MODE_WAVEFRONT)
    assert s1.rays + s2.rays == st.rays
    full, _ = sc.render(cam, 6, seed=5, mode=abi.RT_MODE_WAVEFRONT)
    assert np.array_equal(full, ref)
    # two partial sums added in a different association: equal to rounding only
    np.testing.assert_allclose(acc + acc2, ref, rtol=1e-5, atol=1e-5)
    sc.close()


def test_wavefront_depth_zero_is_black():
    B = _binding()
    arrays, cam, _ = load_scene("quads")
    cam.image_width = 64
    cam.max_depth = 0
    sc = B.Scene(arrays)
    acc, st = sc.render(cam, 2, mode=abi.RT_MODE_WAVEFRONT)
    assert not acc.any() and st.rays == 0
    sc.close()


def test_wavefront_rejects_counters():
    B = _binding()
    arrays, cam, _ = load_scene("quads")
    cam.image_width = 64
    sc = B.Scene(arrays)
    with pytest.raises(B.RtError) as e:
        sc.render(cam, 1, counters=True, mode=abi.RT_MODE_WAVEFRONT)
    assert e.value.code == abi.RT_ERR_UNSUPPORTED
    sc.close()


def test_schedulers_and_kernel_variants_agree(monkeypatch):
    """BVH scenes can be rendered by two persistent kernels (RT_AMD_SCHEDULER: 0 whole ray,
    1 decoupled), each in a full and a lean build (the latter without
    the quad / noise / image branches, chosen for scenes that have none of them): all four
    run the same per-ray code and must give the same image bit for bit."""
    B = _binding()
    arrays, cam, _ = load_scene("bouncing_spheres")
    cam.image_width = 160
    images = []
    for no_lean in ("0", "1"):
        monkeypatch.setenv("RT_AMD_NO_LEAN", no_lean)   # read by rt_scene_create
        sc = B.Scene(arrays)
        for sched in ("0", "1"):
            monkeypatch.setenv("RT_AMD_SCHEDULER", sched)  # read by rt_render
            acc, st = sc.render(cam, 6, seed=9)
            images.append((no_lean, sched, acc, st.rays))
        sc.close()
    for no_lean, sched, acc, rays in images[1:]:
        assert rays == images[0][3], (no_lean, sched)
        assert np.array_equal(acc, images[0][2]), (no_lean, sched)
    # ... and with or without the per-wave LDS rings of camera rays, the scene copy in LDS, or
    # the tree built over an 18-quad scene that would otherwise be scanned
    monkeypatch.delenv("RT_AMD_NO_LEAN")
    for knob in ("RT_AMD_RING", "RT_AMD_FORCE_GLOBAL"):
        monkeypatch.setenv(knob, "0" if knob == "RT_AMD_RING" else "1")   # read by rt_scene_create
        sc = B.Scene(arrays)
        for sched in ("0", "1"):
            monkeypatch.setenv("RT_AMD_SCHEDULER", sched)
            acc, st = sc.render(cam, 6, seed=9)
            assert st.rays == images[0][3] and np.array_equal(acc, images[0][2]), (knob, sched)
        sc.close()
        monkeypatch.delenv(knob)
    monkeypatch.delenv("RT_AMD_SCHEDULER")
    arrays, cam, _ = load_scene("cornell_box")
    cam.image_width = 96
    ref = None
    for ring in ("1", "0"):
        monkeypatch.setenv("RT_AMD_RING", ring)
        sc = B.Scene(arrays)
        acc, st = sc.render(cam, 5, seed=2)
        sc.close()
        if ref is None:
            ref = (acc, st.rays)
        assert st.rays == ref[1] and np.array_equal(acc, ref[0])
