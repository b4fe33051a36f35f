"""GPU parity on random scenes (tests/scene_fuzz.py): every primitive, material, texture and
wrapper kind at once, through the scan, the host-built tree and the device-built tree, the
megakernel and the wavefront variant, against the oracle's emulation of the reference."""
import numpy as np
import pytest

from raytracing_amd import abi
from tests import oracle_lib as O
from tests.scene_fuzz import random_rays, random_scene

pytestmark = pytest.mark.gpu


def _binding():
    from raytracing_amd import binding
    return binding


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_random_scene_trace_and_render(seed):
    B = _binding()
    arrays, cam = random_scene(seed, n_prims=24 + 30 * (seed - 10))   # 54 .. 144: scan and tree sizes
    rays = random_rays(seed, 60_000)
    osc = O.OracleScene(arrays)
    ref, _ = osc.trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    m = ref["prim"] >= 0
    want, ost = osc.render(cam, 4, sampler=O.SAMPLER_CB, trace_mode=O.TRACE_REFERENCE, seed=seed, n_threads=8)
    images = []
    for flags in (0, abi.RT_SCENE_FORCE_LINEAR, abi.RT_SCENE_FORCE_BVH | abi.RT_SCENE_BUILD_DEVICE,
                  abi.RT_SCENE_FORCE_BVH | abi.RT_SCENE_FORCE_GLOBAL):
        sc = B.Scene(arrays, flags=flags)
        hits = sc.trace(rays)
        assert np.array_equal(hits["prim"], ref["prim"]), flags
        for f in ("t", "p", "normal", "front_face"):
            assert np.array_equal(hits[f][m], ref[f][m]), (flags, f)
        du = np.abs(hits["u"][m] - ref["u"][m])
        du = np.minimum(du, 1.0 - du)
        assert np.array_equal(np.isnan(hits["v"][m]), np.isnan(ref["v"][m]))
        assert np.nanmax(du) <= 2e-6 and np.nanmax(np.abs(hits["v"][m] - ref["v"][m])) <= 2e-6
        for mode in (abi.RT_MODE_MEGAKERNEL, abi.RT_MODE_WAVEFRONT):
            acc, st = sc.render(cam, 4, seed=seed, mode=mode)
            assert st.paths == ost.paths
            images.append(acc)
            bad = float((np.abs(acc - want) > 1e-4 * np.maximum(1.0, np.abs(want))).any(axis=2).mean())
            assert bad < 0.04, (flags, mode, bad)
            assert abs(acc.mean() - want.mean()) <= 0.02 * want.mean()
        sc.close()
    # every variant traces the same paths with the same arithmetic
    for img in images[1:]:
        assert np.array_equal(img, images[0])
