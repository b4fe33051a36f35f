"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares, the
host half (camera, tonemap, scene files, error paths) behaves like the reference, and the
kernel sources' per-ray logic (compiled for the host by tests/cpu_sim, test-only) agrees with
the oracle.  No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracing_amd import abi, binding
from tests import oracle_lib as O, sim_lib as S
from tests.conftest import GOLDEN, ROOT, SCENES, load_scene


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes found in rt_amd.h"
    L = binding.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"librt_amd.so does not export {name}"
    assert declared == set(binding.EXPORTS)
    assert L.rt_abi_version() == abi.RT_AMD_ABI_VERSION


def test_struct_sizes_match_header():
    # sizes the C compiler gives the structs (checked when the oracle / sim were compiled) are
    # mirrored by ctypes; the fixture files written by C code parse with these layouts
    assert C.sizeof(abi.CameraDesc) == 76 and C.sizeof(abi.CameraFrame) == 120
    assert C.sizeof(abi.RenderOpts) == 32
    arrays, cam, frame = load_scene("bouncing_spheres")
    assert len(arrays.prims) == 485 and cam.image_width == 400 and frame.image_height == 225


@pytest.mark.parametrize("scene", SCENES)
def test_library_camera_initialize_matches_reference(scene):
    arrays, cam, ref_frame = load_scene(scene)
    assert bytes(binding.camera_initialize(cam)) == bytes(ref_frame)


def test_camera_argument_errors():
    cam = abi.CameraDesc()
    with pytest.raises(binding.RtError) as e:
        binding.camera_initialize(cam)  # image_width 0
    assert e.value.code == abi.RT_ERR_ARG


def test_tonemap_matches_write_color():
    rng = np.random.default_rng(0)
    acc = rng.uniform(-0.5, 3.0, size=(64, 64, 3)).astype(np.float32)
    acc[0, 0] = (0, 1.0, 0.25)
    a = binding.tonemap_u8(acc, 0.5)
    assert np.array_equal(a, O.tonemap_u8(acc, 0.5))
    assert a[0, 0].tolist() == [0, int(256 * np.float32(np.sqrt(np.float32(0.5)))), int(256 * np.sqrt(0.125))]
    assert a.max() == 255  # 0.999 * 256 = 255.7


def test_scene_file_roundtrip(tmp_path):
    arrays, cam, _ = load_scene("cornell_box")
    p = str(tmp_path / "x.rtscene")
    d = arrays.desc()
    binding.check(binding.lib().rt_scene_file_save(C.byref(d), p.encode()))
    assert open(p, "rb").read() == open(os.path.join(GOLDEN, "cornell_box.rtscene"), "rb").read()
    out = C.POINTER(abi.SceneDesc)()
    binding.check(binding.lib().rt_scene_file_load(p.encode(), C.byref(out)))
    assert out.contents.n_prims == 18 and out.contents.n_chains == 2
    binding.lib().rt_scene_file_free(out)
    assert binding.lib().rt_scene_file_load(b"/nonexistent", C.byref(out)) == abi.RT_ERR_IO
    back = abi.SceneArrays.load(p)
    assert back.prims.tobytes() == arrays.prims.tobytes()


def test_no_gpu_means_loud_failure():
    """The product has no CPU path: without a device, scene creation fails with RT_ERR_HIP."""
    if binding.device_count() > 0:
        pytest.skip("a GPU is visible")
    arrays, cam, _ = load_scene("quads")
    with pytest.raises(binding.RtError) as e:
        binding.Scene(arrays)
    assert e.value.code == abi.RT_ERR_HIP


def test_descriptor_validation():
    arrays, cam, _ = load_scene("quads")
    bad = abi.SceneArrays(prims=arrays.prims.copy(), materials=arrays.materials, textures=arrays.textures)
    bad.prims["material"][2] = 99
    with pytest.raises(binding.RtError) as e:
        binding.Scene(bad)
    assert e.value.code == abi.RT_ERR_ARG and "material" in str(e.value)


def test_descriptor_validation_rejects_malformed_counts_and_arrays():
    """Negative counts, null arrays with a positive count, unknown wrapper ops and image indices
    below -1 are refused before anything is copied (the checks run ahead of the device check)."""
    arrays, cam, _ = load_scene("cornell_box")
    h = C.c_void_p()

    def create(d):
        return binding.lib().rt_scene_create(C.byref(d), 0, 0, C.byref(h))

    for field in ("n_prims", "n_groups", "n_materials", "n_textures", "n_perlins", "n_images", "n_xform_ops",
                  "n_chains"):
        d = arrays.desc()
        setattr(d, field, -3)
        assert create(d) == abi.RT_ERR_ARG, field
        assert b"negative" in binding.lib().rt_last_error()
    for field in ("prims", "materials", "textures", "xform_ops", "chains"):
        d = arrays.desc()
        setattr(d, field, None)
        assert create(d) == abi.RT_ERR_ARG, field
        assert b"null array" in binding.lib().rt_last_error()
    bad = abi.SceneArrays(prims=arrays.prims, materials=arrays.materials, textures=arrays.textures,
                          xform_ops=arrays.xform_ops.copy(), chains=arrays.chains)
    bad.xform_ops["type"][0] = 7
    assert create(bad.desc()) == abi.RT_ERR_ARG and b"wrapper op" in binding.lib().rt_last_error()
    bad = abi.SceneArrays(prims=arrays.prims, materials=arrays.materials, textures=arrays.textures,
                          xform_ops=arrays.xform_ops, chains=arrays.chains.copy())
    bad.chains["first_op"][1] = 2**31 - 1  # first_op + n_ops would wrap in 32 bits
    assert create(bad.desc()) == abi.RT_ERR_ARG and b"chain" in binding.lib().rt_last_error()
    e_arrays, _, _ = load_scene("earth")
    bad = abi.SceneArrays(prims=e_arrays.prims, materials=e_arrays.materials, textures=e_arrays.textures.copy(),
                          images=e_arrays.images, texels=e_arrays.texels)
    k = int(np.nonzero(bad.textures["type"] == abi.RT_TEX_IMAGE)[0][0])
    bad.textures["image"][k] = -2
    assert create(bad.desc()) == abi.RT_ERR_ARG and b"image" in binding.lib().rt_last_error()
    bad.textures["image"][k] = 0
    bad.images = e_arrays.images.copy()
    bad.images["offset"][0] = 2**63  # offset + size would wrap in 64 bits
    assert create(bad.desc()) == abi.RT_ERR_ARG and b"texels" in binding.lib().rt_last_error()


def test_scene_file_with_inconsistent_counts_is_refused(tmp_path):
    """The counts in a .rtscene header are checked against the file before anything is
    allocated: wrapping sizes, counts beyond int32 and truncated files all fail to load."""
    blob = bytearray(open(os.path.join(GOLDEN, "quads.rtscene"), "rb").read())
    out = C.POINTER(abi.SceneDesc)()

    def load(b):
        p = str(tmp_path / "bad.rtscene")
        open(p, "wb").write(bytes(b))
        return binding.lib().rt_scene_file_load(p.encode(), C.byref(out))

    assert load(blob) == abi.RT_OK
    binding.lib().rt_scene_file_free(out)
    for slot, value in ((0, 2**64 // 64 + 5),  # sizeof(rt_prim) * count wraps to a small number
                        (2, 2**32 + 5),        # would be truncated to 5 in the int32 field
                        (6, 2**62),            # texel bytes far beyond the file
                        (1, 2**31)):
        b = bytearray(blob)
        b[8 + 8 * slot:16 + 8 * slot] = int(value).to_bytes(8, "little")
        assert load(b) == abi.RT_ERR_IO, slot
    assert load(blob[:-8]) == abi.RT_ERR_IO   # truncated
    assert load(blob + b"\0" * 8) == abi.RT_ERR_IO  # trailing bytes the counts do not account for


# ---- kernel logic on the host (same headers the HIP kernels are built from) -------------------

@pytest.mark.parametrize("scene", SCENES)
def test_kernel_logic_matches_reference_records(scene):
    arrays, cam, _ = load_scene(scene)
    rays, ref_hits, tmin, tmax = O.load_rays_file(os.path.join(GOLDEN, scene + ".rays"))
    # (cornell_box under a forced BVH: its rotated / scaled boxes are tested in object space at
    # the leaves, exactly like the linear scan, so that is bit-exact as well)
    # ... with each of the kernels' stack variants: 32-bit entries, the 16-bit entries of scenes held
    # in LDS, and 32-bit entries that overflow into a second column (deep trees read from HBM)
    try:
        for flags in (0, abi.RT_SCENE_FORCE_LINEAR, abi.RT_SCENE_FORCE_BVH):
            for mode in ((0,) if flags == abi.RT_SCENE_FORCE_LINEAR else (0, 1, 2)):
                S.set_stack_mode(mode)
                hits, _, _ = S.SimScene(arrays, flags).trace(rays, tmin, tmax)
                assert hits.tobytes() == ref_hits.tobytes(), f"flags {flags} stack mode {mode}"
    finally:
        S.set_stack_mode(0)


@pytest.mark.parametrize("scene", ["synth60_list", "synth60_bvh"])
def test_kernel_logic_matches_reference_records_of_the_synthetic_grid(scene):
    """The kernels' own tree and traversal on the reference's records of the synthetic grid scene
    (14 401 primitives, camera 136 units away: the distances of the 4K configuration), whichever way
    the reference held the scene: every field of every record, bit for bit."""
    arrays, cam, _ = load_scene(scene)
    rays, ref_hits, tmin, tmax = O.load_rays_file(os.path.join(GOLDEN, scene + ".rays"))
    try:
        for mode in (0, 1, 2):
            S.set_stack_mode(mode)
            hits, _, _ = S.SimScene(arrays).trace(rays, tmin, tmax)
            assert hits.tobytes() == ref_hits.tobytes(), mode
    finally:
        S.set_stack_mode(0)


def test_kernel_logic_bvh_never_drops_a_hit():
    arrays, cam, _ = load_scene("bouncing_spheres")
    rng = np.random.default_rng(3)
    n = 200_000
    rays = np.zeros(n, abi.RAY_DT)
    o = rng.normal(size=(n, 3))
    o /= np.linalg.norm(o, axis=1, keepdims=True)
    o *= rng.uniform(5, 60, size=(n, 1))
    o[:, 1] = np.abs(o[:, 1]) * 0.3 + 0.05
    tgt = np.stack([rng.uniform(-11, 11, n), rng.uniform(0, 1.0, n), rng.uniform(-11, 11, n)], 1)
    rays["orig"], rays["dir"], rays["tm"] = o, (tgt - o) * rng.uniform(0.2, 3, size=(n, 1)), rng.uniform(0, 1, n)
    # axis-parallel rays (zero direction components) exercise the guarded reciprocal
    rays["dir"][:1000, 0] = 0
    rays["dir"][1000:2000, 1] = 0
    rays["dir"][2000:2500, [0, 2]] = 0
    sim = S.SimScene(arrays)
    # 4-wide tree collapsed from the binary one (484 inner nodes): between a third and all of them
    assert 484 / 3 <= sim.n_nodes < 300 and sim.depth <= 12
    ref, _ = O.OracleScene(arrays).trace(rays, 0.001, np.inf, O.TRACE_BAKED)  # scan of all 485
    try:
        for mode in (0, 1, 2):
            S.set_stack_mode(mode)
            hits, box, prim = sim.trace(rays)
            assert np.array_equal(hits["prim"], ref["prim"]) and np.array_equal(hits["t"], ref["t"]), mode
            assert prim / n < 5  # the tree culls: a handful of exact tests per ray instead of 485
            assert box / n < 40
    finally:
        S.set_stack_mode(0)


def _rays_through_a_sphere_field(n, seed, reach):
    """Origins `reach` away, aimed into the field of small spheres at ground level."""
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, abi.RAY_DT)
    o = rng.normal(size=(n, 3))
    o /= np.linalg.norm(o, axis=1, keepdims=True)
    o *= rng.uniform(0.2 * reach, reach, size=(n, 1))
    o[:, 1] = np.abs(o[:, 1]) * 0.3 + 0.05
    tgt = np.stack([rng.uniform(-11, 11, n), rng.uniform(0, 1.0, n), rng.uniform(-11, 11, n)], 1)
    rays["orig"], rays["dir"], rays["tm"] = o, (tgt - o) * rng.uniform(0.2, 3, size=(n, 1)), rng.uniform(0, 1, n)
    return rays


def test_kernel_logic_closest_hit_does_not_depend_on_the_schedule(tmp_path):
    """The decoupled kernels test a leaf some node steps after they popped it, so the best_t a box is
    culled against differs from launch to launch.  Certificate per ray (tests/cpu_sim
    sim_schedule_certificate): the winner among ALL leaves the ray reaches without culling survives
    the box tests of its own path against its own parameter, the smallest best_t any schedule can
    hold.  Far from small spheres that needs the slack of Trav::cull_t and the own-box rule for
    sphere hits: without the slack the same rays are at risk by the thousand."""
    import subprocess
    arrays, _, _ = load_scene("bouncing_spheres")
    sim = S.SimScene(arrays)
    for reach, seed in ((60, 3), (400, 4)):
        risky, risky_without_slack, differs, leaves = sim.schedule_certificate(_rays_through_a_sphere_field(100_000, seed, reach))
        assert risky == 0 and differs == 0, (reach, risky, differs)
    assert risky_without_slack > 0  # (400 units away the rounding of sphere.h:47-52 reaches the radius)
    # the synthetic grid scene at the distances of its 4K camera: camera rays, and rays leaving the hit points
    rt_scenes = os.path.join(ROOT, "examples", "bin", "rt_scenes")
    out = str(tmp_path / "synth.rtscene")
    subprocess.run([rt_scenes, "synth1m", "--half", "150", "--dump", out], check=True, capture_output=True)
    arrays, cam = abi.SceneArrays.load(out), abi.load_camera_file(out + ".cam")[0]
    sim = S.SimScene(arrays)
    f = binding.camera_initialize(cam)
    rng = np.random.default_rng(8)
    n = 150_000
    px, py = rng.integers(0, f.image_width, n), rng.integers(0, f.image_height, n)
    tgt = (np.array(f.pixel00_loc, np.float32) + px[:, None].astype(np.float32) * np.array(f.pixel_delta_u, np.float32)
           + py[:, None].astype(np.float32) * np.array(f.pixel_delta_v, np.float32))
    rays = np.zeros(n, abi.RAY_DT)
    rays["orig"] = np.array(f.center, np.float32)
    rays["dir"] = tgt - rays["orig"]
    risky, risky_without_slack, differs, _ = sim.schedule_certificate(rays)
    assert risky == 0 and differs == 0 and risky_without_slack > 100, (risky, risky_without_slack, differs)
    hits, _, _ = sim.trace(rays)
    m = hits["prim"] >= 0
    d = rng.normal(size=(int(m.sum()), 3)).astype(np.float32)
    d[:, 1] = np.abs(d[:, 1]) * 0.2
    second = np.zeros(int(m.sum()), abi.RAY_DT)
    second["orig"], second["dir"] = hits["p"][m], d
    risky, _, differs, _ = sim.schedule_certificate(second)
    assert risky == 0 and differs == 0, (risky, differs)


def instanced_scene(n=96, seed=11):
    """Spheres (some moving) and quads inside translate / rotate_y / rotate_xyz / non-uniform
    scaling wrappers, enough of them for a BVH: the case the reference handles with
    hittable.h:51-344 and a baked tree cannot."""
    rng = np.random.default_rng(seed)
    prims = np.zeros(n, abi.PRIM_DT)
    ops, chains = [], []
    for i in range(n):
        c = rng.uniform(-1, 1, 3)
        if i % 3 == 2:
            prims[i]["type"] = abi.RT_PRIM_QUAD
            prims[i]["a"], prims[i]["b"], prims[i]["c"] = c, rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        else:
            prims[i]["type"] = abi.RT_PRIM_SPHERE
            prims[i]["a"], prims[i]["c"] = c, (rng.uniform(0.2, 0.7), 0, 0)
            if i % 4 == 0:
                prims[i]["flags"] = abi.RT_PRIM_FLAG_MOVING
                prims[i]["b"] = rng.uniform(-0.3, 0.3, 3)
                prims[i]["d"] = prims[i]["a"] + prims[i]["b"]
        if i % 5 == 4:
            prims[i]["xform"] = -1
            prims[i]["a"] += rng.uniform(-12, 12, 3).astype(np.float32)
            continue
        first = len(ops)
        kind = i % 4
        if kind == 0:
            ops.append((abi.RT_XF_ROTATE_Y, (rng.uniform(-180, 180), 0, 0)))
        elif kind == 1:
            ops.append((abi.RT_XF_ROTATE_XYZ, tuple(rng.uniform(-90, 90, 3))))
            ops.append((abi.RT_XF_SCALE, tuple(rng.uniform(0.5, 2.0, 3))))
        elif kind == 2:
            ops.append((abi.RT_XF_SCALE, (1.5, 1.5, 1.5)))
        else:
            ops.append((abi.RT_XF_ROTATE_XYZ, tuple(rng.uniform(-90, 90, 3))))
            ops.append((abi.RT_XF_ROTATE_Y, (rng.uniform(-180, 180), 0, 0)))
        ops.append((abi.RT_XF_TRANSLATE, tuple(rng.uniform(-12, 12, 3))))
        chains.append((first, len(ops) - first))
        prims[i]["xform"] = len(chains) - 1
    mats = np.zeros(1, abi.MATERIAL_DT)
    tex = np.zeros(1, abi.TEXTURE_DT)
    tex["color"] = (0.7, 0.6, 0.5)
    return abi.SceneArrays(prims=prims, materials=mats, textures=tex,
                           xform_ops=np.array(ops, abi.XFORM_OP_DT), chains=np.array(chains, abi.CHAIN_DT))


def instanced_rays(n, seed=4):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, abi.RAY_DT)
    o = rng.normal(size=(n, 3))
    o *= rng.uniform(20, 40, size=(n, 1)) / np.linalg.norm(o, axis=1, keepdims=True)
    rays["orig"], rays["dir"], rays["tm"] = o, rng.uniform(-13, 13, (n, 3)) - o, rng.uniform(0, 1, n)
    return rays


def test_kernel_logic_instances_inside_a_bvh():
    arrays = instanced_scene()
    rays = instanced_rays(30_000)
    ref, _ = O.OracleScene(arrays).trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    assert (ref["prim"] >= 0).mean() > 0.05
    sim = S.SimScene(arrays)
    assert (len(arrays.prims) - 1) / 3 <= sim.n_nodes <= len(arrays.prims) - 1
    hits, _, prim_tests = sim.trace(rays)
    assert hits.tobytes() == ref.tobytes()
    assert prim_tests / len(rays) < 12
    lin, _, _ = S.SimScene(arrays, abi.RT_SCENE_FORCE_LINEAR).trace(rays)
    assert lin.tobytes() == ref.tobytes()


def test_kernel_logic_tie_rules():
    """Coincident primitives: the winner is the one the reference's sequential scan keeps
    (a later quad replaces, a later sphere does not), whatever order the tree visits them in."""
    mats = np.zeros(1, abi.MATERIAL_DT)
    mats["texture"] = 0
    tex = np.zeros(1, abi.TEXTURE_DT)
    rng = np.random.default_rng(5)
    for trial in range(20):
        n = 40
        prims = np.zeros(n, abi.PRIM_DT)
        prims["xform"] = -1
        kinds = rng.integers(0, 2, n)
        for i in range(n):
            if kinds[i]:  # one of 4 distinct quads, many duplicates
                k = rng.integers(0, 4)
                prims[i]["type"] = abi.RT_PRIM_QUAD
                prims[i]["a"], prims[i]["b"], prims[i]["c"] = (-2, -2, float(k)), (4, 0, 0), (0, 4, 0)
            else:
                k = rng.integers(0, 4)
                prims[i]["type"] = abi.RT_PRIM_SPHERE
                prims[i]["a"], prims[i]["c"] = (0, 0, float(k) + 10), (1.0, 0, 0)
        arrays = abi.SceneArrays(prims=prims, materials=mats, textures=tex)
        rays = np.zeros(64, abi.RAY_DT)
        rays["orig"] = np.c_[rng.uniform(-1, 1, 64), rng.uniform(-1, 1, 64), np.full(64, 20.0)]
        rays["dir"] = (0, 0, -1)
        ref, _ = O.OracleScene(arrays).trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
        for flags in (abi.RT_SCENE_FORCE_BVH, abi.RT_SCENE_FORCE_LINEAR):
            hits, _, _ = S.SimScene(arrays, flags).trace(rays)
            assert np.array_equal(hits["prim"], ref["prim"]) and np.array_equal(hits["t"], ref["t"])


def test_kernel_logic_render_matches_oracle():
    for scene in ("bouncing_spheres", "cornell_box", "perlin_spheres", "earth"):
        arrays, cam, _ = load_scene(scene)
        cam.image_width = 48
        acc, rays = S.SimScene(arrays).render(cam, 2, seed=3)
        ref, st = O.OracleScene(arrays).render(cam, 2, sampler=O.SAMPLER_CB, trace_mode=O.TRACE_REFERENCE, seed=3)
        assert rays == st.rays
        assert np.allclose(acc, ref, rtol=2e-6, atol=1e-7)


def test_empty_and_degenerate_scenes():
    arrays = abi.SceneArrays()
    rays = np.zeros(4, abi.RAY_DT)
    rays["dir"] = (0, 0, -1)
    hits, _, _ = S.SimScene(arrays).trace(rays)
    assert (hits["prim"] == -1).all()
    h2, _ = O.OracleScene(arrays).trace(rays)
    assert (h2["prim"] == -1).all()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_kernel_logic_on_random_scenes(seed):
    """Random scenes with every primitive, material, texture and wrapper kind: the kernel logic
    (scan and tree) against the oracle's emulation of the reference, hit records bit for bit
    and a small render path for path."""
    from tests.scene_fuzz import random_rays, random_scene
    arrays, cam = random_scene(seed, n_prims=40 + 25 * seed)
    rays = random_rays(seed, 6000)
    osc = O.OracleScene(arrays)
    ref, _ = osc.trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    assert (ref["prim"] > 0).mean() > 0.05
    for flags in (0, abi.RT_SCENE_FORCE_LINEAR):
        hits, _, _ = S.SimScene(arrays, flags).trace(rays)
        assert hits.tobytes() == ref.tobytes(), flags
    cam.image_width = 40
    acc, n_rays = S.SimScene(arrays).render(cam, 2, seed=seed)
    want, st = osc.render(cam, 2, sampler=O.SAMPLER_CB, trace_mode=O.TRACE_REFERENCE, seed=seed)
    assert n_rays == st.rays
    assert np.allclose(acc, want, rtol=2e-6, atol=1e-7)


def test_kernel_logic_ends_on_rays_that_are_not_numbers():
    """The traversal code compiled for the host: NaN / infinite / zero / huge rays end (a NaN
    passes every box test and walks the whole tree) with the oracle's answer, in every stack
    variant.  An unused child slot therefore names a leaf, never a node."""
    from tests import sim_lib
    arrays, _, _ = load_scene("bouncing_spheres")
    nan, inf = np.nan, np.inf
    cases = [((nan, 0, 0), (0, 0, -1)), ((0, 1, 5), (nan, 0, -1)), ((0, 1, 5), (nan, nan, nan)),
             ((inf, 1, 5), (0, 0, -1)), ((0, 1, 5), (0, inf, -1)), ((0, 1, 5), (0, 0, 0)),
             ((0, 1, 5), (0, 0, -1e38)), ((0, 1, 5), (1e-42, 0, -1e-42)), ((0, 1, 5), (0, 0, -1)),
             ((-inf, inf, nan), (inf, -inf, nan)), ((3e38, 3e38, 3e38), (-1, -1, -1))]
    rays = np.zeros(len(cases), abi.RAY_DT)
    for i, (o, d) in enumerate(cases):
        rays["orig"][i] = o
        rays["dir"][i] = d
    osc = O.OracleScene(arrays)
    with np.errstate(all="ignore"):
        oh, _ = osc.trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    assert oh["prim"][8] >= 0 and (oh["prim"][:3] == -1).all()
    sim = sim_lib.SimScene(arrays)
    try:
        for mode in (0, 1, 2):
            sim_lib.set_stack_mode(mode)
            got = sim.trace(rays, 0.001, np.inf)
            hits = got[0] if isinstance(got, tuple) else got
            assert np.array_equal(hits["prim"], oh["prim"]), mode
            assert ((hits["t"] == oh["t"]) | (hits["prim"] < 0)).all(), mode
    finally:
        sim_lib.set_stack_mode(0)


def axis_aligned_quads_scene(seed=3, n=30):
    """Quads along the coordinate axes in every orientation: normal along x, y or z, u and v in
    either order and with either sign, some inside wrappers, a few general quads and spheres in
    between.  The scan tests the axis-aligned ones with a shortened formula (quad_test_axis)."""
    rng = np.random.default_rng(seed)
    prims = np.zeros(n, abi.PRIM_DT)
    prims["xform"] = -1
    ops, chains = [], []
    for i in range(n):
        q = rng.uniform(-3, 3, 3)
        if i % 7 == 5:
            prims[i]["type"] = abi.RT_PRIM_SPHERE
            prims[i]["a"], prims[i]["c"] = q, (rng.uniform(0.3, 0.9), 0, 0)
            continue
        prims[i]["type"] = abi.RT_PRIM_QUAD
        u, v = np.zeros(3), np.zeros(3)
        if i % 7 == 6:  # a general one
            u, v = rng.uniform(-2, 2, 3), rng.uniform(-2, 2, 3)
        else:
            k = i % 3
            a, b = ((k + 1) % 3, (k + 2) % 3) if (i // 3) % 2 == 0 else ((k + 2) % 3, (k + 1) % 3)
            u[a] = rng.uniform(0.5, 3) * rng.choice([-1, 1])
            v[b] = rng.uniform(0.5, 3) * rng.choice([-1, 1])
        prims[i]["a"], prims[i]["b"], prims[i]["c"] = q, u, v
        if i % 4 == 1:
            first = len(ops)
            ops.append((abi.RT_XF_ROTATE_Y, (rng.uniform(-180, 180), 0, 0)))
            ops.append((abi.RT_XF_TRANSLATE, tuple(rng.uniform(-2, 2, 3))))
            chains.append((first, 2))
            prims[i]["xform"] = len(chains) - 1
    mats = np.zeros(1, abi.MATERIAL_DT)
    tex = np.zeros(1, abi.TEXTURE_DT)
    return abi.SceneArrays(prims=prims, materials=mats, textures=tex,
                           xform_ops=np.array(ops, abi.XFORM_OP_DT), chains=np.array(chains, abi.CHAIN_DT))


def axis_aligned_rays(n, seed=9):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, abi.RAY_DT)
    o = rng.normal(size=(n, 3))
    o *= rng.uniform(6, 12, size=(n, 1)) / np.linalg.norm(o, axis=1, keepdims=True)
    d = rng.uniform(-4, 4, (n, 3)) - o
    d[::5, 0] = 0.0   # rays parallel to a coordinate plane: zero denominators for some quads
    d[1::7] = np.sign(d[1::7]) * np.abs(d[1::7]).max(axis=1, keepdims=True) * (np.abs(d[1::7]) == np.abs(d[1::7]).max(axis=1, keepdims=True))
    rays["orig"], rays["dir"] = o, d
    return rays


def test_kernel_logic_axis_aligned_quads_in_a_scan():
    arrays = axis_aligned_quads_scene()
    rays = axis_aligned_rays(40_000)
    ref, _ = O.OracleScene(arrays).trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    assert (ref["prim"] >= 0).mean() > 0.2 and len(set(ref["prim"].tolist())) > 20
    lin, _, _ = S.SimScene(arrays).trace(rays)
    assert lin.tobytes() == ref.tobytes()


def test_host_logic_under_address_and_ub_sanitizers(tmp_path):
    """The host-side code the library shares with the kernels (flattening, the wide tree, the
    traversal and intersection logic, the scene-file reader), compiled with -fsanitize=address,
    undefined and run over every golden scene in every traversal and stack variant: no report."""
    import glob
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = os.path.join(ROOT, "tests", "sanitize", "host_logic_sanitize.cpp")
    exe = str(tmp_path / "host_logic_sanitize")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-ffp-contract=off", "-pthread", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "raytracing_amd", "csrc"), "-o", exe, src], check=True)
    scenes = sorted(glob.glob(os.path.join(GOLDEN, "*.rtscene")))
    run = subprocess.run([exe] + scenes, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    out = run.stdout + run.stderr
    assert run.returncode == 0, out[-3000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out and "LeakSanitizer" not in out, out[-3000:]
    # (three variants per scene: as the library would hold it, tree forced, scan forced; the scan is
    # not forced on scenes of more than 64 primitives)
    expected = sum(3 if len(abi.SceneArrays.load(f).prims) <= 64 else 2 for f in scenes)
    assert out.count("flags") == expected


def test_damaged_scene_files_end_in_a_refusal_or_a_scene_under_sanitizers(tmp_path):
    """Scene files come from users.  Copies of the golden scenes with random bytes, special
    32-bit values (NaN, infinities, huge numbers, -1, INT_MAX) written over their arrays, changed
    counts, cut or extended, go through what the library does on the host before anything reaches
    the GPU (reader, descriptor checks, flattening, tree builders) and then through the kernels'
    traversal and shading code compiled for the host: each ends as refused or as a scene that
    traces and renders, with no report of -fsanitize=address,undefined, no crash and no hang."""
    import glob
    import random
    import shutil
    import struct
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = os.path.join(ROOT, "tests", "sanitize", "scene_file_fuzz.cpp")
    exe = str(tmp_path / "scene_file_fuzz")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-ffp-contract=off", "-pthread", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "raytracing_amd", "csrc"), "-o", exe, src], check=True)
    special = [0x7fc00000, 0x7f800000, 0xff800000, 0x80000000, 0x7f7fffff, 0xff7fffff, 0xffffffff, 0x7fffffff, 0, 1,
               0x3f800000, 0x00000001, 0x80000001, 0x00800000]
    rng = random.Random(17)
    header = 8 + 10 * 8
    n_ok = n_refused = 0
    for path in sorted(glob.glob(os.path.join(GOLDEN, "*.rtscene"))):
        data = open(path, "rb").read()
        if len(data) > 200_000:
            continue
        files = []
        for k in range(24):
            b = bytearray(data)
            kind = k % 5
            if kind == 0:
                for _ in range(rng.randrange(1, 20)):
                    b[rng.randrange(header, len(b))] = rng.randrange(256)
            elif kind == 1:
                for _ in range(rng.randrange(1, 12)):
                    i = rng.randrange(header // 4, len(b) // 4) * 4
                    b[i:i + 4] = struct.pack("<I", rng.choice(special))
            elif kind == 2:
                i = rng.randrange(header // 4, len(b) // 4) * 4
                v = struct.pack("<I", rng.choice(special))
                for j in range(i, min(len(b) - 3, i + 4 * rng.randrange(1, 64)), 4):
                    b[j:j + 4] = v
            elif kind == 3:
                i = 8 + 8 * rng.randrange(10)
                b[i:i + 8] = struct.pack("<Q", rng.choice([0, 1, 2, 0xffffffff, 0x7fffffff, 1 << 40, rng.randrange(1000)]))
            elif rng.random() < 0.5:
                b = b[:rng.randrange(0, len(b))]
            else:
                b += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 64)))
            files.append(str(tmp_path / f"{os.path.basename(path)}.{k}"))
            with open(files[-1], "wb") as f:
                f.write(bytes(b))
        run = subprocess.run([exe] + files, capture_output=True, text=True, timeout=600)
        out = run.stdout + run.stderr
        assert run.returncode == 0 and "runtime error" not in out and "Sanitizer" not in out, (path, out[-3000:])
        counts = out.split()
        n_ok += int(counts[counts.index("ok") + 1])
        n_refused += int(counts[counts.index("refused") + 1])
        for f in files:
            os.remove(f)
    assert n_ok >= 20 and n_refused >= 20, (n_ok, n_refused)  # both outcomes are exercised


def test_geometry_that_is_not_finite_builds_and_is_never_hit():
    """A sphere whose centre or radius is NaN or infinite is never hit in the reference (every
    comparison of sphere.h:47-60 fails); here it must not upset the builders either: such a
    primitive gets a box that spans everything, the tree is built from finite numbers, and the
    exact test still misses it."""
    arrays = abi.SceneArrays.load(os.path.join(GOLDEN, "bouncing_spheres.rtscene"))
    prims = arrays.prims.copy()
    broken = [7, 100, 211, 400]
    prims["a"][7] = (np.nan, 0.2, 1.0)
    prims["c"][100][0] = np.inf
    prims["a"][211] = (np.inf, -np.inf, 0.0)
    prims["c"][400][0] = np.nan
    arrays.prims = prims
    rng = np.random.default_rng(3)
    r = np.zeros(3000, abi.RAY_DT)
    r["orig"] = rng.uniform(-12, 12, (3000, 3)) * (1, 0, 1) + (0, 2.0, 0)
    r["dir"] = rng.normal(size=(3000, 3))
    r["tm"] = 0.5
    hits, _, _ = S.SimScene(arrays, flags=abi.RT_SCENE_FORCE_BVH).trace(r)
    assert (hits["prim"] >= 0).mean() > 0.4
    assert not np.isin(hits["prim"], broken).any()
    # (the oracle's scan of the primitives; the reference's own tree would lose whole subtrees to
    # the boxes that are not numbers, bvh.h:17-44, which is not something to reproduce)
    ref, _ = O.OracleScene(arrays).trace(r, 0.001, np.inf, O.TRACE_BAKED)
    assert np.array_equal(hits["prim"], ref["prim"]) and np.array_equal(hits["t"], ref["t"])


def test_perlin_tables_are_validated():
    """perm_x indexes randvec (perlin.h:33): an entry outside 0..255 would read past the table on
    the GPU, so the descriptor is refused."""
    arrays = abi.SceneArrays.load(os.path.join(GOLDEN, "perlin_spheres.rtscene"))
    perl = arrays.perlins.copy()
    perl["perm_x"][0][17] = 256
    arrays.perlins = perl
    with pytest.raises(binding.RtError) as e:
        binding.Scene(arrays, device=0)
    assert "perm_x" in str(e.value)


def two_level_scene(seed=21):
    """A world of loose primitives plus sub-scenes placed several times (rt_instance): a cluster of
    spheres and quads (some inside wrappers of their own) under 14 different wrapper chains and as it
    is, and a lone sphere under three.  Returns the scene and the same scene with every instance
    written out as primitives (loose primitives first, then instance by instance), with, per
    written-out primitive, the primitive of the instanced scene it stands for."""
    rng = np.random.default_rng(seed)
    ops, chains = [], []

    def chain(*items):
        first = len(ops)
        ops.extend(items)
        chains.append((first, len(items)))
        return len(chains) - 1

    prims = []

    def sphere(c, r, xform=-1, mat=0):
        p = np.zeros((), abi.PRIM_DT)
        p["type"], p["material"], p["xform"] = abi.RT_PRIM_SPHERE, mat, xform
        p["a"], p["c"] = c, (r, 0, 0)
        prims.append(p)

    def quad(q, u, v, xform=-1, mat=0):
        p = np.zeros((), abi.PRIM_DT)
        p["type"], p["material"], p["xform"] = abi.RT_PRIM_QUAD, mat, xform
        p["a"], p["b"], p["c"] = q, u, v
        prims.append(p)

    sphere((0, -1000, 0), 1000.0, mat=1)                      # loose: ground and a few spheres
    for _ in range(9):
        sphere(rng.uniform(-30, 30, 3) * (1, 0, 1) + (0, 0.5, 0), 0.5, mat=int(rng.integers(0, 4)))
    a_first = len(prims)                                      # sub-scene A: 40 spheres, 8 quads
    for i in range(40):
        xf = chain((abi.RT_XF_SCALE, (1.0, 0.6, 1.3)), (abi.RT_XF_ROTATE_Y, (25.0, 0, 0))) if i % 10 == 3 else -1
        sphere(rng.uniform(-2, 2, 3), float(rng.uniform(0.15, 0.5)), xform=xf, mat=int(rng.integers(0, 4)))
    for i in range(8):
        xf = chain((abi.RT_XF_ROTATE_XYZ, (10.0, 20.0, 30.0)), (abi.RT_XF_TRANSLATE, (0.3, 0.0, -0.2))) if i % 4 == 1 else -1
        quad(rng.uniform(-2, 2, 3), rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3), xform=xf, mat=int(rng.integers(0, 4)))
    a_count = len(prims) - a_first
    sphere((20, 3, 20), 1.0, mat=2)                           # a loose primitive between the sub-scenes
    b_first = len(prims)                                      # sub-scene B: one sphere
    sphere((0, 0, 0), 0.8, mat=3)
    instances = [(a_first, a_count, -1, 0)]
    for k in range(14):
        items = []
        if k % 3 == 0:
            items.append((abi.RT_XF_SCALE, tuple(rng.uniform(0.5, 1.8, 3))))
        if k % 2 == 0:
            items.append((abi.RT_XF_ROTATE_XYZ, tuple(rng.uniform(-80, 80, 3))))
        else:
            items.append((abi.RT_XF_ROTATE_Y, (float(rng.uniform(-180, 180)), 0, 0)))
        items.append((abi.RT_XF_TRANSLATE, tuple(rng.uniform(-25, 25, 3) * (1, 0.2, 1) + (0, 4, 0))))
        instances.append((a_first, a_count, chain(*items), 0))
    for k in range(3):
        instances.append((b_first, 1, chain((abi.RT_XF_SCALE, (1.0 + k, 1.0, 1.0)),
                                            (abi.RT_XF_TRANSLATE, (-10.0 + 10 * k, 8.0, -12.0))), 0))
    mats = np.zeros(4, abi.MATERIAL_DT)
    mats["type"] = (abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_METAL, abi.RT_MAT_DIELECTRIC)
    mats["texture"] = (0, 1, 0, 0)
    mats["albedo"][2] = (0.8, 0.8, 0.9)
    mats["param"] = (0, 0, 0.05, 1.5)
    tex = np.zeros(2, abi.TEXTURE_DT)
    tex["color"] = ((0.7, 0.3, 0.2), (0.5, 0.5, 0.5))
    prims = np.array(prims, abi.PRIM_DT)
    scene = abi.SceneArrays(prims=prims, materials=mats, textures=tex, xform_ops=np.array(ops, abi.XFORM_OP_DT),
                            chains=np.array(chains, abi.CHAIN_DT), instances=np.array(instances, abi.INSTANCE_DT))
    # written out: a primitive of an instance carries its own wrappers, then the instance's (outer)
    in_sub = np.zeros(len(prims), bool)
    for f, n, _, _ in instances:
        in_sub[f:f + n] = True
    flat, origin = [prims[i] for i in np.flatnonzero(~in_sub)], list(np.flatnonzero(~in_sub))
    ops2, chains2 = list(ops), list(chains)
    for f, n, xf, _ in instances:
        for j in range(f, f + n):
            p = prims[j].copy()
            items = []
            if p["xform"] >= 0:
                c0, cn = chains[int(p["xform"])]
                items += ops[c0:c0 + cn]
            if xf >= 0:
                c0, cn = chains[xf]
                items += ops[c0:c0 + cn]
            if items:
                chains2.append((len(ops2), len(items)))
                ops2.extend(items)
                p["xform"] = len(chains2) - 1
            else:
                p["xform"] = -1
            flat.append(p)
            origin.append(j)
    written_out = abi.SceneArrays(prims=np.array(flat, abi.PRIM_DT), materials=mats, textures=tex,
                                  xform_ops=np.array(ops2, abi.XFORM_OP_DT), chains=np.array(chains2, abi.CHAIN_DT))
    return scene, written_out, np.array(origin)


def two_level_rays(n, seed=5):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, abi.RAY_DT)
    o = rng.normal(size=(n, 3))
    o *= rng.uniform(5, 60, size=(n, 1)) / np.linalg.norm(o, axis=1, keepdims=True)
    o[:, 1] = np.abs(o[:, 1]) + 0.5
    tgt = rng.uniform(-28, 28, (n, 3)) * (1, 0.25, 1) + (0, 3, 0)
    rays["orig"], rays["dir"], rays["tm"] = o, (tgt - o) * rng.uniform(0.3, 2.0, (n, 1)), rng.uniform(0, 1, n)
    return rays


def check_two_level_hits(hits, ref, origin):
    """hits of the instanced scene against the oracle's of the written-out one."""
    m = ref["prim"] >= 0
    assert np.array_equal(hits["prim"] >= 0, m)
    assert np.array_equal(hits["prim"][m], origin[ref["prim"][m]]), "another primitive"
    for f in ("t", "p", "normal", "front_face"):
        assert np.array_equal(hits[f][m], ref[f][m]), f


def test_kernel_logic_two_level_scene():
    """rt_instance: the tree over instances and the trees of the sub-scenes, walked by the kernels' own
    traversal code on the host (every stack variant), against the oracle's scan of the same scene with
    every instance written out: the same primitive, t, p, normal and face, bit for bit."""
    scene, written_out, origin = two_level_scene()
    rays = two_level_rays(40_000)
    ref, _ = O.OracleScene(written_out).trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    assert 0.3 < (ref["prim"] >= 0).mean() and len(np.unique(ref["prim"])) > 300
    sim = S.SimScene(scene)
    assert sim.n_nodes < 60  # 48 + 1 sub-scene primitives and 29 items at the top: not 48 x 15 primitives
    for mode in (0, 1, 2):
        S.set_stack_mode(mode)
        try:
            hits, _, _ = sim.trace(rays)
        finally:
            S.set_stack_mode(0)
        check_two_level_hits(hits, ref, origin)


def test_instance_descriptors_are_validated():
    """Ranges outside the primitives, partially overlapping ranges and chains out of range are refused
    before anything is built (no GPU needed: validation comes first)."""
    scene, _, _ = two_level_scene()
    L = binding.lib()
    for edit in ("range", "overlap", "chain"):
        bad = abi.SceneArrays(**{name: getattr(scene, name).copy() for name, _ in abi.SceneArrays.FIELDS})
        if edit == "range":
            bad.instances[0]["n_prims"] = len(bad.prims)
        elif edit == "overlap":
            bad.instances[1]["first_prim"] += 1
        else:
            bad.instances[2]["xform"] = len(bad.chains)
        d = bad.desc()
        out = C.c_void_p()
        assert L.rt_scene_create(C.byref(d), 0, 0, C.byref(out)) == abi.RT_ERR_ARG, edit


def test_scene_file_roundtrip_with_instances(tmp_path):
    """A .rtscene file carries the instance records behind the other arrays (its tenth count, zero in
    files written before there were instances): the C writer, the C reader and the numpy reader
    agree, and a count that does not fit the file is refused."""
    scene, _, _ = two_level_scene()
    p = str(tmp_path / "two_level.rtscene")
    d = scene.desc()
    binding.check(binding.lib().rt_scene_file_save(C.byref(d), p.encode()))
    out = C.POINTER(abi.SceneDesc)()
    binding.check(binding.lib().rt_scene_file_load(p.encode(), C.byref(out)))
    assert out.contents.n_instances == len(scene.instances) == 18 and out.contents.n_prims == len(scene.prims)
    got = np.ctypeslib.as_array(C.cast(out.contents.instances, C.POINTER(C.c_int32)), (18, 4)).copy()
    binding.lib().rt_scene_file_free(out)
    assert np.array_equal(got, scene.instances.view("<i4").reshape(18, 4))
    back = abi.SceneArrays.load(p)
    assert back.instances.tobytes() == scene.instances.tobytes() and back.prims.tobytes() == scene.prims.tobytes()
    q = str(tmp_path / "again.rtscene")
    back.save(q)
    assert open(q, "rb").read() == open(p, "rb").read()
    blob = bytearray(open(p, "rb").read())
    blob[8 + 8 * 9:16 + 8 * 9] = (19).to_bytes(8, "little")  # one instance more than the file holds
    open(q, "wb").write(bytes(blob))
    assert binding.lib().rt_scene_file_load(q.encode(), C.byref(out)) == abi.RT_ERR_IO
