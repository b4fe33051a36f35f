"""GPU parity of the closest-hit entry points (rt_trace_batch and rt_trace_batch_sched through
the C ABI; the latter answers the query with the render kernel's own traversal: its launch shape,
LDS copy of the scene, stacks and interleaved scheduling of node steps and parked leaves).

Bar: hit / miss and primitive ids bit-exact against the reference's own records
(tests/golden/*.rays, written by the unmodified reference); t, p and the normal
bit-exact as well (same fp32 operation order); u, v within 2e-6 (device acosf /
atan2f differ from glibc's by an ulp or two)."""
import os

import numpy as np
import pytest

from raytracing_amd import abi
from tests import oracle_lib as O
from tests.conftest import GOLDEN, SCENES, load_scene

pytestmark = pytest.mark.gpu


def _binding():
    from raytracing_amd import binding
    return binding


@pytest.mark.parametrize("scene", SCENES)
def test_trace_matches_reference_records(scene):
    B = _binding()
    arrays, cam, _ = load_scene(scene)
    rays, ref_hits, tmin, tmax = O.load_rays_file(os.path.join(GOLDEN, scene + ".rays"))
    sc = B.Scene(arrays)
    for sched in (False, True):
        hits = sc.trace(rays, tmin, tmax, sched=sched)
        _check_records(hits, ref_hits)
    _check_ids(sc.closest_hits(rays, tmin, tmax), ref_hits)
    sc.close()


def _check_ids(prim_t, ref_hits):
    """rt_closest_hits against the reference's records: hit / miss, primitive and t, bit for bit."""
    prim, t = prim_t
    assert np.array_equal(prim, ref_hits["prim"]), "hit/miss or primitive id differs"
    m = ref_hits["prim"] >= 0
    assert np.array_equal(t[m], ref_hits["t"][m]), "t is not bit-exact"


def _check_records(hits, ref_hits):
    assert np.array_equal(hits["prim"], ref_hits["prim"]), "hit/miss or primitive id differs"
    m = ref_hits["prim"] >= 0
    assert np.array_equal(hits["front_face"][m], ref_hits["front_face"][m])
    assert np.array_equal(hits["t"][m], ref_hits["t"][m]), "t is not bit-exact"
    assert np.array_equal(hits["p"][m], ref_hits["p"][m])
    assert np.array_equal(hits["normal"][m], ref_hits["normal"][m])
    assert np.abs(hits["u"][m] - ref_hits["u"][m]).max(initial=0) <= 2e-6
    assert np.abs(hits["v"][m] - ref_hits["v"][m]).max(initial=0) <= 2e-6


@pytest.mark.parametrize("scene", SCENES)
def test_reference_records_through_the_render_kernels_traversal(scene, monkeypatch):
    """The reference's own ray records through the traversal the benchmarked kernel runs
    (rt_trace_batch_sched), with a tree forced over every scene: the variant that keeps the scene
    in LDS (16-bit stacks), the one that reads it from HBM (32-bit stacks, top of the tree in
    LDS), and the latter with only four stack entries in LDS so that stacks overflow into global
    memory.  Bit-exact: hit / miss, primitive, t, p, normal, face."""
    B = _binding()
    arrays, cam, _ = load_scene(scene)
    rays, ref_hits, tmin, tmax = O.load_rays_file(os.path.join(GOLDEN, scene + ".rays"))
    for flags, lds_stack in ((abi.RT_SCENE_FORCE_BVH, None), (abi.RT_SCENE_FORCE_BVH | abi.RT_SCENE_FORCE_GLOBAL, None),
                             (abi.RT_SCENE_FORCE_BVH | abi.RT_SCENE_FORCE_GLOBAL, "4")):
        if lds_stack:
            monkeypatch.setenv("RT_AMD_LDS_STACK", lds_stack)  # read by rt_scene_create
        sc = B.Scene(arrays, flags=flags)
        if lds_stack:
            monkeypatch.delenv("RT_AMD_LDS_STACK")
        info = sc.info()
        if len(arrays.prims) >= 2:
            assert info.n_nodes > 0 and info.lds_resident == (0 if flags & abi.RT_SCENE_FORCE_GLOBAL else 1)
        _check_records(sc.trace(rays, tmin, tmax, sched=True), ref_hits)
        _check_ids(sc.closest_hits(rays, tmin, tmax), ref_hits)
        sc.close()


@pytest.mark.parametrize("scene", ["bouncing_spheres", "quads", "simple_light", "checkered_spheres", "cornell_box"])
def test_forced_bvh_and_forced_linear_agree_with_oracle(scene):
    """Both traversals must give the oracle's closest hit on the same rays."""
    B = _binding()
    arrays, cam, _ = load_scene(scene)
    rays, ref_hits, tmin, tmax = O.load_rays_file(os.path.join(GOLDEN, scene + ".rays"))
    for flags in (abi.RT_SCENE_FORCE_BVH, abi.RT_SCENE_FORCE_LINEAR,
                  abi.RT_SCENE_FORCE_BVH | abi.RT_SCENE_FORCE_GLOBAL):
        sc = B.Scene(arrays, flags=flags)
        for sched in (False, True):
            hits = sc.trace(rays, tmin, tmax, sched=sched)
            assert np.array_equal(hits["prim"], ref_hits["prim"])
            m = ref_hits["prim"] >= 0
            assert np.array_equal(hits["t"][m], ref_hits["t"][m])
        sc.close()


def test_instances_inside_a_bvh():
    """Rotated / scaled / translated spheres and quads in a scene large enough for a BVH: the
    wrappers are applied to the ray at the leaves (hittable.h:51-344), bit-exact with the
    oracle's emulation of the reference."""
    from tests.test_abi_and_host import instanced_rays, instanced_scene
    B = _binding()
    arrays = instanced_scene()
    rays = instanced_rays(200_000)
    ref, _ = O.OracleScene(arrays).trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    for flags in (0, abi.RT_SCENE_FORCE_BVH | abi.RT_SCENE_FORCE_GLOBAL, abi.RT_SCENE_FORCE_LINEAR):
        sc = B.Scene(arrays, flags=flags)
        assert sc.trace(rays, sched=True).tobytes() == sc.trace(rays).tobytes()
        hits = sc.trace(rays)
        assert np.array_equal(hits["prim"], ref["prim"])
        m = ref["prim"] >= 0
        for f in ("t", "p", "normal", "front_face"):
            assert np.array_equal(hits[f][m], ref[f][m]), f
        du = np.abs(hits["u"][m] - ref["u"][m])
        du = np.minimum(du, 1.0 - du)  # atan2 at the seam: u = 0 and u = 1 are the same meridian
        dv = np.abs(hits["v"][m] - ref["v"][m])
        # acosf(-y) of an outward normal that rounding left a hair outside the unit sphere is
        # NaN in the reference too (sphere.h:105): NaN must then meet NaN
        assert np.array_equal(np.isnan(hits["v"][m]), np.isnan(ref["v"][m]))
        assert np.nanmax(du) <= 2e-6 and np.nanmax(dv) <= 2e-6
        sc.close()
    # and a render of it: megakernel == wavefront, bit for bit
    cam = abi.CameraDesc(1.0, 96, 4, 8, (0.6, 0.7, 0.9), 50.0, (0, 5, 32), (0, 0, 0), (0, 1, 0), 0.0, 10.0)
    sc = B.Scene(arrays)
    a, st = sc.render(cam, 4, seed=1)
    w, sw = sc.render(cam, 4, seed=1, mode=abi.RT_MODE_WAVEFRONT)
    assert st.rays == sw.rays and np.array_equal(a, w) and a.std() > 0
    sc.close()


def test_rotated_objects_inside_a_bvh_node_are_whole_on_the_gpu():
    """A documented difference: the reference bounds rotate_y / rotate_xyz objects too tightly
    (hittable.h:123-126, 241-244) and clips them when they sit inside a bvh_node (bvh.h:78); this
    library bounds the flattened primitives itself.  On the reference's own rays of such a scene
    (tests/golden/rotated_in_bvh.rays) the GPU must give what a scan of every primitive through
    its wrappers gives (the oracle without the bvh_node group), bit for bit; it must agree with
    the reference's records wherever they agree with that scan; and where they do not, the GPU's
    hit is the nearer one and lies on a rotated primitive."""
    B = _binding()
    arrays, cam, _ = load_scene("rotated_in_bvh")
    rays, ref_hits, tmin, tmax = O.load_rays_file(os.path.join(GOLDEN, "rotated_in_bvh.rays"))
    flat = abi.SceneArrays.load(os.path.join(GOLDEN, "rotated_in_bvh.rtscene"))
    flat.groups = np.array([(abi.RT_GROUP_LIST, 0, len(flat.prims), 0)], abi.GROUP_DT)
    scan, _ = O.OracleScene(flat).trace(rays, tmin, tmax, O.TRACE_REFERENCE)
    rotated = np.zeros(len(arrays.prims), bool)  # primitives with a rotation among their wrappers
    for i, pr in enumerate(arrays.prims):
        c = int(pr["xform"])
        if c >= 0:
            ch = arrays.chains[c]
            ops = arrays.xform_ops[int(ch["first_op"]):int(ch["first_op"]) + int(ch["n_ops"])]
            rotated[i] = np.isin(ops["type"], (abi.RT_XF_ROTATE_Y, abi.RT_XF_ROTATE_XYZ)).any()
    assert rotated.sum() == 13  # two boxes of six quads and one sphere
    for flags in (0, abi.RT_SCENE_FORCE_GLOBAL, abi.RT_SCENE_FORCE_LINEAR):
        sc = B.Scene(arrays, flags=flags)
        for sched in (False, True):
            hits = sc.trace(rays, tmin, tmax, sched=sched)
            _check_records(hits, scan)
        sc.close()
    same = (scan["prim"] == ref_hits["prim"]) & ((scan["t"] == ref_hits["t"]) | (scan["prim"] < 0))
    diff = ~same
    assert 50 < diff.sum() < len(rays) // 4           # the reference does clip in this scene
    assert (scan["prim"][diff] >= 0).all() and rotated[scan["prim"][diff]].all()
    behind = ref_hits["prim"][diff] >= 0
    assert (scan["t"][diff][behind] <= ref_hits["t"][diff][behind]).all()


def test_large_random_batch_bouncing_spheres():
    """1M rays (far origins, grazing directions): BVH culling must never drop a hit that the
    oracle's scan of every primitive finds."""
    B = _binding()
    arrays, cam, _ = load_scene("bouncing_spheres")
    rng = np.random.default_rng(7)
    n = 1_000_000
    rays = np.zeros(n, abi.RAY_DT)
    o = rng.normal(size=(n, 3))
    o /= np.linalg.norm(o, axis=1, keepdims=True)
    o *= rng.uniform(5, 60, size=(n, 1))
    o[:, 1] = np.abs(o[:, 1]) * 0.3 + 0.05
    tgt = np.stack([rng.uniform(-11, 11, n), rng.uniform(0, 1.0, n), rng.uniform(-11, 11, n)], 1)
    rays["orig"] = o
    rays["dir"] = (tgt - o) * rng.uniform(0.2, 3, size=(n, 1))
    rays["tm"] = rng.uniform(0, 1, n)
    osc = O.OracleScene(arrays)
    oh, _ = osc.trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    for flags in (0, abi.RT_SCENE_FORCE_GLOBAL):  # the render kernel's LDS and HBM variants
        sc = B.Scene(arrays, flags=flags)
        for sched in (False, True):
            hits = sc.trace(rays, sched=sched)
            assert np.array_equal(hits["prim"], oh["prim"]), (flags, sched)
            assert np.array_equal(hits["t"], oh["t"]), (flags, sched)
        sc.close()


def test_edge_cases():
    """Empty batch, a ray with a zero direction component, tmax cutting the hit off, and the
    open / closed interval rules at t == tmax (sphere.h:58 vs quad.h:46)."""
    B = _binding()
    arrays, cam, _ = load_scene("quads")
    sc = B.Scene(arrays)
    assert len(sc.trace(np.zeros(0, abi.RAY_DT))) == 0
    rays = np.zeros(3, abi.RAY_DT)
    rays["orig"] = [(0, 0, 9), (0, 0, 9), (0, 0, 9)]
    rays["dir"] = [(0, 0, -1), (0, 0, -1), (0, 0, -1)]
    osc = O.OracleScene(arrays)
    for tmax in (np.inf, 9.0, 8.999):  # the back quad sits at z = 0: t = 9 exactly
        hits = sc.trace(rays, 0.001, tmax)
        oh, _ = osc.trace(rays, 0.001, tmax, O.TRACE_REFERENCE)
        assert np.array_equal(hits["prim"], oh["prim"])
        assert np.array_equal(hits["t"], oh["t"])
    hits = sc.trace(rays, 0.001, 9.0)
    assert (hits["prim"] >= 0).all()      # closed interval: t == tmax is a hit for a quad
    hits = sc.trace(rays, 0.001, 8.999)
    assert (hits["prim"] == -1).all()
    sc.close()
    # sphere: open interval, t == tmax is a miss
    arrays, cam, _ = load_scene("earth")
    sc = B.Scene(arrays)
    rays = np.zeros(1, abi.RAY_DT)
    rays["orig"] = [(0, 0, 12)]
    rays["dir"] = [(0, 0, -1)]
    assert sc.trace(rays, 0.001, np.inf)["t"][0] == 10.0
    assert sc.trace(rays, 0.001, 10.0)["prim"][0] == -1
    sc.close()


def test_rays_that_are_not_numbers_end():
    """NaN / infinite / zero / huge rays: every traversal ends (a NaN passes every box test, so
    such a ray walks the whole tree and fails every exact test) and reports what the oracle's
    scan reports; on the scene held in LDS, read from HBM, with a short LDS stack, through both
    batch kernels."""
    B = _binding()
    arrays, cam, _ = load_scene("bouncing_spheres")
    nan, inf = np.nan, np.inf
    cases = [((nan, 0, 0), (0, 0, -1)), ((0, 1, 5), (nan, 0, -1)), ((0, 1, 5), (nan, nan, nan)),
             ((inf, 1, 5), (0, 0, -1)), ((0, 1, 5), (0, inf, -1)), ((0, 1, 5), (0, 0, 0)),
             ((0, 1, 5), (0, 0, -1e38)), ((0, 1, 5), (1e-42, 0, -1e-42)), ((0, 1, 5), (0, 0, -1)),
             ((-inf, inf, nan), (inf, -inf, nan)), ((3e38, 3e38, 3e38), (-1, -1, -1))]
    rays = np.zeros(64 * len(cases), abi.RAY_DT)  # whole waves of each, so that none hides behind a neighbour
    for i, (o, d) in enumerate(cases):
        rays["orig"][i::len(cases)] = o
        rays["dir"][i::len(cases)] = d
    osc = O.OracleScene(arrays)
    with np.errstate(all="ignore"):
        oh, _ = osc.trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    assert (oh["prim"][:len(cases)][[0, 1, 2]] == -1).all() and oh["prim"][8] >= 0
    for flags, env in ((0, {}), (abi.RT_SCENE_FORCE_GLOBAL, {}), (abi.RT_SCENE_FORCE_GLOBAL, {"RT_AMD_LDS_STACK": "4"})):
        for k, v in env.items():
            os.environ[k] = v
        try:
            sc = B.Scene(arrays, flags=flags)
        finally:
            for k in env:
                del os.environ[k]
        for sched in (False, True):
            hits = sc.trace(rays, sched=sched)
            assert np.array_equal(hits["prim"], oh["prim"]), (flags, env, sched)
            same_t = (hits["t"] == oh["t"]) | (hits["prim"] < 0)
            assert same_t.all(), (flags, env, sched)
        sc.close()


def test_axis_aligned_quads_in_a_scan():
    """The scan's shortened test for quads along the coordinate axes (every orientation, both
    orders and signs of u and v, some inside wrappers) next to general quads and spheres:
    records equal to the oracle's (t, p, normal, and a quad's u and v bit for bit), through both
    batch kernels."""
    from tests.test_abi_and_host import axis_aligned_quads_scene, axis_aligned_rays
    B = _binding()
    arrays = axis_aligned_quads_scene()
    rays = axis_aligned_rays(200_000)
    ref, _ = O.OracleScene(arrays).trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    sc = B.Scene(arrays)
    assert sc.info().n_nodes == 0  # scanned, not a tree
    for sched in (False, True):
        hits = sc.trace(rays, sched=sched)
        _check_records(hits, ref)
        quads = (ref["prim"] >= 0) & (arrays.prims["type"][np.maximum(ref["prim"], 0)] == abi.RT_PRIM_QUAD)
        assert quads.sum() > 10_000  # (u, v of a quad are alpha and beta: exact; a sphere's go through acos / atan2)
        assert np.array_equal(hits["u"][quads], ref["u"][quads]) and np.array_equal(hits["v"][quads], ref["v"][quads])
    sc.close()


def test_geometry_that_is_not_finite_is_never_hit_on_the_gpu():
    """Spheres whose centre or radius is NaN or infinite (never hit in the reference: every
    comparison of sphere.h:47-60 fails) in a scene that is otherwise bouncing_spheres: the tree
    is built from finite boxes, the kernels run to their end, and both closest-hit entry points
    and a short render agree with the oracle's scan of the primitives."""
    import torch
    B = _binding()
    arrays, cam, _ = load_scene("bouncing_spheres")
    prims = arrays.prims.copy()
    broken = [7, 100, 211, 400]
    prims["a"][7] = (np.nan, 0.2, 1.0)
    prims["c"][100][0] = np.inf
    prims["a"][211] = (np.inf, -np.inf, 0.0)
    prims["c"][400][0] = np.nan
    arrays.prims = prims
    rng = np.random.default_rng(3)
    r = np.zeros(20_000, abi.RAY_DT)
    r["orig"] = rng.uniform(-12, 12, (len(r), 3)) * (1, 0, 1) + (0, 2.0, 0)
    r["dir"] = rng.normal(size=(len(r), 3))
    r["tm"] = 0.5
    ref, _ = O.OracleScene(arrays).trace(r, 0.001, np.inf, O.TRACE_BAKED)
    assert (ref["prim"] >= 0).mean() > 0.4 and not np.isin(ref["prim"], broken).any()
    for flags in (0, abi.RT_SCENE_FORCE_GLOBAL):
        sc = B.Scene(arrays, flags=flags)
        for sched in (False, True):
            hits = sc.trace(r, 0.001, np.inf, sched=sched)
            assert np.array_equal(hits["prim"], ref["prim"]) and np.array_equal(hits["t"], ref["t"])
        H, W = cam.image_height(), cam.image_width
        accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        sc.render_device(cam, 4, accum.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, seed=1)
        torch.cuda.synchronize()
        assert torch.isfinite(accum).all() and float(accum.mean()) > 0
        sc.close()
