"""The drop-in C++ scene API (raytracing_amd/dropin) and the programs built on it.

CPU part: the scenes that examples/scenes.cpp builds through the reference's API names
(hittable_list, make_shared<sphere|quad>, box, translate/rotate_xyz/scaling, bvh_node, the
materials and textures, camera fields) flatten to byte-identical rt_scene_desc dumps as the
ones the reference harness extracted from the reference's own objects: same primitives in
the same test order (including the leaf order of the reference's bvh_node), same random
stream, same Perlin tables, same camera frame.
GPU part: the programs render through librt_amd.so and write the PNG camera::render promises."""
import os
import subprocess

import numpy as np
import pytest

from raytracing_amd import abi
from tests import oracle_lib as O
from tests.conftest import EXTRA_SCENES, GOLDEN, ROOT, SCENES, load_scene

BIN = os.path.join(ROOT, "examples", "bin")
RT_SCENES = os.path.join(BIN, "rt_scenes")
REF_MAIN = os.path.join(BIN, "ref_main")


def _ensure_built():
    if not os.path.exists(RT_SCENES):
        subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)


def _write_rtimg(path, arrays):
    im = arrays.images[0]
    n = int(im["width"]) * int(im["height"]) * 3
    with open(path, "wb") as f:
        f.write(b"RTIMG1\n%d %d\n" % (im["width"], im["height"]))
        f.write(arrays.texels[int(im["offset"]):int(im["offset"]) + n].tobytes())


@pytest.mark.parametrize("scene", SCENES + EXTRA_SCENES)
def test_scene_programs_flatten_like_the_reference(scene, tmp_path):
    _ensure_built()
    env = dict(os.environ)
    gold = abi.SceneArrays.load(os.path.join(GOLDEN, scene + ".rtscene"))
    if scene == "earth":  # the reference's texture file, decoded by the drop-in's own JPEG decoder
        env["RTW_IMAGES"] = os.path.join(GOLDEN, "images")
    out = str(tmp_path / "dump.rtscene")
    subprocess.run([RT_SCENES, scene, "--dump", out], check=True, env=env, capture_output=True)
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, scene + ".rtscene"), "rb").read()
    assert open(out + ".cam", "rb").read() == open(os.path.join(GOLDEN, scene + ".cam"), "rb").read()


def test_raw_texel_dump_is_accepted_too(tmp_path):
    """A texture may also be handed over as a raw dump of linear texels (<name>.rtimg)."""
    _ensure_built()
    gold = abi.SceneArrays.load(os.path.join(GOLDEN, "earth.rtscene"))
    _write_rtimg(str(tmp_path / "earthmap.rtimg"), gold)
    out = str(tmp_path / "dump.rtscene")
    subprocess.run([RT_SCENES, "earth", "--dump", out], check=True, env=dict(os.environ, RTW_IMAGES=str(tmp_path)),
                   capture_output=True)
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, "earth.rtscene"), "rb").read()


JPEG_DIR = os.path.join(GOLDEN, "jpeg")
JPEG_CASES = sorted(f[:-4] for f in os.listdir(JPEG_DIR) if f.endswith(".jpg"))


@pytest.mark.parametrize("name", JPEG_CASES)
def test_jpeg_decoder_matches_the_reference_texels(name, tmp_path):
    """tests/golden/jpeg/<name>.rtimg holds what the unmodified reference (rtw_image over
    stb_image) makes of <name>.jpg (oracle/make_jpeg_fixtures.py): sampling 4:4:4 / 4:2:2 /
    4:2:0, grayscale, restart intervals, odd sizes, and progressive frames (spectral bands,
    successive approximation, restart intervals).  The drop-in's decoder must give the same
    bytes.  (Vertical-only subsampling is not covered: Pillow cannot write it.)"""
    _ensure_built()
    out = str(tmp_path / "out.rtimg")
    subprocess.run([os.path.join(BIN, "rt_image"), os.path.join(JPEG_DIR, name + ".jpg"), out], check=True,
                   capture_output=True)
    assert open(out, "rb").read() == open(os.path.join(JPEG_DIR, name + ".rtimg"), "rb").read()


def test_arithmetic_coded_jpeg_is_refused_loudly(tmp_path):
    """A frame type the decoder does not implement (here SOF9, arithmetic coding: the frame marker of
    a sequential file rewritten) is an error with a message, not a wrong picture."""
    _ensure_built()
    raw = bytearray(open(os.path.join(JPEG_DIR, "s444.jpg"), "rb").read())
    k = raw.find(b"\xff\xc0")
    assert k > 0
    raw[k + 1] = 0xC9
    src = tmp_path / "arith.jpg"
    src.write_bytes(bytes(raw))
    r = subprocess.run([os.path.join(BIN, "rt_image"), str(src), str(tmp_path / "o.rtimg")], capture_output=True)
    assert r.returncode != 0 and b"arithmetic" in r.stderr


def test_truncated_progressive_jpeg_does_not_crash(tmp_path):
    """Cut anywhere, a progressive file either decodes to a picture of the right size or fails with a
    message (never a crash or a hang)."""
    _ensure_built()
    raw = open(os.path.join(JPEG_DIR, "prog_420_q30.jpg"), "rb").read()
    for cut in range(len(raw) // 7, len(raw), max(1, len(raw) // 23)):
        src = tmp_path / f"cut{cut}.jpg"
        src.write_bytes(raw[:cut])
        r = subprocess.run([os.path.join(BIN, "rt_image"), str(src), str(tmp_path / "o.rtimg")], capture_output=True,
                           timeout=20)
        assert r.returncode in (0, 1), (cut, r.returncode, r.stderr[-200:])


def test_jpeg_header_claiming_a_huge_image_is_refused_before_allocating(tmp_path):
    """Four bytes of the frame header set to 65535 x 65535 in a 2 KB file: tens of GB of planes if
    taken at its word.  Refused with a message (and quickly), not an allocation failure."""
    import time
    _ensure_built()
    raw = bytearray(open(os.path.join(JPEG_DIR, "prog_gray.jpg"), "rb").read())
    i = raw.index(b"\xff\xc2")
    raw[i + 5:i + 9] = b"\xff\xff\xff\xff"
    src = tmp_path / "huge.jpg"
    src.write_bytes(bytes(raw))
    t0 = time.time()
    r = subprocess.run([os.path.join(BIN, "rt_image"), str(src), str(tmp_path / "o.rtimg")], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 1 and "implausible" in r.stderr, r.stderr[-300:]
    assert time.time() - t0 < 10


def test_idct_of_a_damaged_file_wraps_instead_of_overflowing(tmp_path):
    """A damaged file can hold any 16-bit coefficients; the inverse transform then runs modulo
    2^32 (no signed overflow: -fsanitize=undefined stays silent), and blocks a real file can hold
    come out as before (a flat block of DC 8*v gives 128 + v everywhere)."""
    src = tmp_path / "idct_extremes.cpp"
    src.write_text(r"""
#include "rtamd_jpeg.hpp"
#include <cstdio>
int main() {
    using rtamd::jpeg::detail::idct8x8;
    short c[64];
    uint8_t out[64];
    unsigned sum = 0;
    const short ext[4] = {32767, -32768, 32767, -1};
    for (int pat = 0; pat < 6; pat++) {
        for (int i = 0; i < 64; i++) c[i] = pat < 4 ? ext[(i * (pat + 1) + pat) & 3] : (short)((i * 7919 + pat * 104729) * 2654435761u >> 16);
        idct8x8(c, out, 8);
        for (int i = 0; i < 64; i++) sum += out[i];
    }
    for (int v = -128; v < 128; v += 17) {
        for (int i = 0; i < 64; i++) c[i] = 0;
        c[0] = (short)(8 * v);
        idct8x8(c, out, 8);
        for (int i = 0; i < 64; i++)
            if (out[i] != 128 + v) { std::printf("flat block %d: %d\n", v, out[i]); return 1; }
    }
    std::printf("ok %u\n", sum);
    return 0;
}
""")
    exe = str(tmp_path / "idct_extremes")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I" + os.path.join(ROOT, "raytracing_amd", "dropin"), str(src), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.startswith("ok") and "runtime error" not in r.stderr, r.stdout + r.stderr


def test_missing_image_is_not_fatal(tmp_path):
    """rtw_stb_image.h:42 / texture.h:62: a missing texture file prints an error and the
    texture renders blue; the scene still flattens (image = -1)."""
    _ensure_built()
    out = str(tmp_path / "dump.rtscene")
    env = dict(os.environ, RTW_IMAGES=str(tmp_path / "nowhere"))
    r = subprocess.run([RT_SCENES, "earth", "--dump", out], check=True, env=env, capture_output=True, cwd=str(tmp_path))
    assert b"Could not load image file" in r.stderr
    a = abi.SceneArrays.load(out)
    assert len(a.images) == 0 and a.textures[0]["type"] == abi.RT_TEX_IMAGE and a.textures[0]["image"] == -1


def test_reference_main_compiles_against_the_dropin():
    """The reference's own main.cpp, unmodified, builds against the drop-in headers (only where
    the reference sources are mounted; elsewhere the prebuilt binary is what exists)."""
    if not os.path.exists("/root/reference/main.cpp"):
        pytest.skip("reference sources not mounted")
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "-B", "ref_main"], check=True, capture_output=True)
    assert os.path.exists(REF_MAIN)


def test_synth1m_scene_shape(tmp_path):
    """A small cut of the 10^6-cell stress scene: twelve draws per cell, about 3/4 spheres."""
    _ensure_built()
    out = str(tmp_path / "s.rtscene")
    subprocess.run([RT_SCENES, "synth1m", "--half", "20", "--dump", out], check=True, capture_output=True)
    a = abi.SceneArrays.load(out)
    assert len(a.prims) == 40 * 40 + 1
    frac = (a.prims["type"][1:] == abi.RT_PRIM_SPHERE).mean()
    assert 0.70 < frac < 0.80
    cam, frame = abi.load_camera_file(out + ".cam")
    assert (frame.image_width, frame.image_height) == (3840, 2160)


@pytest.mark.parametrize("scene", SCENES + EXTRA_SCENES)
def test_host_side_hit_matches_the_reference_records(scene, tmp_path):
    """hittable::hit on the drop-in classes (sphere, quad, hittable_list, bvh_node, the instance
    wrappers; hittable.h:34-36), called on the host for the rays the reference recorded: hit /
    miss, t, p, normal and face bit for bit, u / v bit for bit as well (glibc's acosf / atan2f on
    both sides).  The primitive id has no counterpart in hit_record.  rotated_in_bvh: the
    reference clips rotated objects inside a bvh_node to their too-small boxes, and so must
    bvh_node::hit of the drop-in."""
    rays_file = os.path.join(GOLDEN, scene + ".rays")
    out = str(tmp_path / "hits.bin")
    env = dict(os.environ, RTW_IMAGES=os.path.join(GOLDEN, "images"))
    subprocess.run([RT_SCENES, scene, "--hits", rays_file, out], check=True, capture_output=True, env=env)
    rays, ref, tmin, tmax = O.load_rays_file(rays_file)
    hits = np.fromfile(out, abi.HIT_DT)
    assert len(hits) == len(ref)
    m = ref["prim"] >= 0
    assert np.array_equal(hits["prim"] >= 0, m)
    for f in ("t", "p", "normal", "front_face", "u", "v"):
        assert np.array_equal(hits[f][m], ref[f][m]), f


def test_shared_objects_become_instances(tmp_path):
    """An object handed to many instance wrappers (examples/scenes.cpp grove: one shrub under 120
    rotate_y / scaling / translate chains) is recorded once and placed by rt_instance records;
    RTAMD_INSTANCES=0 writes every copy out.  The kernels' traversal (compiled for the host) over the
    instanced scene gives, ray for ray, what the oracle's scan of the written-out scene gives: the
    same primitive of the shrub, t, p, normal and face, bit for bit."""
    from tests import sim_lib as S
    _ensure_built()
    inst, flat = str(tmp_path / "grove.rtscene"), str(tmp_path / "grove_flat.rtscene")
    subprocess.run([RT_SCENES, "grove", "--dump", inst], check=True, capture_output=True)
    subprocess.run([RT_SCENES, "grove", "--dump", flat], check=True, capture_output=True,
                   env=dict(os.environ, RTAMD_INSTANCES="0"))
    a, b = abi.SceneArrays.load(inst), abi.SceneArrays.load(flat)
    assert len(a.prims) == 65 and len(a.instances) == 120 and len(b.prims) == 1 + 120 * 64 and len(b.instances) == 0
    assert (a.instances["first_prim"] == 1).all() and (a.instances["n_prims"] == 64).all()
    cam, _ = abi.load_camera_file(inst + ".cam")
    rng = np.random.default_rng(3)
    n = 20_000
    rays = np.zeros(n, abi.RAY_DT)
    rays["orig"] = rng.uniform(-1, 1, (n, 3)) * (20, 4, 20) + (0, 6, 0)
    rays["dir"] = rng.uniform(-18, 18, (n, 3)) * (1, 0.05, 1) + (0, 0.8, 0) - rays["orig"]
    ref, _ = O.OracleScene(b).trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    hits, _, _ = S.SimScene(a).trace(rays)
    m = ref["prim"] >= 0
    assert m.mean() > 0.5 and (ref["prim"][m] > 0).mean() > 0.2  # the shrubs are hit, not only the ground
    origin = np.where(ref["prim"] > 0, 1 + (ref["prim"] - 1) % 64, ref["prim"])
    assert np.array_equal(hits["prim"], origin)
    for f in ("t", "p", "normal", "front_face"):
        assert np.array_equal(hits[f][m], ref[f][m]), f


def test_shared_objects_inside_shared_objects(tmp_path):
    """Sharing on two levels (examples/scenes.cpp orchard: one shrub placed nine times in a tree,
    the one tree placed forty times): the shrub and the tree's own primitives are recorded once
    each, and every shrub of every tree is one rt_instance under the tree's wrappers followed by
    its own.  The kernels' traversal over that scene against the oracle's scan of the scene with
    all 18 001 primitives written out (RTAMD_INSTANCES=0): the same primitive, t, p, normal and
    face, bit for bit."""
    from tests import sim_lib as S
    a, b, origin_of, rays = _orchard(tmp_path)
    ref, _ = O.OracleScene(b).trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    hits, _, _ = S.SimScene(a).trace(rays)
    _check_orchard_hits(hits, ref, origin_of)


def _orchard(tmp_path, n=20_000):
    _ensure_built()
    inst, flat = str(tmp_path / "orchard.rtscene"), str(tmp_path / "orchard_flat.rtscene")
    subprocess.run([RT_SCENES, "orchard", "--dump", inst], check=True, capture_output=True)
    subprocess.run([RT_SCENES, "orchard", "--dump", flat], check=True, capture_output=True,
                   env=dict(os.environ, RTAMD_INSTANCES="0"))
    a, b = abi.SceneArrays.load(inst), abi.SceneArrays.load(flat)
    assert len(a.prims) == 1 + 18 + 48 and len(a.instances) == 40 * (1 + 9)
    assert len(b.prims) == 1 + 40 * (18 + 9 * 48) and len(b.instances) == 0
    n_of = {int(f): int(k) for f, k in zip(a.instances["first_prim"], a.instances["n_prims"])}
    assert n_of == {1: 18, 19: 48}
    # a written-out primitive and the recorded one it is a copy of hold the same numbers, in the
    # object's own space; only the chain differs
    def key(p):
        return (int(p["type"]), int(p["material"]), p["a"].tobytes(), p["b"].tobytes(), p["c"].tobytes(), p["d"].tobytes())
    recorded = {key(p): i for i, p in enumerate(a.prims)}
    assert len(recorded) == len(a.prims)
    origin_of = np.array([recorded[key(p)] for p in b.prims])
    rng = np.random.default_rng(4)
    rays = np.zeros(n, abi.RAY_DT)
    rays["orig"] = rng.uniform(-1, 1, (n, 3)) * (24, 5, 16) + (0, 7, -2)
    rays["dir"] = rng.uniform(-1, 1, (n, 3)) * (22, 2.2, 14) + (0, 2.2, -4) - rays["orig"]
    return a, b, origin_of, rays


def _check_orchard_hits(hits, ref, origin_of):
    m = ref["prim"] >= 0
    in_shrub = origin_of[ref["prim"][m]] >= 19
    assert m.mean() > 0.5 and in_shrub.mean() > 0.15 and ((origin_of[ref["prim"][m]] >= 1) & ~in_shrub).mean() > 0.01
    assert np.array_equal(hits["prim"] >= 0, m)
    assert np.array_equal(hits["prim"][m], origin_of[ref["prim"][m]])
    for f in ("t", "p", "normal", "front_face"):
        assert np.array_equal(hits[f][m], ref[f][m]), f


def test_host_side_materials_and_textures(tmp_path):
    """material::scatter / emitted and texture::value on the host (material.h:13-20, texture.h:13):
    tests/host_virtuals.cpp checks them against closed forms and against the oracle-independent
    invariants of the reference's formulas; here it only has to build and pass."""
    exe = str(tmp_path / "host_virtuals")
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I" + os.path.join(ROOT, "raytracing_amd", "dropin"),
                    "-I" + os.path.join(ROOT, "raytracing_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "host_virtuals.cpp"), "-o", exe,
                    "-L" + os.path.join(ROOT, "raytracing_amd", "lib"), "-lrt_amd",
                    "-Wl,-rpath," + os.path.join(ROOT, "raytracing_amd", "lib")], check=True)
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()


def test_image_decoder_survives_damaged_files_under_sanitizers(tmp_path):
    """The drop-in's image reader (its own JPEG decoder, PPM and raw dumps) takes files from
    users: truncated, bit-flipped and spliced copies of the JPEG fixtures must end in a decoded
    image or a refusal, never in a report of -fsanitize=address,undefined."""
    import glob
    import random
    exe = str(tmp_path / "rt_image_sanitized")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-I" + os.path.join(ROOT, "raytracing_amd", "dropin"), "-I" + os.path.join(ROOT, "raytracing_amd", "csrc"),
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "image_dump.cpp"), "-o", exe,
                    "-L" + os.path.join(ROOT, "raytracing_amd", "lib"), "-lrt_amd",
                    "-Wl,-rpath," + os.path.join(ROOT, "raytracing_amd", "lib")], check=True)
    rng = random.Random(5)
    out = str(tmp_path / "out.rtimg")
    bad = str(tmp_path / "damaged.jpg")
    runs = 0
    for path in sorted(glob.glob(os.path.join(GOLDEN, "jpeg", "*.jpg"))):
        data = open(path, "rb").read()
        for k in range(10):
            b = bytearray(data)
            if k % 4 == 0:
                b = b[:rng.randrange(2, len(b))]
            elif k % 4 == 1:
                for _ in range(rng.randrange(1, 6)):
                    b[rng.randrange(0, min(len(b), 700))] = rng.randrange(256)
            elif k % 4 == 2:
                for _ in range(rng.randrange(1, 30)):
                    b[rng.randrange(0, len(b))] = rng.randrange(256)
            else:
                i, j, n = rng.randrange(0, len(b)), rng.randrange(0, len(b)), rng.randrange(1, 200)
                b[i:i + n] = b[j:j + n]
            with open(bad, "wb") as f:
                f.write(bytes(b))
            r = subprocess.run([exe, bad, out], capture_output=True, text=True, timeout=120)
            text = r.stdout + r.stderr
            assert r.returncode >= 0 and "runtime error" not in text and "AddressSanitizer" not in text, (path, k, text[-2000:])
            runs += 1
    assert runs >= 100


@pytest.mark.gpu
def test_scene_program_renders_png(tmp_path):
    from PIL import Image
    from raytracing_amd import binding
    png = str(tmp_path / "quads.png")
    r = subprocess.run([RT_SCENES, "quads", "--spp", "32", "--out", png], check=True, capture_output=True)
    assert b"Render took" in r.stderr
    img = np.asarray(Image.open(png).convert("RGB"))
    arrays, cam, _ = load_scene("quads")
    sc = binding.Scene(arrays)
    acc, _ = sc.render(cam, 32, seed=0)
    assert np.array_equal(img, binding.tonemap_u8(acc, 1.0 / 32))
    sc.close()


@pytest.mark.gpu
def test_shared_objects_render_the_same_as_instances_and_written_out(tmp_path):
    """The grove scene through camera::render of the drop-in, once with its shrub recorded as a
    sub-scene with 120 instances (the default) and once with every copy written out
    (RTAMD_INSTANCES=0): the same picture (a hit that rounding alone decides may fall either way
    with the order boxes are visited in, which differs between the two trees: a handful of pixels)."""
    from PIL import Image
    imgs = []
    for env in ({}, {"RTAMD_INSTANCES": "0"}):
        png = str(tmp_path / f"grove{len(imgs)}.png")
        r = subprocess.run([RT_SCENES, "grove", "--spp", "8", "--width", "400", "--out", png], check=True,
                           capture_output=True, env=dict(os.environ, **env))
        assert b"Render took" in r.stderr
        imgs.append(np.asarray(Image.open(png).convert("RGB")).astype(int))
    assert imgs[0].std() > 10
    assert (imgs[0] != imgs[1]).any(axis=2).mean() < 1e-3


@pytest.mark.gpu
def test_shared_objects_inside_shared_objects_on_the_gpu(tmp_path):
    """The orchard scene (a shared shrub inside a shared tree, 400 instances of 67 primitives) on
    the GPU: both batch kernels against the oracle's scan of the written-out scene, bit for bit, and
    camera::render of the drop-in against the render with every copy written out."""
    from PIL import Image
    from raytracing_amd import binding
    a, b, origin_of, rays = _orchard(tmp_path, 60_000)
    ref, _ = O.OracleScene(b).trace(rays, 0.001, np.inf, O.TRACE_REFERENCE)
    sc = binding.Scene(a)
    for sched in (False, True):
        _check_orchard_hits(sc.trace(rays, sched=sched), ref, origin_of)
    sc.close()
    imgs = []
    for env in ({}, {"RTAMD_INSTANCES": "0"}):
        png = str(tmp_path / f"orchard{len(imgs)}.png")
        subprocess.run([RT_SCENES, "orchard", "--spp", "8", "--width", "400", "--out", png], check=True,
                       capture_output=True, env=dict(os.environ, **env))
        imgs.append(np.asarray(Image.open(png).convert("RGB")).astype(int))
    assert imgs[0].std() > 10
    assert (imgs[0] != imgs[1]).any(axis=2).mean() < 1e-3


@pytest.mark.gpu
def test_unmodified_reference_main_renders_on_the_gpu(tmp_path):
    """bin/ref_main is the reference's main.cpp (literal switch (7): cornell_box, 600x600, 10 spp,
    written to ../RayTracing.png) compiled against the drop-in.  Its image must have the mean
    of the render the reference ships (RayTracing.png: 32.34, 26.34, 22.26 on 8-bit pixels)."""
    if not os.path.exists(REF_MAIN):
        pytest.skip("bin/ref_main was not built (reference sources were not mounted)")
    from PIL import Image
    work = tmp_path / "build"
    work.mkdir()
    subprocess.run([REF_MAIN], check=True, cwd=str(work), capture_output=True)
    img = np.asarray(Image.open(str(tmp_path / "RayTracing.png")).convert("RGB"))
    assert img.shape == (600, 600, 3)
    assert np.allclose(img.reshape(-1, 3).mean(0), (32.34, 26.34, 22.26), atol=1.0)
