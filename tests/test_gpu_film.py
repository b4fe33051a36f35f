"""Film (rt_film_*): device-resident accumulation in passes, asynchronous snapshots, checkpoint /
resume.  The reference renders in one go and writes once (camera.h:39-57); the checks here are
against this library's own one-call entry points, which the other GPU tests pin to the oracle."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import load_scene

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT_SCENES = os.path.join(ROOT, "examples", "bin", "rt_scenes")


def _passes(scene, cam, film, begin, n_passes, spp):
    for k in range(n_passes):
        film.render(scene, cam, spp, begin + k * spp, seed=3)


def test_passes_equal_separate_render_calls_and_snapshot_is_write_color():
    from raytracing_amd import binding
    arrays, cam, _ = load_scene("bouncing_spheres")
    sc = binding.Scene(arrays)
    H, W = cam.image_height(), cam.image_width
    film = binding.Film(W, H)
    _passes(sc, cam, film, 0, 4, 4)
    film.snapshot(with_accum=True)
    _passes(sc, cam, film, 16, 1, 4)  # a pass enqueued behind the snapshot must not leak into it
    rgb, acc, n = film.wait(want_accum=True)
    assert n == 16
    ref = np.zeros((H, W, 3), np.float32)
    for k in range(4):
        ref, _ = sc.render(cam, 4, sample_begin=4 * k, seed=3, accum=ref)
    assert np.array_equal(acc, ref)
    assert np.array_equal(rgb, binding.tonemap_u8(ref, 1.0 / 16))
    film.snapshot(with_accum=True)
    _, acc2, n2 = film.wait(want_accum=True)
    ref, _ = sc.render(cam, 4, sample_begin=16, seed=3, accum=ref)
    assert n2 == 20 and np.array_equal(acc2, ref)
    film.close()
    sc.close()


def test_resume_from_uploaded_sums_is_bit_identical():
    from raytracing_amd import binding
    arrays, cam, _ = load_scene("simple_light")
    sc = binding.Scene(arrays)
    H, W = cam.image_height(), cam.image_width
    straight = binding.Film(W, H)
    _passes(sc, cam, straight, 0, 4, 8)
    straight.snapshot(with_accum=True)
    rgb_a, acc_a, _ = straight.wait(want_accum=True)
    first = binding.Film(W, H)
    _passes(sc, cam, first, 0, 2, 8)
    first.snapshot(with_accum=True)
    _, half, n = first.wait(want_accum=True)
    first.close()
    second = binding.Film(W, H)
    second.upload(half, n)
    _passes(sc, cam, second, n, 2, 8)
    second.snapshot(with_accum=True)
    rgb_b, acc_b, n_b = second.wait(want_accum=True)
    assert n_b == 32 and np.array_equal(acc_a, acc_b) and np.array_equal(rgb_a, rgb_b)
    second.upload(None, 0)
    with pytest.raises(binding.RtError):
        second.snapshot()  # nothing accumulated
    for f in (straight, second):
        f.close()
    sc.close()


def test_film_argument_errors():
    from raytracing_amd import binding
    arrays, cam, _ = load_scene("quads")
    sc = binding.Scene(arrays)
    with pytest.raises(binding.RtError):
        binding.Film(0, 10)
    film = binding.Film(cam.image_width + 1, cam.image_height())
    with pytest.raises(binding.RtError, match="film is"):
        film.render(sc, cam, 1, 0)
    with pytest.raises(binding.RtError):
        film.wait()  # no snapshot yet
    film.close()
    sc.close()


def test_scene_program_renders_in_passes_and_resumes(tmp_path):
    """The drop-in camera with RTAMD_PASS_SPP: same image as the library's passes, the PNG is
    rewritten as passes finish, and a second run continues from the checkpoint file."""
    from PIL import Image
    from raytracing_amd import binding
    png, ckpt = str(tmp_path / "q.png"), str(tmp_path / "q.rtaccum")
    env = dict(os.environ, RTAMD_PASS_SPP="8", RTAMD_WRITE_EVERY="2", RTAMD_CHECKPOINT=ckpt)
    r = subprocess.run([RT_SCENES, "quads", "--spp", "24", "--out", png], check=True, capture_output=True, env=env)
    assert b"Samples per pixel done: 16 of 24" in r.stderr and b"Samples per pixel done: 24 of 24" in r.stderr
    arrays, cam, _ = load_scene("quads")
    sc = binding.Scene(arrays)
    film = binding.Film(cam.image_width, cam.image_height())
    for k in range(3):
        film.render(sc, cam, 8, 8 * k, seed=0)
    film.snapshot(with_accum=True)
    rgb, acc, _ = film.wait(want_accum=True)
    assert np.array_equal(np.asarray(Image.open(png).convert("RGB")), rgb)
    raw = open(ckpt, "rb").read()
    assert raw[:8] == b"RTACCUM1" and int(np.frombuffer(raw[16:24], np.int64)[0]) == 24
    assert np.array_equal(np.frombuffer(raw[32:], np.float32).reshape(acc.shape), acc)
    # a longer render of the same image picks the 24 samples up and adds 16
    r = subprocess.run([RT_SCENES, "quads", "--spp", "40", "--out", png], check=True, capture_output=True, env=env)
    assert b"Resuming at sample 24 of 40" in r.stderr
    for k in range(3, 5):
        film.render(sc, cam, 8, 8 * k, seed=0)
    film.snapshot()
    rgb, _, n = film.wait()
    assert n == 40 and np.array_equal(np.asarray(Image.open(png).convert("RGB")), rgb)
    film.close()
    sc.close()
