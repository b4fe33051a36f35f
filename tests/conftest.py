import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = ["bouncing_spheres", "checkered_spheres", "earth", "perlin_spheres", "quads",
          "simple_light", "cornell_box"]
# Scenes of the reference harness itself (not in the reference's main.cpp), with a scene dump, a camera
# and ray records only.  rotated_in_bvh: rotated objects inside a bvh_node, which the reference clips
# to the too-small boxes of its rotate_y / rotate_xyz constructors (hittable.h:123-126, 241-244).
# synth60_list / synth60_bvh: the synthetic grid scene of the 4K configuration at 14 401 primitives, built
# from the reference's classes as a plain hittable_list and under one bvh_node.
EXTRA_SCENES = ["rotated_in_bvh", "synth60_list", "synth60_bvh"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_initialises_the_gpu_first():
    """PyTorch's wheel carries its own copy of the HIP runtime.  When librt_amd.so (linked
    against the system's) has been the first to touch the GPU in a process, torch's later
    initialisation reports "No HIP GPUs are available"; the other order works (it is the order
    of bench.py and of any torch program that calls into the library).  So on a GPU box torch
    goes first."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_scene(name):
    from raytracing_amd import abi
    arrays = abi.SceneArrays.load(os.path.join(GOLDEN, name + ".rtscene"))
    cam, frame = abi.load_camera_file(os.path.join(GOLDEN, name + ".cam"))
    return arrays, cam, frame
