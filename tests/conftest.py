import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = ["bouncing_spheres", "checkered_spheres", "earth", "perlin_spheres", "quads",
          "simple_light", "cornell_box"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_scene(name):
    from raytracing_amd import abi
    arrays = abi.SceneArrays.load(os.path.join(GOLDEN, name + ".rtscene"))
    cam, frame = abi.load_camera_file(os.path.join(GOLDEN, name + ".cam"))
    return arrays, cam, frame
