import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the reference compiled in the build container)")


@pytest.fixture(scope="session")
def monkey_sd():
    from raytracer_amd import scene
    return scene.parse_scene_file(os.path.join(ROOT, "tests", "golden", "monkey.txt"))
