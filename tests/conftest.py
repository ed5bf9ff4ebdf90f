import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def manifest():
    import json
    with open(os.path.join(GOLDEN, "manifest.json")) as handle:
        return json.load(handle)


# Every GPU test runs once per kernel path, three DIFFERENT kernels for the small robots most tests use: "auto" (the engine's own
# choice: robots of up to 512 voxels go to the wide kernel, kernels_wide.hpp; larger ones to the resident kernel, lattices above
# 1024 voxels to the tiled one), "narrow" (wide = 0, tiled = 0: the resident kernel k_robot_steps with its three bond rounds /
# the streaming kernels: the round-1 paths) and "tiles3" (every robot cut into about three tiles whatever its size:
# k_tile_steps).  A test that sets the options itself overrides this.
KERNEL_PATHS = {"auto": "", "narrow": "tiled=0,wide=0", "tiles3": "tiled=2,tiles_per_robot=3"}


def pytest_generate_tests(metafunc):
    if "kernel_path" in metafunc.fixturenames:
        on_gpu = metafunc.definition.get_closest_marker("gpu") is not None
        metafunc.parametrize("kernel_path", sorted(KERNEL_PATHS) if on_gpu else ["auto"], indirect=True)


@pytest.fixture(autouse=True)
def kernel_path(request, monkeypatch):
    if KERNEL_PATHS[request.param]:
        monkeypatch.setenv("VXH_ENGINE_OPTIONS", KERNEL_PATHS[request.param])
    else:
        monkeypatch.delenv("VXH_ENGINE_OPTIONS", raising=False)
    yield request.param


def free_port():
    """a rendezvous port nobody listens on right now: a FIXED --master-port made a second launch within the first one's TIME_WAIT retry for
    half a minute and more (round 6: test_two_ranks_share_the_gpu 34-77 s instead of 7 on one box in six)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
