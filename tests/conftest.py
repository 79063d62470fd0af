import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The parity oracle (oracle/oracle.py); builds liboracle.so on first use."""
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def golden():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))


@pytest.fixture(scope="session")
def ref_images():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "ref_small_images.npz")))


@pytest.fixture(scope="session")
def scenes():
    from spath_amd import scene
    return {
        "default": scene.default_scene(),
        "closed_room_200": scene.closed_room(200),
        "open_clutter_100": scene.open_clutter(100),
    }


@pytest.fixture(scope="session")
def hip():
    """A capi.Context on cuda:0; fails (does not skip) when the HIP library cannot be used on a GPU box."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this container")
    from spath_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
