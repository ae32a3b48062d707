import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vectors():
    """Literal vectors transcribed from the reference's tests (tests/golden/make_golden.py part A)."""
    return json.loads((GOLDEN / "reference_vectors.json").read_text())


@pytest.fixture(scope="session")
def generated():
    """Outputs of the reference's NumPy helpers (tests/golden/make_golden.py part B)."""
    return np.load(GOLDEN / "generated.npz")


@pytest.fixture(scope="session")
def generated_meta():
    return json.loads((GOLDEN / "generated_meta.json").read_text())


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="module")
def hip_env():
    """(ctx, cq) -- the analogue of the reference's cl_env fixture (tests/conftest.py:4-12)."""
    from collision_amd import hip
    ctx = hip.Context()
    cq = hip.CommandQueue(ctx)
    return ctx, cq
