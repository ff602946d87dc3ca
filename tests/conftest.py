import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "l1_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_hc():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "l234_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_lazy():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "l59_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_near_optimal():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "l1012_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def emu_lib():
    """The product sources compiled against the CPU SIMT emulator (kernel-logic checks only)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from gzp_amd import _native
    return _native.GzpxLib(build_emu.build())


@pytest.fixture(scope="session")
def hip_lib():
    """The real HIP library; GPU tests fail loudly if it is missing."""
    from gzp_amd import _native
    return _native.load()
