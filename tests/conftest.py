import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def golden():
    """Known-answer vectors transcribed from the reference's own unit tests by
    tests/golden/extract_reference_fixtures.py."""
    with open(os.path.join(ROOT, "tests", "golden", "reference_fixtures.json")) as f:
        return json.load(f)


def _has_gpu():
    try:
        from neuronika_amd import capi
        return capi.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def dev():
    """One `Device` (mirror of the reference's `cuda::Device`, cuda/device.rs:34-58) on GPU 0.
    The HIP library must load — a GPU test never falls back to a CPU path."""
    from neuronika_amd import capi
    d = capi.Device(0)
    yield d
    d.sync()
