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


# ---- measured tolerance margins -------------------------------------------------------------------------------------
# The parity bound for contractions is  err_gpu <= max(k * err_cpu32, a * K * |a| * |b|)  against the f64 oracle
# (SURVEY.md 8c ii: k = 2, a = 1e-6 - every contraction test asserts exactly that since round 4).
# Every such check reports here; the session writes gpurun_out/tolerance_margins.json = per label the worst
# err_gpu / (2 * err_cpu32), err_gpu / abs_term(1e-6) and their minimum (<= 1: inside the stated policy).
_MARGINS = {}


def record_margin(label, err_gpu, err_cpu32, abs_term_1e6):
    """abs_term_1e6: the absolute term of the STATED policy (1e-6 * K * |a| * |b|, or 1e-6 * scale)."""
    m = _MARGINS.setdefault(label, {"n": 0, "vs_2x_cpu32": 0.0, "vs_abs_1e-6": 0.0, "vs_stated_policy": 0.0})
    r_cpu = float(err_gpu) / (2.0 * float(err_cpu32)) if err_cpu32 > 0 else float("inf")
    r_abs = float(err_gpu) / float(abs_term_1e6) if abs_term_1e6 > 0 else float("inf")
    m["n"] += 1
    m["vs_2x_cpu32"] = max(m["vs_2x_cpu32"], r_cpu if r_cpu != float("inf") else 0.0)
    m["vs_abs_1e-6"] = max(m["vs_abs_1e-6"], r_abs if r_abs != float("inf") else 0.0)
    m["vs_stated_policy"] = max(m["vs_stated_policy"], min(r_cpu, r_abs))   # <= 1: inside the stated 2x / 1e-6 policy


def pytest_sessionfinish(session, exitstatus):
    if not _MARGINS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "tolerance_margins.json"), "w") as f:
            json.dump(_MARGINS, f, indent=1, sort_keys=True)
    except OSError:
        pass
