"""CPU checks of the drop-in boundary: the C-ABI library builds, loads, and exports every
symbol include/neuronika_hip.h declares (no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "neuronika_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from neuronika_amd import capi
    syms = header_symbols()
    assert len(syms) > 60
    missing = [s for s in syms if not hasattr(capi.lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    # and the ctypes table binds exactly the header's functions
    assert sorted(capi.EXPORTED) == syms


def test_no_gpu_means_loud_failure():
    from neuronika_amd import capi
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(capi.NeuronikaHipError):
        capi.Device(0)


def test_product_path_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under neuronika_amd/ or include/ may use it."""
    bad = []
    for base in ("neuronika_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle|neuronika_oracle", txt, re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_examples_do_not_depend_on_test_infrastructure():
    """examples/ ships its own data: the quickstart model JSON (the serialized network of the reference's
    examples/quickstart.rs:53-169) lives next to the example and equals the vectors the extractor took from the
    reference; nothing under examples/ reads tests/ or the oracle."""
    import json
    for f in os.listdir(os.path.join(ROOT, "examples")):
        if f.endswith(".py"):
            txt = open(os.path.join(ROOT, "examples", f)).read()
            assert '"tests"' not in txt and "tests/" not in txt and "oracle" not in txt, f
    model = json.load(open(os.path.join(ROOT, "examples", "quickstart_model.json")))["model"]
    q = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_fixtures.json")))["quickstart_mlp"]
    for name in ("lin1", "lin2", "lin3"):
        for p in ("weight", "bias"):
            assert model[name][p]["dim"] == q[f"{name}.{p}"]["dim"] and model[name][p]["data"] == q[f"{name}.{p}"]["data"]
