#!/bin/bash
# Round 6, session f: Winograd forward / input gradient for odd output extents (ODD instantiations) and the forward's whole-grid split:
# the conv suites, then the layer-shape table (benchmarks/conv_shapes.py) against profiles/r05_conv_shapes.jsonl.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r06f; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv_fuzz.py tests/test_gpu_parity.py -m gpu -q -x -k "conv or wino or pad" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout -k 5 300 python benchmarks/conv_shapes.py > $out/conv_shapes.jsonl 2> $out/conv_shapes.err
python - <<'P'
import json
old = {json.loads(l)["shape"]: json.loads(l) for l in open("profiles/r05_conv_shapes.jsonl") if l.startswith("{")}
for l in open("gpurun_out/r06f/conv_shapes.jsonl"):
    if not l.startswith("{"): continue
    d = json.loads(l); o = old.get(d["shape"], {})
    print("%-44s fwd %7.1f (%7.1f)  dx %7.1f (%7.1f)  dw %7.1f (%7.1f) us" % (d["shape"], d["fwd"][0], o.get("fwd", [0])[0], d["bwd_input"][0], o.get("bwd_input", [0])[0], d["bwd_kernel"][0], o.get("bwd_kernel", [0])[0]))
P
