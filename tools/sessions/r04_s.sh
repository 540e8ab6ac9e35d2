#!/bin/bash
# GPU session r04-s: the whole GPU suite and the default bench line at the mixed conv kernel-gradient default + C2 through nk_mm_bwd
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 1500 python -m pytest tests -x -q -m gpu > $out/s_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/s_pytest.log
grep -E "passed|failed|Error" $out/s_pytest.log | tail -5
timeout 400 python bench.py > $out/s_bench.json 2> $out/s_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/s_bench.json"))
print("C4", d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("gemm_share_of_step"))
for k, v in d.items():
    if isinstance(v, dict) and "ms_per_step" in v:
        print(k, v["ms_per_step"], v.get("value"), v["roofline"]["frac"], v.get("frac_of_mfma_peak"))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
