#!/bin/bash
# GPU session r04-r: conv kernel gradient, mixed wide / narrow launch (parity + price sweep at C3); the parity suites of the
# round's changes with their pass / fail lines; default bench line
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_attention.py tests/test_gpu_tape.py tests/test_gpu_conv_fuzz.py -x -q -m gpu > $out/r_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/r_pytest.log
grep -E "passed|failed|Error|error" $out/r_pytest.log | tail -5
timeout 300 python benchmarks/ab_conv_narrow.py 2>&1 | tail -1 | tee $out/r_conv_narrow.txt
timeout 300 python benchmarks/ab_conv_narrow.py 0 70 0 70 2>&1 | tail -1 | tee -a $out/r_conv_narrow.txt
for f in "0 0 32768 1024 3072" "0 0 32768 1024 1024" "0 1 32768 3072 1024" "1 0 3072 1024 32768"; do echo "$f: $(timeout 120 python benchmarks/ab_force.py $f 2>&1 | tail -1)"; done | tee $out/r_c5_shapes.txt
timeout 400 python bench.py > $out/r_bench.json 2> $out/r_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r_bench.json"))
print("C4", d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("gemm_share_of_step"))
for k, v in d.items():
    if isinstance(v, dict) and "ms_per_step" in v:
        print(k, v["ms_per_step"], v.get("value"), v["roofline"]["frac"])
PY
