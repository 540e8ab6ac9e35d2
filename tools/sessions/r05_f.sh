#!/bin/bash
# Round 5, session f: the whole GPU suite and the driver's bench invocation on the tree with Winograd v3 (buffer addressing,
# quad V layout, XCD ranges, two-wave input-gradient blocks, staggered start).
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r05f; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/pytest.log
timeout -k 5 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r05f/bench_driver.json").read().strip().splitlines()[-1])
print("C4", r["ms_per_step"], r["roofline"]["frac"], "gemm share", r.get("gemm_share_of_step"))
for k in ("matmul_1024", "matmul_2048", "matmul_4096", "matmul_8192", "conv_c3", "mha_c5"):
    s = r.get(k) or {}
    print(k, s.get("ms_per_step"), (s.get("roofline") or {}).get("frac"), (s.get("roofline") or {}).get("executed_frac"))
PY
