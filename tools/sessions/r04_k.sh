#!/bin/bash
# GPU session r04-k: bench with launch events on every 4th timed step: the bench tests + the default line
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests/test_gpu_multi.py -m gpu -x -q > $out/k_pytest.log 2>&1; echo "pytest rc=$?" >> $out/k_pytest.log
tail -4 $out/k_pytest.log
timeout -k 5 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/k_bench_default.json 2> $out/k_bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/k_bench_default.json').read().strip().splitlines()[-1])
print("C4", d["value"], d["ms_per_step"], d["roofline"], "share", d["gemm_share_of_step"])
for k in ("matmul_1024","matmul_2048","matmul_4096","matmul_8192","conv_c3","mha_c5"):
    r=d[k]; print(k, r.get("ms_per_step"), r.get("value"), r.get("roofline",{}).get("frac"), r.get("roofline",{}).get("launches"))
P
