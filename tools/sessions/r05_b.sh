#!/bin/bash
# Round 5, session b: the whole GPU suite (every failure listed, no -x), the interleaved under-load A/B of the shared-chip
# schedule, the driver's bench invocation with the new sub-records (hbm_kernels, per-configuration CPU baselines).
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r05b; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_multi.py > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -12 $out/pytest.log
cp gpurun_out/tolerance_margins.json $out/tolerance_margins.json 2>/dev/null
timeout -k 5 300 python benchmarks/gemm_under_load.py defence > $out/gemm_under_load_defence.jsonl 2> $out/gemm_under_load_defence.err; echo "under_load rc=$?"
cat $out/gemm_under_load_defence.jsonl; tail -3 $out/gemm_under_load_defence.err
/usr/bin/time -v timeout -k 5 400 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
grep -E "Elapsed|Maximum resident" $out/bench_default.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r05b/bench_default.json"))
print({k: r[k] for k in ("value","ms_per_step","gemm_share_of_step")}, r["roofline"]["frac"])
for k in ("matmul_1024","matmul_4096","conv_c3","mha_c5"):
    print(k, r[k].get("ms_per_step"), r[k].get("value"), (r[k].get("cpu_baseline") or {}).get("value"))
print(json.dumps(r["hbm_kernels"], indent=0))
PY
