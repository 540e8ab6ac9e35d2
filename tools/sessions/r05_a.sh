#!/bin/bash
# Round 5, session a: the whole GPU suite at the new tolerance helper / peephole / shared-chip schedule; the driver's bench
# invocation (C4 now spelled forward(x).relu()); GEMMs beside CU-occupying load with and without nk_device_set_busy_slots;
# the C4 step over the paced stand-in exchange with and without GradientSync.set_busy_slots.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r05a; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_multi.py > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
cp gpurun_out/tolerance_margins.json $out/tolerance_margins.json 2>/dev/null
timeout -k 5 300 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
r=json.load(open("gpurun_out/r05a/bench_default.json"))
print({k: r[k] for k in ("value","ms_per_step","gemm_share_of_step")}, r["roofline"]["frac"])
for k in ("matmul_1024","matmul_4096","conv_c3","mha_c5"):
    print(k, r[k].get("ms_per_step"), r[k].get("value"))
PY
timeout -k 5 300 python benchmarks/gemm_under_load.py defence > $out/gemm_under_load_defence.jsonl 2> $out/gemm_under_load_defence.err; echo "under_load rc=$?"
cat $out/gemm_under_load_defence.jsonl
timeout -k 5 300 python benchmarks/overlap_projection.py 30 defence > $out/overlap_defence.jsonl 2> $out/overlap_defence.err; echo "overlap rc=$?"
cat $out/overlap_defence.jsonl
