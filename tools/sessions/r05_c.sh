#!/bin/bash
# Round 5, session c: Winograd F(2x2, 3x3) forward / input gradient - parity (new tests first, then the whole suite), same-box
# A/B of the two passes, the C3 step both ways with its kernel table, the driver's bench invocation (timed).
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r05c; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests/test_gpu_winograd.py -q -x > $out/pytest_winograd.log 2>&1; echo "pytest winograd rc=$?" | tee -a $out/pytest_winograd.log
tail -25 $out/pytest_winograd.log
timeout -k 5 300 python benchmarks/ab_winograd.py > $out/ab_winograd.jsonl 2> $out/ab_winograd.err; echo "ab rc=$?"
cat $out/ab_winograd.jsonl; tail -3 $out/ab_winograd.err
for m in 0 1; do
  NK_CONV_WINOGRAD=$m timeout -k 5 300 python bench.py --workload conv --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_conv_wino$m.json 2> $out/bench_conv_wino$m.err
  python -c "
import json; r=json.load(open('$out/bench_conv_wino$m.json')); print('winograd=$m', r['ms_per_step'], 'ms/step', r['roofline']['frac'], r.get('conv_share_of_step'))"
done
export TMPDIR=/tmp
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $out/prof_conv -o r -- python $root/benchmarks/conv_step_once.py > $out/prof_conv.log 2>&1 ); echo "rocprof rc=$?"
db=$(find $out/prof_conv -name "*_results.db" | head -1)
[ -n "$db" ] && python tools/rocpd_kernel_stats.py "$db" > $out/conv_step_kernel_stats.md 2>> $out/prof_conv.log; head -30 $out/conv_step_kernel_stats.md
find $out -name "*.db" -delete
timeout -k 5 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_multi.py > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -8 $out/pytest.log
cp gpurun_out/tolerance_margins.json $out/tolerance_margins.json 2>/dev/null
start=$(date +%s.%N)
timeout -k 5 400 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
end=$(date +%s.%N); echo "bench wall $(echo "$end - $start" | bc) s"
python - <<'PY'
import json
r=json.load(open("gpurun_out/r05c/bench_default.json"))
print({k: r[k] for k in ("value","ms_per_step","gemm_share_of_step")}, r["roofline"]["frac"])
for k in ("matmul_1024","matmul_4096","conv_c3","mha_c5"):
    print(k, r[k].get("ms_per_step"), r[k].get("value"), (r[k].get("cpu_baseline") or {}).get("value"))
print(json.dumps(r["hbm_kernels"]["kernels"], indent=0)[:1500])
PY
