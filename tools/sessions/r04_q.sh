#!/bin/bash
# GPU session r04-q: nk_sgemm_pair rule (256-thread blocks, saves-a-wave criterion), batched pair for dK / dV, conv small launches
# (tap table in LDS, one reduce launch for dW + db), the refactored single-problem GEMM kernels against HEAD's build
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_attention.py tests/test_gpu_tape.py tests/test_gpu_conv_fuzz.py -x -q -m gpu 2>&1 | tail -5 | tee $out/q_tests.txt
timeout 300 python benchmarks/ab_pair.py 1024 2048 4096 2>&1 | tee $out/q_ab_pair.txt
{
for rep in 1 2; do
  for v in head main; do
    lib=$root/benchmarks/_ab/$v.so; [ $v = main ] && lib=$root/neuronika_amd/lib/libneuronika_hip.so
    for n in 4096 2048 1024; do
      cv=0; [ $n = 4096 ] && cv=1
      echo "rep$rep $v $(AB_N=$n AB_CONV=$cv NEURONIKA_HIP_LIB=$lib timeout 120 python benchmarks/ab_gemm.py 2>&1 | tail -1)"
    done
  done
done
} 2>&1 | tee $out/q_ab_refactor.txt
for w in matmul conv mha; do timeout 300 python bench.py --workload $w --no-cpu-baseline > $out/q_bench_$w.json 2> $out/q_bench_$w.err; done
python - <<'PY'
import json
for w in ("matmul", "conv", "mha"):
    try:
        d = json.load(open(f"gpurun_out/q_bench_{w}.json"))
        print(w, d.get("ms_per_step"), d.get("value"), d.get("roofline", {}).get("frac"), {k: (v.get("ms_per_step"), v.get("roofline", {}).get("frac")) for k, v in d.items() if isinstance(v, dict) and "ms_per_step" in v})
    except Exception as e:
        print(w, "failed", e)
PY
