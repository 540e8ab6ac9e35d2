#!/bin/bash
# GPU session r04-f: k-pair blocks for 64x64 tiles (1024^3: one block per CU -> two waves per SIMD without split-K): parity,
# same-box A/B at 1024^3 and on the shapes the new rule touches, the C2 lines.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $out/f_pytest.log 2>&1; echo "pytest rc=$?" >> $out/f_pytest.log
tail -8 $out/f_pytest.log
{
for rep in 1 2 3; do
  for l in "0 0" "0 1" "1 0"; do
    echo "rep$rep 1024^3 layout($l): none $(NK_GEMM_KPAIR=0 python benchmarks/ab_force.py $l 1024 1024 1024)  lock-step $(NK_GEMM_KPAIR=1 python benchmarks/ab_force.py $l 1024 1024 1024)  skewed $(NK_GEMM_KPAIR=2 python benchmarks/ab_force.py $l 1024 1024 1024)  rules $(python benchmarks/ab_force.py $l 1024 1024 1024)"
  done
  echo "rep$rep 1024x1024x2048 NN: none $(NK_GEMM_KPAIR=0 python benchmarks/ab_force.py 0 0 1024 1024 2048) rules $(python benchmarks/ab_force.py 0 0 1024 1024 2048)   768^3 NN: none $(NK_GEMM_KPAIR=0 python benchmarks/ab_force.py 0 0 768 768 768) rules $(python benchmarks/ab_force.py 0 0 768 768 768)  512x512x1024 NT: none $(NK_GEMM_KPAIR=0 python benchmarks/ab_force.py 0 1 512 512 1024) rules $(python benchmarks/ab_force.py 0 1 512 512 1024)"
done
} 2>&1 | tee $out/f_pair64_ab.txt
for rep in 1 2; do
  python bench.py --workload matmul --n 1024 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("C2 1024 rules:", d.get("value"), d.get("ms_per_step"), d.get("roofline",{}).get("frac"))'
  NK_GEMM_KPAIR=0 python bench.py --workload matmul --n 1024 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("C2 1024 no pair:", d.get("value"), d.get("ms_per_step"), d.get("roofline",{}).get("frac"))'
done 2>&1 | tee $out/f_c2_1024.txt
