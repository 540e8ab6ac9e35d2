#!/bin/bash
# GPU session r04-c: the fold as whole pipeline runs (run-segmented k-loop): GEMM parity tests + same-box A/B against no fold
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gemm_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q -k "sgemm or gemm or C4 or linear or mm" > $out/c_pytest.log 2>&1; echo "pytest rc=$?" >> $out/c_pytest.log
tail -5 $out/c_pytest.log
{
for rep in 1 2 3; do
  for v in main nofold; do
    lib=$root/benchmarks/_ab/$v.so; [ $v = main ] && lib=$root/neuronika_amd/lib/libneuronika_hip.so
    for l in "0 1" "0 0" "1 0"; do
      echo "rep$rep $v layout($l) 4096^3: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py $l 4096 4096 4096)"
    done
    echo "rep$rep $v TN 1024x1024x32768: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 1 0 1024 1024 32768)  NT 8192^3: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 1 8192 8192 8192) NN 4096x4096x8192: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 0 4096 4096 8192)"
  done
done
} 2>&1 | tee $out/c_fold_ab.txt
cp neuronika_amd/lib/libneuronika_hip.so /tmp/main.so
line='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], d["gemm_share_of_step"], d["loss"])'
{
for rep in 1 2; do
  echo "rep$rep fused      $(NK_BENCH_NO_SUBRECORDS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"
  cp $root/benchmarks/_ab/nofold.so neuronika_amd/lib/libneuronika_hip.so
  echo "rep$rep fused-nofold $(NK_BENCH_NO_SUBRECORDS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"
  cp /tmp/main.so neuronika_amd/lib/libneuronika_hip.so
done
} 2>&1 | tee $out/c_c4_ab.txt
