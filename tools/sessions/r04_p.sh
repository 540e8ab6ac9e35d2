#!/bin/bash
# GPU session r04-p: two GEMMs in one launch (nk_mm_bwd / nk_mm_t_bwd, sgemm_pair_kernel): parity tests, same-box A/B against two
# launches at 1024 / 2048 / 4096, and the refactored single-problem kernels against HEAD's build (benchmarks/_ab/head.so)
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mm_backward or sgemm or mm_golden or mm_t_golden or C2 or gemm" 2>&1 | tail -5 | tee $out/p_tests.txt
timeout 300 python -m pytest tests/test_gpu_tape.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out/p_tests.txt
timeout 300 python benchmarks/ab_pair.py 1024 2048 4096 2>&1 | tee $out/p_ab_pair.txt
{
for rep in 1 2; do
  for v in head main; do
    lib=$root/benchmarks/_ab/$v.so; [ $v = main ] && lib=$root/neuronika_amd/lib/libneuronika_hip.so
    for n in 4096 2048 1024; do
      echo "rep$rep $v $(AB_N=$n AB_CONV=0 NEURONIKA_HIP_LIB=$lib timeout 120 python benchmarks/ab_gemm.py)"
    done
  done
done
} 2>&1 | tee $out/p_ab_refactor.txt
timeout 400 python bench.py > $out/p_bench.json 2> $out/p_bench.err; tail -c 3000 $out/p_bench.json
