#!/bin/bash
# GPU session Q (round 3): GEMM kernels with the thread index masked to 8 bits (shorter address code) against the shipped build, alternating
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
for rep in 1 2 3; do
  for lib in "" benchmarks/_ab/tmask.so; do
    NEURONIKA_HIP_LIB=$lib AB_CONV=0 python benchmarks/ab_gemm.py
    NEURONIKA_HIP_LIB=$lib AB_CONV=0 AB_N=2048 python benchmarks/ab_gemm.py
    NEURONIKA_HIP_LIB=$lib AB_CONV=0 AB_N=1024 python benchmarks/ab_gemm.py
  done
done 2>&1 | tee $out/q_tmask.jsonl
for lib in "" benchmarks/_ab/tmask.so; do
  for sh in "0 1 32768 1024 1024" "0 0 32768 1024 1024" "1 0 1024 1024 32768"; do
    echo "lib=$lib $sh $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py $sh)"
  done
done 2>&1 | tee -a $out/q_tmask.jsonl
