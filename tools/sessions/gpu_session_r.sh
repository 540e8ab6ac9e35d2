#!/bin/bash
# GPU session R (round 3): conv and attention kernels with the thread index masked to 8 bits against the unmasked build, alternating
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
for rep in 1 2 3; do
  for lib in "" benchmarks/_ab/notmask.so; do
    NEURONIKA_HIP_LIB=$lib python benchmarks/ab_gemm.py
    echo "lib=$lib"; NEURONIKA_HIP_LIB=$lib python benchmarks/attention_core.py 32 1024 16 10 | grep fused | tail -2
  done
done 2>&1 | tee $out/r_tmask.jsonl
