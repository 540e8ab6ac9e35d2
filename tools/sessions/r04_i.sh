#!/bin/bash
# GPU session r04-i: look-ahead / tile sweep for the N = 64 attention products (dK / dV: TN, context: NN; 512 heads, S = 1024)
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
{
for rep in 1 2; do
  for op in dV context; do
    for f in "" "2,1,1,1,8,16" "2,1,1,1,8,8" "2,1,1,1,8,24" "2,1,1,2,8,100" "1,1,1,1,8,8" "1,1,1,1,8,100"; do
      NK_GEMM_FORCE="$f" python benchmarks/ab_attn_gemm.py $op
    done
  done
done
} 2>&1 | tee $out/i_attn_gemm_sweep.jsonl
