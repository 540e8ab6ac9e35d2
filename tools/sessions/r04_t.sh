#!/bin/bash
# GPU session r04-t: (1) does a block walking several tiles pay at 18 k-tiles per tile?  The dense GEMM on the conv-forward-like
# shape 128 x 401408 x 576 (NN) with 1 / 2 / 3 / 4 tiles per block (sgemm_kernel's chunk loop); (2) issued MFMAs of the kernel
# gradient's mixed launch (PMC group a)
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
{
for rep in 1 2; do
  for f in "2,2,1,1" "2,2,1,2" "2,2,1,3" "2,2,1,4" "2,2,1,6"; do
    echo "rep$rep chunk-force $f: NN 128x401408x576 $(NK_GEMM_FORCE=$f timeout 120 python benchmarks/ab_force.py 0 0 128 401408 576 2>&1 | tail -1)  NT $(NK_GEMM_FORCE=$f timeout 120 python benchmarks/ab_force.py 0 1 128 401408 576 2>&1 | tail -1)"
  done
done
} | tee $out/t_chunk_conv_shape.txt
PMC_GROUPS="a" bash tools/pmc_profile.sh gpurun_out/t_pmc conv_bwd_kernel 2>&1 | tail -30 | tee $out/t_pmc_conv_bwd_kernel.txt
find $out/t_pmc -name "*.db" -delete
