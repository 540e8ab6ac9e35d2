#!/bin/bash
# GPU session r04-x: PMC groups a / b / e (instruction mix, MFMA-pipe occupancy) of the round's hot kernels as they are at the round's end:
# the three C3 passes (kernel gradient = the mixed launch), the 4096^3 GEMM, the fused attention core, the dK / dV products
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
PMC_GROUPS="a b e" bash tools/pmc_profile.sh gpurun_out/x_pmc conv_fwd conv_bwd_input conv_bwd_kernel gemm4k attn_fwd attn_bwd 2>&1 | tee $out/x_pmc_summary.txt | tail -5
find $out/x_pmc -name "*.db" -delete
