#!/bin/bash
# Round 5, session d: counters of the two Winograd kernels as the C3 step launches them.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=gpurun_out/r05d; mkdir -p $root/$out
cd $root
PMC_GROUPS="a b c e" bash tools/pmc_profile.sh $out/pmc conv_fwd conv_bwd_input > $root/$out/pmc_summary.txt 2>&1
cat $root/$out/pmc_summary.txt | grep -v "^  *$" | head -150
find $root/$out -name "*.db" -delete
