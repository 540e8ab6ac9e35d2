#!/bin/bash
# Round 5, session h: the closing profile set on the tree with the Winograd kernel gradient (tools/gpu_session_final.sh at the commit
# passed as $1), counters of the three Winograd kernels, the Winograd A/B tables (C3, block shapes, kernel gradient).
set -u
commit=${1:-unknown}
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out/r05h
cd $root
bash tools/gpu_session_final.sh $commit r05 2>&1 | tail -40
PMC_GROUPS="a b c e" bash tools/pmc_profile.sh gpurun_out/r05h/pmc conv_fwd conv_bwd_input conv_bwd_kernel > $out/r05h/pmc_summary.txt 2>&1
timeout -k 5 300 python benchmarks/ab_winograd.py > $out/r05h/ab_winograd.jsonl 2> $out/r05h/ab_winograd.err
timeout -k 5 300 python benchmarks/ab_winograd.py 128 shape > $out/r05h/ab_winograd_shape.jsonl 2>> $out/r05h/ab_winograd.err
timeout -k 5 300 python benchmarks/ab_winograd.py 128 dw > $out/r05h/ab_winograd_dw.jsonl 2>> $out/r05h/ab_winograd.err
find $out -name "*.db" -delete
