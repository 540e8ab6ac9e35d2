#!/bin/bash
# Round 5, session g: the round's closing profile set (tools/gpu_session_final.sh at the commit passed as $1) plus the counters of
# the two Winograd kernels as the C3 step launches them and the Winograd A/B tables.
set -u
commit=${1:-unknown}
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out/r05g
cd $root
bash tools/gpu_session_final.sh $commit r05 2>&1 | tail -40
PMC_GROUPS="a b c e" bash tools/pmc_profile.sh gpurun_out/r05g/pmc conv_fwd conv_bwd_input > $out/r05g/pmc_summary.txt 2>&1
grep -E "^## wino|GRBM_GUI|MFMA|INSTS_VALU|WAIT_INST_ANY|WAVE_CYCLES|LDS_BANK|BUSY_CYCLES|TCC_|VMEM_RD|INSTS_LDS|INSTS_SALU" $out/r05g/pmc_summary.txt | head -80
timeout -k 5 300 python benchmarks/ab_winograd.py > $out/r05g/ab_winograd.jsonl 2> $out/r05g/ab_winograd.err
timeout -k 5 300 python benchmarks/ab_winograd.py 128 shape > $out/r05g/ab_winograd_shape.jsonl 2>> $out/r05g/ab_winograd.err
timeout -k 5 300 python benchmarks/ab_winograd.py 128 stagger > $out/r05g/ab_winograd_stagger.jsonl 2>> $out/r05g/ab_winograd.err
find $out -name "*.db" -delete
