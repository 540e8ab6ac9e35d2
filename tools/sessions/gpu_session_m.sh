#!/bin/bash
# GPU session M (round 3): counters of the 2048^3 GEMM with plain and k-pair blocks
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
NK_GEMM_KPAIR=0 PMC_GROUPS="a b e" bash tools/pmc_profile.sh gpurun_out/m_pmc_plain gemm2k > $out/m_pmc_plain.txt 2>&1
PMC_GROUPS="a b e" bash tools/pmc_profile.sh gpurun_out/m_pmc_pair gemm2k > $out/m_pmc_pair.txt 2>&1
find $out/m_pmc_plain $out/m_pmc_pair -name "*.db" -delete
tail -40 $out/m_pmc_plain.txt; tail -40 $out/m_pmc_pair.txt
