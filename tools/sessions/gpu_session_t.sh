#!/bin/bash
# GPU session T (round 3): tile-order group height (NK_GEMM_FORCE 5th field) against fabric fetch traffic and speed at 4096^3
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out/t_pmc
cd /tmp && export TMPDIR=/tmp
for gm in 8 4 16 32; do
  export NK_GEMM_FORCE="2,2,1,1,$gm"
  timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/t_pmc/g$gm -o r -- python $root/benchmarks/gemm_once.py > $out/t_pmc/g$gm.log 2>&1
  echo "group_m=$gm"
  python $root/tools/rocpd_pmc_summary.py $(find $out/t_pmc/g$gm -name "*_results.db") 2>&1 | grep -E "^## sgemm|FETCH_SIZE"
  for l in "0 1" "0 0" "1 0"; do echo "  $l $(python $root/benchmarks/ab_force.py $l 4096 4096 4096)"; done
done 2>&1 | tee $out/t_group_m.txt
find $out/t_pmc -name "*.db" -delete
