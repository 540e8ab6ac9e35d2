#!/bin/bash
# HBM / fabric bytes per launch of the dominant kernels, as the MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 --pmc passes (TCC has four counter slots), never combined with sys / hip / hsa traces, every run
# under its own timeout.      bash tools/traffic_pmc.sh OUTDIR        then      python tools/make_roofline_traffic.py OUTDIR
set -u
out=$1
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$root/$out"
for tgt in gemm_once conv_step_once mha_step_once; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 120 rocprofv3 --pmc $c --kernel-trace -d "$root/$out/$tgt/$c" -o r -- python "$root/benchmarks/$tgt.py" > "$root/$out/${tgt}_$c.log" 2>&1 || echo "FAILED $tgt $c"
  done
done
