#!/bin/bash
# GPU session r04-e: packed Q/K/V projections of nn::MultiheadAttention (one GEMM each way, attention kernels on the packed layout):
# parity suite, C5 step packed against three Linear nodes, the default bench line.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $out/e_pytest.log 2>&1; echo "pytest rc=$?" >> $out/e_pytest.log
tail -12 $out/e_pytest.log
cp $out/tolerance_margins.json $out/e_tolerance_margins.json 2>/dev/null
line='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], d.get("gemm_share_of_step"))'
{
for rep in 1 2 3; do
  echo "rep$rep packed   $(python bench.py --workload mha --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"
  echo "rep$rep unpacked $(NK_BENCH_UNPACKED_QKV=1 python bench.py --workload mha --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"
done
} 2>&1 | tee $out/e_c5_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $out/e_prof_mha -o r -- python $root/bench.py --workload mha --steps 10 --warmup 2 --no-cpu-baseline > $out/e_prof_mha.log 2>&1
db=$(find $out/e_prof_mha -name "*_results.db" | head -1)
[ -n "$db" ] && python $root/tools/rocpd_kernel_stats.py "$db" > $out/e_mha_step_kernel_stats.md
find $out/e_prof_mha -name "*.db" -delete
head -30 $out/e_mha_step_kernel_stats.md
