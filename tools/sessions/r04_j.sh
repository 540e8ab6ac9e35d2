#!/bin/bash
# GPU session r04-j: what the per-launch HIP event pairs of bench.py's roofline instrumentation cost the timed steps
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
{
for rep in 1 2 3; do
  for w in mlp mha conv; do
    a=$(NK_BENCH_NO_SUBRECORDS=1 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')
    b=$(NK_BENCH_NO_LAUNCH_EVENTS=1 NK_BENCH_NO_SUBRECORDS=1 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')
    echo "rep$rep $w with events $a  without $b"
  done
done
} 2>&1 | tee $out/j_event_cost.txt
