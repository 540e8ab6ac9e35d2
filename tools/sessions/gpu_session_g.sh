#!/bin/bash
# GPU session G (round 3): which weight gradients to hand over in two row blocks (NK_DP_PARTS) under the paced stand-in exchange
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
: > $out/g_parts.jsonl
for rep in 1 2; do
for g in 0 240 120 60; do
  for parts in all last none; do
    ch=32; [ $g = 0 ] && ch=0
    NK_DP_PARTS=$parts NK_BENCH_REPLICAS=8 NK_REPLICA_CHANNELS=$ch NK_REPLICA_GBPS=$g NK_BENCH_NO_SUBRECORDS=1 timeout -k 5 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2> $out/g_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'algbw_GBps': $g, 'parts': '$parts', 'ms_per_step': d['ms_per_step'], 'overhead_ms': d.get('replica_step_overhead_ms'), 'gemm': d['gemm_contention'], 'launches': d['allreduce_launches_per_step']}))" >> $out/g_parts.jsonl
  done
done
done
cat $out/g_parts.jsonl | cut -c1-230
timeout -k 5 400 python -m pytest tests/test_gpu_tape.py tests/test_gpu_multi.py -m gpu -x -q > $out/g_pytest.log 2>&1; tail -3 $out/g_pytest.log
