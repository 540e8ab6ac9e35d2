#!/bin/bash
# GPU session r04-m: did the per-pointer row strides cost the attention kernels anything?  C5 with three projection nodes (all strides
# equal) under the current library and a variant whose kernel uses one stride.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
cp neuronika_amd/lib/libneuronika_hip.so /tmp/main.so
line='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], json.dumps(d.get("attention_core")))'
{
for rep in 1 2 3; do
  for v in main attn1ld; do
    src=/tmp/main.so; [ $v != main ] && src=$root/benchmarks/_ab/$v.so
    cp $src neuronika_amd/lib/libneuronika_hip.so
    echo "rep$rep $v $(NK_BENCH_UNPACKED_QKV=1 python bench.py --workload mha --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"
  done
done
cp /tmp/main.so neuronika_amd/lib/libneuronika_hip.so
} 2>&1 | tee $out/m_attn_ld_ab.txt
