#!/bin/bash
# GPU session A (round 3): parity suite, the new default bench line, attention core A/B against the round-2 library.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests -m gpu -x -q > $out/a_pytest.log 2>&1; echo "pytest rc=$?" >> $out/a_pytest.log
tail -5 $out/a_pytest.log
timeout -k 5 300 python bench.py --steps 20 --warmup 5 > $out/a_bench_default.json 2> $out/a_bench_default.err; echo "bench rc=$?"
NK_BENCH_FORCE_RCCL=1 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,TUNING,ENV NCCL_DEBUG_FILE=$out/a_rccl_1rank.log timeout -k 5 200 python bench.py --steps 3 --warmup 1 --hidden 1024 --batch 512 --no-cpu-baseline > $out/a_bench_force_rccl.json 2> $out/a_bench_force_rccl.err
for i in 1 2; do
  NEURONIKA_HIP_LIB=$root/benchmarks/_ab/r02.so timeout -k 5 120 python benchmarks/attention_core.py 32 1024 16 10 > $out/a_attn_r02_$i.jsonl 2>&1
  timeout -k 5 120 python benchmarks/attention_core.py 32 1024 16 10 > $out/a_attn_new_$i.jsonl 2>&1
done
grep -h fused $out/a_attn_*.jsonl | head -20
