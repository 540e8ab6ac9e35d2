#!/bin/bash
# GPU session r04-d: the round's GEMM as it stays (one chain, fused ReLU epilogues, explicit tuning API), full parity suite, C4 step
# fused / node by node, and the default bench line as the driver runs it.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $out/d_pytest.log 2>&1; echo "pytest rc=$?" >> $out/d_pytest.log
tail -12 $out/d_pytest.log
cp $out/tolerance_margins.json $out/d_tolerance_margins.json 2>/dev/null
line='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], d["gemm_share_of_step"], d["loss"])'
{
for rep in 1 2 3; do
  echo "rep$rep fused      $(NK_BENCH_NO_SUBRECORDS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"
  echo "rep$rep unfused    $(NK_BENCH_UNFUSED_RELU=1 NK_BENCH_NO_SUBRECORDS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"
done
} 2>&1 | tee $out/d_c4_ab.txt
timeout -k 5 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/d_bench_default.json 2> $out/d_bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/d_bench_default.json').read().strip().splitlines()[-1])
print("C4", d["ms_per_step"], d["roofline"]["frac"], "gemm_share", d["gemm_share_of_step"])
for k in ("matmul_1024","matmul_2048","matmul_4096","matmul_8192","conv_c3","mha_c5"):
    r=d[k]; print(k, r.get("ms_per_step"), r.get("value"), r.get("roofline",{}).get("frac"))
P
