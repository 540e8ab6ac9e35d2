#!/bin/bash
# Round 6, session a: the parity suite on the tree with chained GEMM launches (GEMM_CHAIN_K = 2048), the folded-entry fallback and the
# rewritten C3 full-size test; then what the chain costs, same box, alternating runs: C4 step and C2 at 4096 / 8192 with
# NK_GEMM_CHAIN=0 (one chain, round 5's launches) against the rule.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r06a; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
cp $root/gpurun_out/tolerance_margins.json $out/tolerance_margins.json 2>/dev/null
export NK_BENCH_NO_SUBRECORDS=2
for rep in 1 2 3; do
  for chain in 0 rule; do
    if [ $chain = rule ]; then unset NK_GEMM_CHAIN; else export NK_GEMM_CHAIN=$chain; fi
    for w in "mlp" "matmul --n 4096" "matmul --n 8192"; do
      tag=$(echo $w | tr -d ' -')
      timeout -k 5 200 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>> $out/ab.err | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print(json.dumps({'chain': '$chain', 'rep': $rep, 'workload': '$w', 'ms_per_step': d.get('ms_per_step'), 'value': d.get('value'), 'unit': d.get('unit'), 'gemm_frac': r.get('frac'), 'launches': r.get('launches'), 'avg_launch_ms': r.get('avg_launch_ms')}))
" >> $out/chain_ab.jsonl
    done
  done
done
unset NK_GEMM_CHAIN
cat $out/chain_ab.jsonl
