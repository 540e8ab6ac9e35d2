#!/bin/bash
# GPU session K (round 3): k-pair blocks, halves vs alternate k-tiles
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kpair or override" > $out/k_pytest.log 2>&1; echo "pytest rc=$?" >> $out/k_pytest.log
tail -3 $out/k_pytest.log
timeout -k 5 700 python benchmarks/ab_kpair.py > $out/k_ab_kpair.jsonl 2> $out/k_ab_kpair.err
cat $out/k_ab_kpair.jsonl
