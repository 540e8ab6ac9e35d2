#!/bin/bash
# GPU session I (round 3): k-pair GEMM blocks - parity of the GEMM suite, then the same-box sweep against unsplit / split-K
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sgemm or gemm or matmul or mm_" > $out/i_pytest.log 2>&1; echo "pytest rc=$?" >> $out/i_pytest.log
tail -3 $out/i_pytest.log
timeout -k 5 500 python benchmarks/ab_kpair.py > $out/i_ab_kpair.jsonl 2> $out/i_ab_kpair.err
cat $out/i_ab_kpair.jsonl
