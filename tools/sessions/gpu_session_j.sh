#!/bin/bash
# GPU session J (round 3): k-pair blocks with the swapped-halves epilogue - GEMM parity, then plain vs pair on large grids
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py -m gpu -x -q -k "sgemm or gemm or matmul or mm_ or determin" > $out/j_pytest.log 2>&1; echo "pytest rc=$?" >> $out/j_pytest.log
tail -3 $out/j_pytest.log
timeout -k 5 700 python benchmarks/ab_kpair_large.py 3 > $out/j_ab_kpair_large.jsonl 2> $out/j_ab_kpair_large.err
cat $out/j_ab_kpair_large.jsonl
