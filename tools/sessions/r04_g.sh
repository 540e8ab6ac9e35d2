#!/bin/bash
# GPU session r04-g: parity suite at HEAD, GEMMs under background fabric load (benchmarks/gemm_under_load.py)
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $out/g_pytest.log 2>&1; echo "pytest rc=$?" >> $out/g_pytest.log
tail -6 $out/g_pytest.log
timeout -k 5 300 python benchmarks/gemm_under_load.py > $out/g_gemm_under_load.jsonl 2> $out/g_gemm_under_load.err; echo "rc=$?"
cat $out/g_gemm_under_load.jsonl; tail -3 $out/g_gemm_under_load.err
