#!/bin/bash
# GPU session L (round 3): ragged sequence lengths on the fused attention core - parity, then fused vs node path at S = 1000
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests/test_gpu_attention.py -m gpu -x -q > $out/l_pytest_attn.log 2>&1; echo "pytest rc=$?" >> $out/l_pytest_attn.log
tail -15 $out/l_pytest_attn.log
timeout -k 5 600 python -m pytest tests/test_gpu_tape.py -m gpu -x -q -k "mha or attention" > $out/l_pytest_tape.log 2>&1; echo "pytest rc=$?" >> $out/l_pytest_tape.log
tail -5 $out/l_pytest_tape.log
timeout -k 5 300 python benchmarks/attention_core.py 32 1000 16 5 > $out/l_attn_1000.jsonl 2> $out/l_attn_1000.err; cat $out/l_attn_1000.jsonl; tail -2 $out/l_attn_1000.err
timeout -k 5 300 python benchmarks/attention_core.py 32 1024 16 5 > $out/l_attn_1024.jsonl 2> $out/l_attn_1024.err; cat $out/l_attn_1024.jsonl
