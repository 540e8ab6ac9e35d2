#!/bin/bash
# GPU session S (round 3): stride-2 variant of the fast conv forward - parity of the conv suite, then the strided layer shapes
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_tape.py -m gpu -x -q -k "conv" > $out/s_pytest.log 2>&1; echo "pytest rc=$?" >> $out/s_pytest.log
tail -4 $out/s_pytest.log
for sh in "s2" "C3"; do timeout -k 5 300 python benchmarks/conv_shapes.py "$sh"; done 2>&1 | tee $out/s_conv_shapes.jsonl
