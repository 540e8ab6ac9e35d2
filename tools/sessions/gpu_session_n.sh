#!/bin/bash
# GPU session N (round 3): graph capture refuses mask-drawing forwards; the tape / parity files that touch dropout
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests/test_gpu_tape.py -m gpu -x -q -k "hipgraph or graph or dropout" > $out/n_pytest.log 2>&1; echo "pytest rc=$?" >> $out/n_pytest.log
tail -15 $out/n_pytest.log
