#!/bin/bash
# Round 6, session c: the Winograd tail cut (the ragged last round of tile blocks cut along the reduction channels): parity of the
# (the "cut" mode of benchmarks/ab_winograd.py this script calls lives in docs/r06_winograd_tail_cut.patch: measured, not kept)
# Winograd / conv / tape suites, then the A/B (cut off / on, stagger rule / off) at C3 and two deeper layers.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r06c; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_multi.py tests/test_gpu_tape.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout -k 5 300 python benchmarks/ab_winograd.py 128 cut > $out/ab_winograd_cut.jsonl 2> $out/ab.err
cat $out/ab_winograd_cut.jsonl
