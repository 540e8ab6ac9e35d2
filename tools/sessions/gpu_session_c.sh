#!/bin/bash
# GPU session C (round 3): parity suite, PMC of the attention kernels, overlap projection with the faster stand-in, GEMM chunk sweep
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests -m gpu -x -q > $out/c_pytest.log 2>&1; echo "pytest rc=$?" >> $out/c_pytest.log
tail -4 $out/c_pytest.log
timeout -k 5 300 python benchmarks/overlap_projection.py 40 > $out/c_overlap_projection.jsonl 2> $out/c_overlap_projection.err; echo "overlap rc=$?"
PMC_GROUPS="a b e" timeout -k 5 600 bash tools/pmc_profile.sh gpurun_out/c_pmc attn_fwd attn_fwd_nodrop attn_bwd > $out/c_pmc_attn.txt 2>&1
find $out/c_pmc -name "*.db" -delete 2>/dev/null
tail -3 $out/c_pmc_attn.txt
timeout -k 5 400 python benchmarks/ab_chunk.py > $out/c_ab_chunk.jsonl 2>&1
tail -3 $out/c_ab_chunk.jsonl
