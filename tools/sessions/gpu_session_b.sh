#!/bin/bash
# GPU session B (round 3): parity suite on the new mask layout, overlap projection, attention A/B + occupancy, PMC of the attention kernels
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests -m gpu -x -q > $out/b_pytest.log 2>&1; echo "pytest rc=$?" >> $out/b_pytest.log
tail -4 $out/b_pytest.log
timeout -k 5 300 python benchmarks/overlap_projection.py 40 > $out/b_overlap_projection.jsonl 2> $out/b_overlap_projection.err; echo "overlap rc=$?"
for i in 1 2; do
  NEURONIKA_HIP_LIB=$root/benchmarks/_ab/r02.so timeout -k 5 120 python benchmarks/attention_core.py 32 1024 16 10 > $out/b_attn_r02_$i.jsonl 2>&1
  timeout -k 5 120 python benchmarks/attention_core.py 32 1024 16 10 > $out/b_attn_new_$i.jsonl 2>&1
  NK_ATTN_OCC=2 timeout -k 5 120 python benchmarks/attention_core.py 32 1024 16 10 > $out/b_attn_new_occ2_$i.jsonl 2>&1
done
grep -h -c . $out/b_attn_*.jsonl > /dev/null
PMC_GROUPS="a b e" timeout -k 5 600 bash tools/pmc_profile.sh gpurun_out/b_pmc attn_fwd attn_fwd_nodrop attn_bwd > $out/b_pmc_attn.txt 2>&1
find $out/b_pmc -name "*.db" -delete 2>/dev/null
tail -3 $out/b_pmc_attn.txt
