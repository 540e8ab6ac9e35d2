#!/bin/bash
# GPU session r04-y: every schedule a stream-K decomposition of 1024^3 on 256 CUs degenerates to, forced (NK_GEMM_FORCE = ti,tj,splits):
# 128x128 tiles split 4 / 2 ways (64 tiles x 4 = one block per CU), 128x64 and 64x128 split 2 ways, 64x64 unsplit with and without k-pair
# blocks, 64x64 split 2 ways - against the rules (64x64 k-pair); TFLOP/s per layout, beta = 1
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
{
for f in "rules" "2,2,4" "2,2,2" "2,2,1" "2,1,2" "1,2,2" "2,1,1" "1,1,2" "1,1,1"; do
  for kp in "" 0; do
    [ "$f" != "1,1,1" ] && [ "$kp" = "0" ] && [ "$f" != "rules" ] && continue
    line="force=$f kpair=${kp:-rule}:"
    for l in "0 0" "0 1" "1 0"; do
      if [ "$f" = "rules" ]; then v=$(NK_GEMM_KPAIR=$kp timeout 60 python benchmarks/ab_force.py $l 1024 1024 1024 2>&1 | tail -1)
      else v=$(NK_GEMM_FORCE=$f NK_GEMM_KPAIR=$kp timeout 60 python benchmarks/ab_force.py $l 1024 1024 1024 2>&1 | tail -1); fi
      line="$line  ($l) $v"
    done
    echo "$line"
  done
done
} | tee $out/y_streamk_equivalents_1024.txt
