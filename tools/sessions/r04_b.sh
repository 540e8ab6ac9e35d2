#!/bin/bash
# GPU session r04-b: K-blocked accumulation through a workspace slab (chains of 2048), Linear+ReLU epilogues, multi-parameter SGD:
# parity suite, same-box A/B of the fold (none / 64 / 32 k-tiles), the C4 step fused / node by node / without the fold, and
# a kernel trace of the fused C4 step.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $out/b_pytest.log 2>&1; echo "pytest rc=$?" >> $out/b_pytest.log
tail -25 $out/b_pytest.log
cp $out/tolerance_margins.json $out/b_tolerance_margins.json 2>/dev/null
{
for rep in 1 2 3; do
  for v in main nofold; do
    lib=$root/benchmarks/_ab/$v.so; [ $v = main ] && lib=$root/neuronika_amd/lib/libneuronika_hip.so
    for l in "0 1" "0 0" "1 0"; do
      echo "rep$rep $v layout($l) 4096^3: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py $l 4096 4096 4096)"
    done
    echo "rep$rep $v TN 1024x1024x32768: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 1 0 1024 1024 32768)  NT 8192^3: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 1 8192 8192 8192) NN 4096x4096x8192: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 0 4096 4096 8192)"
  done
done
} 2>&1 | tee $out/b_fold_ab.txt
cp neuronika_amd/lib/libneuronika_hip.so /tmp/main.so
line='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], d["gemm_share_of_step"], d["loss"])'
{
for rep in 1 2 3; do
  echo "rep$rep fused      $(NK_BENCH_NO_SUBRECORDS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"
  echo "rep$rep unfused    $(NK_BENCH_UNFUSED_RELU=1 NK_BENCH_NO_SUBRECORDS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"
  cp $root/benchmarks/_ab/nofold.so neuronika_amd/lib/libneuronika_hip.so
  echo "rep$rep fused-nofold $(NK_BENCH_NO_SUBRECORDS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"
  cp /tmp/main.so neuronika_amd/lib/libneuronika_hip.so
done
} 2>&1 | tee $out/b_c4_ab.txt
cd /tmp && export TMPDIR=/tmp
NK_BENCH_NO_SUBRECORDS=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $out/b_prof_mlp -o r -- python $root/bench.py --workload mlp --steps 10 --warmup 2 --no-cpu-baseline > $out/b_prof_mlp.log 2>&1
db=$(find $out/b_prof_mlp -name "*_results.db" | head -1)
[ -n "$db" ] && python $root/tools/rocpd_kernel_stats.py "$db" > $out/b_mlp_step_kernel_stats.md
find $out/b_prof_mlp -name "*.db" -delete
cat $out/b_mlp_step_kernel_stats.md | head -40
