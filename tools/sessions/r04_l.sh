#!/bin/bash
# GPU session r04-l: the epilogue's relu / mask code as a template parameter: plain launches against the round-3 epilogue variant,
# the parity suite, the C4 / C5 steps.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $out/l_pytest.log 2>&1; echo "pytest rc=$?" >> $out/l_pytest.log
tail -4 $out/l_pytest.log
{
for rep in 1 2 3; do
  for v in main oldepi; do
    lib=$root/benchmarks/_ab/$v.so; [ $v = main ] && lib=$root/neuronika_amd/lib/libneuronika_hip.so
    echo "rep$rep $v NT $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 1 4096 4096 4096) NN $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 0 4096 4096 4096) TN $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 1 0 4096 4096 4096)"
  done
done
} 2>&1 | tee $out/l_epilogue_ab2.txt
line='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], d.get("gemm_share_of_step"))'
for rep in 1 2; do for w in mlp mha; do echo "$w $(NK_BENCH_NO_SUBRECORDS=1 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"; done; done 2>&1 | tee $out/l_steps.txt
