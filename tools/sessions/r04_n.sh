#!/bin/bash
# GPU session r04-n: LLVM scheduler strategies for the whole library (hipcc -mllvm ...): 4096^3 GEMMs, then the three steps
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
cp neuronika_amd/lib/libneuronika_hip.so /tmp/main.so
line='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'
{
for rep in 1 2; do
  for v in main maxilp trackers nohirp maxmem; do
    lib=$root/benchmarks/_ab/$v.so; [ $v = main ] && lib=/tmp/main.so
    g="NT $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 1 4096 4096 4096) NN $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 0 4096 4096 4096) TN $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 1 0 4096 4096 4096)"
    cp $lib neuronika_amd/lib/libneuronika_hip.so
    s=""
    for w in mlp conv mha; do s="$s $w $(NK_BENCH_NO_SUBRECORDS=2 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"; done
    echo "rep$rep $v $g |$s"
  done
done
cp /tmp/main.so neuronika_amd/lib/libneuronika_hip.so
} 2>&1 | tee $out/n_sched_flags.txt
