#!/bin/bash
# GPU session r04-o: loop-header alignment (-mllvm -align-loops=N): identical hot loops of the EPX / plain NT kernels differ by 1.6 %
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
{
for rep in 1 2 3; do
  for v in main align32 align64 align128 align256; do
    lib=$root/benchmarks/_ab/$v.so; [ $v = main ] && lib=$root/neuronika_amd/lib/libneuronika_hip.so
    echo "rep$rep $v NT $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 1 4096 4096 4096) NN $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 0 4096 4096 4096) TN $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 1 0 4096 4096 4096) NT-proj $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 1 32768 1024 1024) 2048NN $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 0 2048 2048 2048)"
  done
done
} 2>&1 | tee $out/o_align_loops.txt
