#!/bin/bash
# GPU session r04-a: (1) which k of an MFMA's pair is added first (pins oracle/device_order_sgemm.c), (2) the GPU suite with the
# K-blocked accumulation, the non-finite test and the mask-separated C4 test, (3) same-box A/B of the fold length at 4096^3 and
# on the C4 step.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
python - > $out/a_pair_order.txt 2>&1 <<'P'
import os, numpy as np
from neuronika_amd import capi as c
from oracle.build_c import sgemm_device_order as model
dev = c.Device(0)
rng = np.random.default_rng(0)
for K in (8, 64, 1024, 2080):
    a = (rng.random((128, K), dtype=np.float32) - 0.5); b = (rng.random((K, 128), dtype=np.float32) - 0.5)
    os.environ["NK_GEMM_FORCE"] = "2,2,1"
    A, B, C = dev.array(a), dev.array(b), dev.zeros((128, 128))
    c.sgemm(dev, 0, 0, 128, 128, K, 1.0, A, K, B, 128, 0.0, C, 128)
    got = C.numpy()
    print(K, "k-first", np.array_equal(got, model(a, b, 2048, False)), "k+4-first", np.array_equal(got, model(a, b, 2048, True)),
          "maxdiff", float(np.abs(got - model(a, b, 2048, False)).max()))
P
cat $out/a_pair_order.txt
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $out/a_pytest.log 2>&1; echo "pytest rc=$?" >> $out/a_pytest.log
tail -15 $out/a_pytest.log
cp $out/tolerance_margins.json $out/a_tolerance_margins.json 2>/dev/null
{
for rep in 1 2 3; do
  for v in main nofold kfold16 kfold8; do
    lib=$root/benchmarks/_ab/$v.so; [ $v = main ] && lib=$root/neuronika_amd/lib/libneuronika_hip.so
    for l in "0 1" "0 0" "1 0"; do
      echo "rep$rep $v layout($l) 4096^3: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py $l 4096 4096 4096)"
    done
    echo "rep$rep $v NT 32768x1024x1024: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 1 32768 1024 1024)  TN 1024x1024x32768: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 1 0 1024 1024 32768)  NN 2048^3: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 0 2048 2048 2048) NT 8192^3: $(NEURONIKA_HIP_LIB=$lib python benchmarks/ab_force.py 0 1 8192 8192 8192)"
  done
done
} 2>&1 | tee $out/a_fold_ab.txt
# C4 step with each library in place of the product one (the tape links it by rpath)
cp neuronika_amd/lib/libneuronika_hip.so /tmp/main.so
{
for rep in 1 2; do
  for v in main nofold kfold8; do
    src=/tmp/main.so; [ $v != main ] && src=$root/benchmarks/_ab/$v.so
    cp $src neuronika_amd/lib/libneuronika_hip.so
    echo "rep$rep $v $(NK_BENCH_NO_SUBRECORDS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], d["matmul_4096"]["value"] if "matmul_4096" in d else "")')"
  done
done
cp /tmp/main.so neuronika_amd/lib/libneuronika_hip.so
} 2>&1 | tee $out/a_c4_ab.txt
