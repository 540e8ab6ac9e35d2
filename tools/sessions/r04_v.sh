#!/bin/bash
# GPU session r04-v: two-k-tile look-ahead in the conv kernel-gradient body (-DNK_BWK_LOOKAHEAD2=1, benchmarks/_ab/la2.so) against the
# one-tile loop, same box: C3 kernel gradient (mixed and uniform launch), layer shapes, parity of the variant
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
NEURONIKA_HIP_LIB=$root/benchmarks/_ab/la2.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conv_fuzz.py -x -q -m gpu -k "conv" > $out/v_pytest.log 2>&1; echo "pytest(la2) rc=$?"; grep -E "passed|failed" $out/v_pytest.log | tail -2
{
for rep in 1 2 3; do
  for v in main la2; do
    lib=$root/benchmarks/_ab/$v.so; [ $v = main ] && lib=$root/neuronika_amd/lib/libneuronika_hip.so
    echo "rep$rep $v $(NEURONIKA_HIP_LIB=$lib timeout 120 python benchmarks/ab_conv_narrow.py 65 0 2>&1 | tail -1)"
  done
done
for v in main la2; do
  lib=$root/benchmarks/_ab/$v.so; [ $v = main ] && lib=$root/neuronika_amd/lib/libneuronika_hip.so
  echo "== conv_shapes $v"; NEURONIKA_HIP_LIB=$lib timeout 300 python benchmarks/conv_shapes.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d.get('layer', d.get('shape')), {k: v for k, v in d.items() if 'bwd_kernel' in k})"
done
} 2>&1 | tee $out/v_la2_ab.txt
