#!/bin/bash
# GPU session r04-aa: the out-projection's parameter gradients between the attention backward kernel and its dK / dV products
# (nn::MultiheadAttention::interleave_out_gradients, nk_attention_qkv_bwd_part): tape + attention parity, C5 step on / off alternating
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 600 python -m pytest tests/test_gpu_tape.py tests/test_gpu_attention.py tests/test_gpu_fullsize.py -x -q -m gpu -k "mha or attention or C5 or sync or gradient_sync or replica" > $out/aa_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" $out/aa_pytest.log | tail -3
for rep in 1 2 3; do for v in 1 0; do
  echo "rep$rep interleave=$v $(NK_MHA_INTERLEAVE=$v timeout 200 python bench.py --workload mha --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"
done; done | tee $out/aa_c5_interleave.txt
