#!/bin/bash
# GPU session r04-u: conv small launches (destination-order weight re-ordering, tail reductions with every load in flight): conv parity
# suites + the C3 step traced
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conv_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_determinism.py tests/test_gpu_tape.py -x -q -m gpu -k "conv or Conv or C3 or determin" > $out/u_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/u_pytest.log
grep -E "passed|failed|Error" $out/u_pytest.log | tail -3
cd /tmp && export TMPDIR=/tmp
NK_BENCH_NO_SUBRECORDS=2 timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $out/u_prof_conv -o r -- python $root/bench.py --workload conv --steps 10 --warmup 2 --no-cpu-baseline > $out/u_prof_conv.log 2>&1
db=$(find $out/u_prof_conv -name "*_results.db" | head -1)
python $root/tools/rocpd_kernel_stats.py $db | tee $out/u_conv_step_kernel_stats.md
find $out/u_prof_conv -name "*.db" -delete
cd $root; timeout 200 python bench.py --workload conv --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
