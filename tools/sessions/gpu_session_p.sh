#!/bin/bash
# GPU session P (round 3): conv kernel gradient with a narrow last column tile - parity, then same-box A/B (NK_CONV_NARROW=0 / 1)
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "conv" > $out/p_pytest.log 2>&1; echo "pytest rc=$?" >> $out/p_pytest.log
tail -6 $out/p_pytest.log
for rep in 1 2 3; do
  for v in 0 1; do
    NK_CONV_NARROW=$v timeout -k 5 200 python bench.py --workload conv --steps 100 --warmup 5 --no-cpu-baseline > $out/p_conv_$v.json 2> $out/p_conv_$v.err
    python -c "
import json; d=json.loads(open('$out/p_conv_$v.json').read().strip().splitlines()[-1]); print('narrow=$v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  NK_CONV_NARROW=$v NK_BENCH_NO_SUBRECORDS=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $out/p_prof_$v -o r -- python $root/bench.py --workload conv --steps 10 --warmup 2 --no-cpu-baseline > $out/p_prof_$v.log 2>&1
  db=$(find $out/p_prof_$v -name "*_results.db" | head -1)
  [ -n "$db" ] && python $root/tools/rocpd_kernel_stats.py "$db" | head -12 | cut -c1-140
  find $out/p_prof_$v -name "*.db" -delete
done
