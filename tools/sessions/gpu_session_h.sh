#!/bin/bash
# GPU session H (round 3): validation of the round's last build - parity suite, the driver's bench invocation, the C4 kernel table
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests -m gpu -x -q > $out/h_pytest.log 2>&1; echo "pytest rc=$?" >> $out/h_pytest.log
grep -E "passed|failed" $out/h_pytest.log | tail -2
cp $out/tolerance_margins.json $out/h_tolerance_margins.json
timeout -k 5 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/h_bench_driver_invocation.json 2> $out/h_bench_driver_invocation.err; echo "bench rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" > $out/h_smoke.log 2>&1; tail -1 $out/h_smoke.log
cd /tmp && export TMPDIR=/tmp
NK_BENCH_NO_SUBRECORDS=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $out/h_prof_mlp -o r -- python $root/bench.py --workload mlp --steps 10 --warmup 2 --no-cpu-baseline > $out/h_prof_mlp.log 2>&1
db=$(find $out/h_prof_mlp -name "*_results.db" | head -1)
[ -n "$db" ] && python $root/tools/rocpd_kernel_stats.py "$db" > $out/r03_mlp_step_kernel_stats.md
find $out/h_prof_mlp -name "*.db" -delete
head -12 $out/r03_mlp_step_kernel_stats.md | cut -c1-150
