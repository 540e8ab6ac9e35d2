#!/bin/bash
# The round's closing profile set (run at the round's last kernel commit, passed as $1): parity suite, the driver's bench
# invocation, bench lines + rocprofv3 kernel stats of the four workloads, micro-benchmarks, traffic PMC (roofline.traffic).
#   bash tools/gpu_session_final.sh <commit> [tag]          (tag: rNN, default r04)
set -u
commit=${1:-unknown}
tag=${2:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests -m gpu -x -q > $out/z_pytest.log 2>&1; echo "pytest rc=$?" >> $out/z_pytest.log
grep -E "passed|failed" $out/z_pytest.log | tail -2
cp $out/tolerance_margins.json $out/${tag}_tolerance_margins.json
timeout -k 5 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver_invocation.json 2> $out/z_bench_driver.err; echo "bench rc=$?"
bash tools/profile_round.sh gpurun_out/${tag}z $tag > $out/z_profile_round.log 2>&1
bash tools/traffic_pmc.sh gpurun_out/${tag}z_traffic > $out/z_traffic.log 2>&1
python tools/make_roofline_traffic.py gpurun_out/${tag}z_traffic $commit > $out/${tag}_traffic_pmc.md 2>> $out/z_traffic.log
cp profiles/roofline_traffic.json $out/${tag}_roofline_traffic.json
find $out/${tag}z_traffic -name "*.db" -delete 2>/dev/null
ls $out/${tag}z | head -30
