#!/bin/bash
# GPU session F (round 3): the round's profile set at commit 2779416 - parity suite, bench lines + rocprofv3 kernel stats of the four
# workloads, micro-benchmarks, traffic PMC (roofline.traffic), PMC of the attention kernels
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests -m gpu -x -q > $out/f_pytest.log 2>&1; echo "pytest rc=$?" >> $out/f_pytest.log
tail -3 $out/f_pytest.log
bash tools/profile_round.sh gpurun_out/r03 r03 > $out/f_profile_round.log 2>&1
bash tools/traffic_pmc.sh gpurun_out/r03_traffic > $out/f_traffic.log 2>&1
python tools/make_roofline_traffic.py gpurun_out/r03_traffic 2779416 > $out/r03_traffic_pmc.md 2>> $out/f_traffic.log
cp profiles/roofline_traffic.json $out/r03_roofline_traffic.json
find $out/r03_traffic -name "*.db" -delete 2>/dev/null
PMC_GROUPS="a b e" timeout -k 5 600 bash tools/pmc_profile.sh gpurun_out/r03_pmc attn_fwd attn_fwd_nodrop attn_bwd conv_bwd_input gemm4k > $out/r03_pmc_raw.txt 2>&1
find $out/r03_pmc -name "*.db" -delete 2>/dev/null
ls $out/r03
