#!/bin/bash
# Round 5, session e: the pipelined, persistent Winograd kernels (two V buffers, the next item's transform inside the MFMA loop):
# parity, same-box A/B, counters.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r05e; mkdir -p $out
cd $root
timeout -k 5 600 python -m pytest tests/test_gpu_winograd.py -q -x > $out/pytest_winograd.log 2>&1; echo "pytest winograd rc=$?" | tee -a $out/pytest_winograd.log
tail -25 $out/pytest_winograd.log
timeout -k 5 300 python benchmarks/ab_winograd.py > $out/ab_winograd.jsonl 2> $out/ab_winograd.err; echo "ab rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r05e/ab_winograd.jsonl"):
    r=json.loads(l); print(r["round"], r["pass"][:16], r["implicit_gemm_us"], r["winograd_us"], r["speedup"])
PY
tail -3 $out/ab_winograd.err
timeout -k 5 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_conv_fuzz.py -q -x -k "C3 or conv" > $out/pytest_conv.log 2>&1; echo "pytest conv rc=$?"; tail -4 $out/pytest_conv.log
NK_CONV_WINOGRAD=1 timeout -k 5 300 python bench.py --workload conv --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_conv.json 2> $out/bench_conv.err
python -c "
import json; r=json.load(open('$out/bench_conv.json')); print('conv step', r['ms_per_step'], 'ms', r['roofline']['frac'], r.get('conv_share_of_step'))"
PMC_GROUPS="a b" bash tools/pmc_profile.sh gpurun_out/r05e/pmc conv_fwd conv_bwd_input > $out/pmc_summary.txt 2>&1
grep -A 30 "## wino_kernel" $out/pmc_summary.txt | grep -E "##|GRBM_GUI|MFMA|INSTS_VALU|WAIT_INST_ANY|WAVE_CYCLES|ACTIVE_INST_VALU|LDS_BANK|BUSY_CYCLES"
find $out -name "*.db" -delete
