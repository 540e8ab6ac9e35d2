#!/bin/bash
# GPU session r04-h: column-reduction change (parity + microbench), the launch sequence of one C4 step (which memset?), quick lines
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $out/h_pytest.log 2>&1; echo "pytest rc=$?" >> $out/h_pytest.log
tail -4 $out/h_pytest.log
timeout -k 5 300 python benchmarks/microbench.py stream > $out/h_microbench_stream.jsonl 2>&1; tail -24 $out/h_microbench_stream.jsonl
line='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], d.get("gemm_share_of_step"))'
for w in mlp mha conv; do echo "$w $(NK_BENCH_NO_SUBRECORDS=1 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$line")"; done
cd /tmp && export TMPDIR=/tmp
NK_BENCH_NO_SUBRECORDS=1 timeout -k 5 200 rocprofv3 --kernel-trace -d $out/h_prof_mlp -o r -- python $root/bench.py --workload mlp --steps 3 --warmup 1 --no-cpu-baseline > $out/h_prof_mlp.log 2>&1
db=$(find $out/h_prof_mlp -name "*_results.db" | head -1)
[ -n "$db" ] && python $root/tools/rocpd_kernel_sequence.py "$db" > $out/h_mlp_sequence.md 2>&1
find $out/h_prof_mlp -name "*.db" -delete
tail -60 $out/h_mlp_sequence.md
