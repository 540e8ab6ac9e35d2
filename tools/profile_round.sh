#!/bin/bash
# Round profile set: bench lines of the four workloads, rocprofv3 --kernel-trace of each (summarised by
# tools/rocpd_kernel_stats.py), micro-benchmarks.   bash tools/profile_round.sh OUTDIR TAG     (on the GPU box)
set -u
out=$1; tag=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
for w in mlp matmul conv mha; do
  timeout -k 5 300 python "$root/bench.py" --workload $w > "$root/$out/${tag}_bench_$w.json" 2> "$root/$out/${tag}_bench_$w.err"
  # (the traced run is the workload alone: the default line's C2 / C3 / C5 sub-records would mix their kernels into the C4 table)
  NK_BENCH_NO_SUBRECORDS=2 timeout -k 5 200 rocprofv3 --kernel-trace --stats -d "$root/$out/prof_$w" -o r -- python "$root/bench.py" --workload $w --steps 10 --warmup 2 --no-cpu-baseline > "$root/$out/prof_$w.log" 2>&1
  db=$(find "$root/$out/prof_$w" -name "*_results.db" | head -1)
  [ -n "$db" ] && python "$root/tools/rocpd_kernel_stats.py" "$db" > "$root/$out/${tag}_${w}_step_kernel_stats.md"
  [ -n "$db" ] && python "$root/tools/rocpd_kernel_sequence.py" "$db" 90 > "$root/$out/${tag}_${w}_step_sequence.md"   # the last launches in order, with gaps
done
timeout -k 5 300 python "$root/benchmarks/microbench.py" > "$root/$out/${tag}_microbench.jsonl" 2>&1
timeout -k 5 300 python "$root/benchmarks/conv_shapes.py" > "$root/$out/${tag}_conv_shapes.jsonl" 2>&1
find "$root/$out" -name "*.db" -delete
