#!/bin/bash
# The first thing to run when a multi-GPU MI355X node is available (no round of this build has had one): the 2-rank parity
# tests of the product path, then the weak-scaling bench at 2 / 4 / 8 GPUs, everything a post-mortem needs in one tarball.
#     bash tools/first_multi_gpu.sh [OUTDIR]          (about 5 minutes on an 8-GPU node)
# bench.py --gpus N spawns its own N ranks (one process per GPU, RCCL over xGMI, rendezvous on 127.0.0.1) and prints one JSON
# line per run with per-rank device times, the stand-alone all-reduce, exposed communication and what RCCL chose.
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=${1:-$root/gpurun_out/first_multi_gpu}
mkdir -p "$out"
cd "$root"
export HSA_ENABLE_IPC_MODE_LEGACY=0
ngpu=$(python -c "from neuronika_amd import capi; print(capi.device_count())")
echo "GPUs visible: $ngpu" | tee "$out/summary.txt"
rocm-smi --showtopo > "$out/topology.txt" 2>&1
python -c "import __graft_entry__ as g; g.build()" > "$out/build.log" 2>&1 || { echo "build failed" | tee -a "$out/summary.txt"; }
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -x -q > "$out/pytest_multi.log" 2>&1
echo "pytest tests/test_gpu_multi.py rc=$?  $(tail -1 "$out/pytest_multi.log")" | tee -a "$out/summary.txt"
for n in 1 2 4 8; do
  [ "$n" -gt "$ngpu" ] && break
  NCCL_DEBUG=INFO NCCL_DEBUG_FILE="$out/rccl_n${n}_%h_%p.log" timeout -k 10 600 python bench.py --gpus $n --steps 50 --warmup 5 \
      > "$out/bench_n$n.json" 2> "$out/bench_n$n.err"
  echo "bench --gpus $n rc=$?" | tee -a "$out/summary.txt"
  python - "$out/bench_n$n.json" <<'P' | tee -a "$out/summary.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ", d["n_gpus"], "GPUs:", d["value"], d["unit"], d["ms_per_step"], "ms/step, exposed comm", d.get("exposed_comm_ms"),
          "ms, all-reduce alone", (d.get("allreduce_alone") or {}).get("ms"), "ms", (d.get("allreduce_alone") or {}).get("algbw_GBps"), "GB/s")
except Exception as e:
    print("    no record:", e)
P
done
# How many RCCL channels, and does the shared-chip GEMM schedule pay?  At the largest N: channel caps 4 / 8 / 16 (the default run
# above is 16 told), each with the backward GEMMs told that many busy slots and - same cap - planning for an idle chip.
nmax=$ngpu; [ "$nmax" -gt 8 ] && nmax=8
if [ "$nmax" -gt 1 ]; then
  for ch in 4 8 16; do
    for told in 1 0; do
      tag="n${nmax}_ch${ch}_told${told}"
      if [ "$told" = 1 ]; then flag="--dp-channels $ch"; env_ch=""; else flag="--dp-channels 0"; env_ch="NCCL_MAX_NCHANNELS=$ch"; fi
      env $env_ch timeout -k 10 600 python bench.py --gpus $nmax --steps 50 --warmup 5 --no-cpu-baseline $flag \
          > "$out/bench_$tag.json" 2> "$out/bench_$tag.err"
      python - "$out/bench_$tag.json" "$tag" <<'P' | tee -a "$out/summary.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ", sys.argv[2], d["ms_per_step"], "ms/step, exposed comm", d.get("exposed_comm_ms"), "ms, GEMM slowdown", (d.get("gemm_contention") or {}).get("slowdown"), d.get("dp_channels"))
except Exception as e:
    print("   ", sys.argv[2], "no record:", e)
P
    done
  done
fi
# The curve in the driver's SCALE_rNN.json shape (one object, one entry per N): absolute samples/s, ms per step, weak-scaling efficiency
# against the N = 1 run of this same session, the ranks RCCL really ran with, the stand-alone all-reduce (algorithm and bus bandwidth:
# busbw = algbw * 2 (p - 1) / p) and the exposed communication per step.
python - "$out" <<'P' | tee -a "$out/summary.txt"
import json, os, sys
runs = {}
for n in (1, 2, 4, 8):
    try:
        runs[n] = json.loads(open(os.path.join(sys.argv[1], f"bench_n{n}.json")).read().strip().splitlines()[-1])
    except Exception:
        pass
scale = {"metric": None, "unit": None, "scaling": "weak", "measured_on": "tools/first_multi_gpu.sh", "runs": []}
for n, d in sorted(runs.items()):
    ar = d.get("allreduce_alone") or {}
    alg = ar.get("algbw_GBps")
    row = {"n_gpus": d.get("n_gpus", n), "value": d.get("value"), "ms_per_step": d.get("ms_per_step"), "steps": d.get("steps"),
           "efficiency_vs_n1": round(d["value"] / (n * runs[1]["value"]), 4) if 1 in runs and d.get("value") else None,
           "rccl_ranks": (d.get("rccl") or {}).get("ranks", d.get("rccl_ranks", n if n > 1 else None)),
           "exposed_comm_ms": d.get("exposed_comm_ms"), "allreduce_alone_ms": ar.get("ms"), "allreduce_algbw_GBps": alg,
           "allreduce_busbw_GBps": round(alg * 2 * (n - 1) / n, 1) if alg and n > 1 else None,
           "dp_channels": d.get("dp_channels"), "gemm_slowdown_under_exchange": (d.get("gemm_contention") or {}).get("slowdown")}
    scale["metric"], scale["unit"] = d.get("metric"), d.get("unit")
    scale["runs"].append(row)
    if row["efficiency_vs_n1"] is not None:
        print(f"weak-scaling efficiency at {n}: {row['efficiency_vs_n1']:.3f}  ({d['value']} {d.get('unit')}, busbw {row['allreduce_busbw_GBps']} GB/s)")
json.dump(scale, open(os.path.join(sys.argv[1], "SCALE_first_multi_gpu.json"), "w"), indent=1)
print("wrote", os.path.join(sys.argv[1], "SCALE_first_multi_gpu.json"))
P
tar czf "$out.tgz" -C "$(dirname "$out")" "$(basename "$out")" && echo "wrote $out.tgz"
