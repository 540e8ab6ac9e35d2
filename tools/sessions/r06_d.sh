#!/bin/bash
# Round 6, session d: (1) benchmarks/native/rmw_stream.hip - what bounds the 3-read + 1-write streaming kernels; (2) the whole GPU suite;
# (3) the driver-shaped record with the exchange forced through a one-rank RCCL communicator; (4) the default driver line.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r06d; mkdir -p $out
cd $root
(cd benchmarks/native && hipcc --offload-arch=gfx950 -O3 -w -o /tmp/rmw_stream rmw_stream.hip) && timeout -k 5 300 /tmp/rmw_stream > $out/rmw_stream.jsonl 2> $out/rmw_stream.err
cat $out/rmw_stream.jsonl | cut -c1-260
timeout -k 5 900 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
cp gpurun_out/tolerance_margins.json $out/
NK_BENCH_FORCE_RCCL=1 timeout -k 5 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_forced_rccl.json 2> $out/bench_forced_rccl.err; echo "forced rccl rc=$?"
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
d = json.loads(open("gpurun_out/r06d/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"])
print("conv", d["conv_c3"]["ms_per_step"], {k: d["conv_c3"]["roofline"].get(k) for k in ("frac", "achieved", "algorithmic_speedup", "direct_equivalent_frac_of_peak")})
print("mha", d["mha_c5"]["ms_per_step"])
h = d["hbm_kernels"]
print(h["ceilings"])
for k, v in h["kernels"].items(): print(k, v["achieved"], v["frac"], v["frac_of_stream_ceiling_this_run"])
print("256MB:", {k: v["achieved"] for k, v in h["cache_assisted_256MB"]["kernels"].items()}, h["cache_assisted_256MB"]["ceilings"])
P
