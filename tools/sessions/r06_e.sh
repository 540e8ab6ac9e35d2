#!/bin/bash
# Round 6, session e: the span walk in the library's streaming kernels (nk_span_walk): parity suite, then the stream rows of microbench
# and the default driver line (hbm_kernels at 1 GiB, C4 / C5 step times).
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r06e; mkdir -p $out
cd $root
timeout -k 5 900 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
timeout -k 5 300 python benchmarks/microbench.py stream > $out/microbench_stream.jsonl 2>&1
timeout -k 5 300 python benchmarks/microbench.py softmax > $out/microbench_softmax.jsonl 2>&1
python - <<'P'
import json
for f in ("microbench_stream", "microbench_softmax"):
    for l in open("gpurun_out/r06e/%s.jsonl" % f):
        try: d = json.loads(l)
        except Exception: continue
        print("%-28s %-42s %7.1f GB/s  %.4f ms" % (d.get("kernel"), d.get("size", ""), d.get("gbps", 0), d.get("ms", 0)))
P
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
d = json.loads(open("gpurun_out/r06e/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"])
print("conv", d["conv_c3"]["ms_per_step"], "mha", d["mha_c5"]["ms_per_step"])
h = d["hbm_kernels"]
print(h["ceilings"])
for k, v in h["kernels"].items(): print(k, v["achieved"], v["frac"], v["frac_of_stream_ceiling_this_run"])
print("256MB:", {k: v["achieved"] for k, v in h["cache_assisted_256MB"]["kernels"].items()}, h["cache_assisted_256MB"]["ceilings"])
P
