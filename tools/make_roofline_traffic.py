#!/usr/bin/env python3
"""profiles/roofline_traffic.json (+ a markdown table) from the rocprofv3 PMC databases written by tools/traffic_pmc.sh.

    python tools/make_roofline_traffic.py gpurun_out/traffic [commit-ish]

bytes per launch = 2 x FETCH_SIZE (KB; gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md "HBM") +
WRITE_SIZE (KB), each averaged over the dispatches of the kernel family in its own pass."""
import glob
import json
import os
import re
import sqlite3
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha16():
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "neuronika_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def per_kernel(db):
    """kernel name -> [counter value summed over XCDs, per dispatch]"""
    c = sqlite3.connect(db)
    acc, meta = defaultdict(float), {}
    for kn, cn, v, did in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        acc[did] += v
        meta[did] = re.sub(r"\(.*$", "", kn.replace("(anonymous namespace)::", "")).replace("void ", "")
    out = defaultdict(list)
    for did, v in acc.items():
        out[meta[did]].append(v)
    return out


def main():
    src = sys.argv[1]
    commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    rows = {}
    for tgt in ("gemm_once", "conv_step_once", "mha_step_once"):
        f = glob.glob(os.path.join(src, tgt, "FETCH_SIZE", "**", "*_results.db"), recursive=True)
        w = glob.glob(os.path.join(src, tgt, "WRITE_SIZE", "**", "*_results.db"), recursive=True)
        if not f or not w:
            print("missing passes for", tgt, file=sys.stderr)
            continue
        fk, wk = per_kernel(f[0]), per_kernel(w[0])
        for k in fk:
            if k in wk:
                rows[(tgt, k)] = (len(fk[k]), 2048.0 * sum(fk[k]) / len(fk[k]), 1024.0 * sum(wk[k]) / len(wk[k]))

    def family(tgt, pred):
        sel = [(n, f, w) for (t, k), (n, f, w) in rows.items() if t == tgt and pred(k)]
        n = sum(s[0] for s in sel)
        return int(sum(s[0] * (s[1] + s[2]) for s in sel) / n) if n else None

    traffic = {
        "_note": "HBM/fabric bytes per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in KB, separate runs; FETCH doubled "
                 "per MI355X_MICROARCH.md's gfx950 wide-read correction), launch-weighted means. sgemm_kernel: the NT / NN / TN launches of "
                 "benchmarks/gemm_once.py - 4096^3 products, since round 6 each as two chained launches over K (4096 x 4096 x 2048 per "
                 "launch: algorithmic 134 - 201 MB); conv: the three passes of the C3 module step (benchmarks/conv_step_once.py); "
                 "mha_gemm: every sgemm_kernel launch of one C5 step (benchmarks/mha_step_once.py), attention: the forward and backward "
                 "launches of the fused attention core in the same step. Regenerate after a GEMM / conv "
                 "change: bash tools/traffic_pmc.sh DIR && python tools/make_roofline_traffic.py DIR",
        "_measured_at_commit": commit,
        # sha256 over the kernel sources the measured library was built from: bench.py hashes the sources it runs with and says
        # in its line whether `roofline.traffic` still belongs to them
        "_kernel_sources_sha16": kernel_sources_sha16(),
        "sgemm_kernel": family("gemm_once", lambda k: k.startswith(("sgemm_kernel", "sgemm_pair_kernel"))),
        "conv": family("conv_step_once", lambda k: k.startswith(("conv_fwd_fast", "conv_bwd_input_fast", "conv_bwd_kernel_kernel", "conv_bwd_kernel_mixed_kernel", "wino_kernel", "wino_dw_kernel"))),
        "mha_gemm": family("mha_step_once", lambda k: k.startswith(("sgemm_kernel", "sgemm_pair_kernel"))),
        "attention": family("mha_step_once", lambda k: k.startswith("attention_kernel")),
    }
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "roofline_traffic.json"), "w"), indent=1)
    print("| run | kernel | dispatches | fetched (x2) MB | written MB |\n|---|---|---:|---:|---:|")
    for (t, k), (n, f, w) in sorted(rows.items(), key=lambda kv: -kv[1][0] * (kv[1][1] + kv[1][2])):
        if (f + w) * n > 5e7:
            print(f"| {t} | `{k}` | {n} | {f / 1e6:.1f} | {w / 1e6:.1f} |")
    print("\n```json\n" + json.dumps(traffic, indent=1) + "\n```")


if __name__ == "__main__":
    main()
