#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in one or more rocprofv3 rocpd databases.

    python tools/rocpd_pmc_summary.py gpurun_out/pmc/*_results.db
"""
import re
import sqlite3
import sys
from collections import defaultdict


def main(paths):
    out = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [values per dispatch]
    dur = defaultdict(list)
    for path in paths:
        c = sqlite3.connect(path)
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        q = "select kernel_name, counter_name, value, dispatch_id, start, end from counters_collection" \
            if "kernel_name" in cols else None
        if q is None:
            print("columns:", cols); continue
        acc = defaultdict(float)
        meta = {}
        for kn, cn, v, did, s, e in c.execute(q):
            acc[(did, cn)] += v
            meta[did] = (kn, e - s)
        for (did, cn), v in acc.items():
            kn, d = meta[did]
            kn = re.sub(r"\(.*$", "", kn.replace("(anonymous namespace)::", "")).replace("void ", "")
            out[kn][cn].append(v)
        for did, (kn, d) in meta.items():
            kn = re.sub(r"\(.*$", "", kn.replace("(anonymous namespace)::", "")).replace("void ", "")
            dur[kn].append(d)
    for kn in sorted(out):
        print(f"## {kn}  (dispatches: {len(dur[kn])}, avg duration under PMC {sum(dur[kn]) / len(dur[kn]) / 1e3:.1f} us)")
        for cn, vals in sorted(out[kn].items()):
            print(f"  {cn:32s} avg {sum(vals) / len(vals):16.1f}   n={len(vals)}")


if __name__ == "__main__":
    main(sys.argv[1:])
