#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table
(name, calls, total / average / min / max duration, share of GPU time) — the same content as
rocprofv3's `--stats` CSV, produced from the `*_results.db` this ROCm version writes.

    python tools/rocpd_kernel_stats.py gpurun_out/prof/x_results.db > profiles/rNN_xxx_kernel_stats.md
"""
import re
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        short = name.replace("(anonymous namespace)::", "")
        short = re.sub(r"\(.*$", "", short)
        short = re.sub(r"^void\s+", "", short)
        a = agg.setdefault(short, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of `{path}`\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % of GPU time |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} | {100 * a[1] / total:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
