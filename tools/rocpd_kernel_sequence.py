#!/usr/bin/env python3
"""The LAST `n` kernel dispatches of a rocprofv3 (rocpd sqlite) kernel trace, in launch order, with duration and the gap to
the previous kernel's end - to see WHICH launch of a step is slow, not only the per-name average.

    python tools/rocpd_kernel_sequence.py x_results.db [n]
"""
import re
import sqlite3
import sys


def main(path, n):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()[-n:]
    prev = None
    print("| # | kernel | us | gap us |\n|---:|---|---:|---:|")
    for i, (name, s, e) in enumerate(rows):
        short = re.sub(r"^void\s+", "", re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "")))
        print(f"| {i} | `{short}` | {(e - s) / 1e3:.1f} | {((s - prev) / 1e3) if prev else 0:.1f} |")
        prev = e


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
