"""Times the three GEMM layouts of the MLP step (4096^3) and two conv passes with whatever library
NEURONIKA_HIP_LIB points at; run it alternately for two builds on the same box (see ab_build.py)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import timeit, rand  # noqa: E402

dev = c.Device(0)
n = int(os.environ.get("AB_N", "4096"))
A, B, C = rand(dev, (n, n), 0, 0, 1), rand(dev, (n, n), 1, 0, 1), dev.zeros((n, n))
out = {"lib": os.path.basename(c.LIB_PATH)}
fns = {"NT": lambda: c.sgemm(dev, 0, 1, n, n, n, 1.0, A, n, B, n, 0.0, C, n),
       "NN": lambda: c.sgemm(dev, 0, 0, n, n, n, 1.0, A, n, B, n, 0.0, C, n),
       "TN": lambda: c.sgemm(dev, 1, 0, n, n, n, 1.0, A, n, B, n, 0.0, C, n)}
timeit(dev, fns["NT"], 80)
for k, f in fns.items():
    timeit(dev, f, 20)
    ms = timeit(dev, f, 40)
    out[k] = round(2 * n ** 3 / ms / 1e9, 1)
if os.environ.get("AB_CONV", "1") == "1":
    x = rand(dev, (128, 64, 58, 58), 2, 0, 1); w = rand(dev, (128, 64, 3, 3), 3, -1, 1)
    y = dev.zeros((128, 128, 56, 56)); g = rand(dev, (128, 128, 56, 56), 4, 0, 1)
    dx = dev.zeros(x.shape); dw = dev.zeros(w.shape)
    flop = 2 * 128 * 128 * 56 * 56 * 64 * 9
    for k, f in {"conv_fwd": lambda: c.conv_fwd(dev, x, w, y, (1, 1), (1, 1)),
                 "conv_bwd_in": lambda: c.conv_bwd_input(dev, dx, g, w, (1, 1), (1, 1)),
                 "conv_bwd_k": lambda: c.conv_bwd_kernel(dev, dw, g, x, (1, 1), (1, 1))}.items():
        timeit(dev, f, 10)
        out[k] = round(flop / timeit(dev, f, 20) / 1e9, 1)
print(json.dumps(out))
