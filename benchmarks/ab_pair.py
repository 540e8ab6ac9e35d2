"""nk_mm_bwd / nk_mm_t_bwd (two products, one launch: sgemm_pair_kernel) against two launches, same box, per size.

    python benchmarks/ab_pair.py [1024 2048 4096]

Per size and node (mm: NT + TN, mm_t: NN + TN): microseconds of the backward pair under NK_TUNE_GEMM_PAIR = 0 (two launches,
the k-pair rule as it is), 1 (one launch, k-pair blocks where both plans have them and the CUs can hold two), 2 (one launch,
256-thread blocks) and -1 (the rule); `same_bits` = the one-launch results equal the two-launch results under the matching
k-pair setting, bit for bit."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import timeit, rand  # noqa: E402

dev = c.Device(0)
sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096]
for n in sizes:
    A, B, G = rand(dev, (n, n), 0, 0, 1), rand(dev, (n, n), 1, 0, 1), rand(dev, (n, n), 2, 0, 1)
    dA, dB = dev.zeros((n, n)), dev.zeros((n, n))
    for node, fn in (("mm", c.mm_bwd), ("mm_t", c.mm_t_bwd)):
        rec = {"n": n, "node": node, "lib": os.path.basename(c.LIB_PATH)}
        ref = {}
        for kp in (None, 0):  # reference bits: two launches under the k-pair rule / without k-pair blocks
            dev.gemm_pair(0); dev.gemm_kpair(kp)
            fn(dev, dA, dB, G, A, B, True, True)
            ref[kp] = (dA.numpy().copy(), dB.numpy().copy())
        dev.gemm_kpair(None)
        for mode in (0, 1, 2, -1, 0):
            dev.gemm_pair(mode)
            us = timeit(dev, lambda: fn(dev, dA, dB, G, A, B, True, True), 50) * 1e3
            key = f"pair={mode}"
            rec[key if key not in rec else key + " again"] = round(us, 2)
            if mode in (1, 2):
                fn(dev, dA, dB, G, A, B, True, True)
                a, b = dA.numpy(), dB.numpy()
                rec[f"same_bits {mode}"] = [bool(np.array_equal(a, ref[k][0]) and np.array_equal(b, ref[k][1])) for k in (None, 0)]
        dev.gemm_pair(None)
        rec["tflops two / rule"] = [round(4 * n ** 3 / rec["pair=0"] / 1e6, 1), round(4 * n ** 3 / rec["pair=-1"] / 1e6, 1)]
        print(json.dumps(rec), flush=True)
