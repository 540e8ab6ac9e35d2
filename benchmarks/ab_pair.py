"""nk_mm_bwd / nk_mm_t_bwd (two products, one launch: sgemm_pair_kernel) against two launches, same box, per size.

    python benchmarks/ab_pair.py [1024 2048 4096]

Per size and node (mm: NT + TN, mm_t: NN + TN): microseconds of the backward pair under NK_TUNE_GEMM_PAIR = 0 (two launches,
the k-pair rule as it is), 1 (one launch whenever eligible) and -1 (the rule); `same_bits` = the one-launch results equal the
two-launch results [under the k-pair rule, without k-pair blocks], bit for bit.  `dkdv`: the attention backward's two
batched TN products at the C5 geometry (B = 32, H = 16, S = 1024, dh = 64, packed projection layout)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import timeit, rand  # noqa: E402

dev = c.Device(0)
sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096]
assert all(0 < n <= 16384 for n in sizes), "sizes are matrix extents (N of N^3), at most 16384"   # (a typo here once cost 14 GPU-minutes)
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
        for mode in (0, 1, -1, 0):
            dev.gemm_pair(mode)
            us = timeit(dev, lambda: fn(dev, dA, dB, G, A, B, True, True), 50) * 1e3
            key = f"pair={mode}"
            rec[key if key not in rec else key + " again"] = round(us, 2)
            if mode == 1:
                fn(dev, dA, dB, G, A, B, True, True)
                a, b = dA.numpy(), dB.numpy()
                rec[f"same_bits {mode}"] = [bool(np.array_equal(a, ref[k][0]) and np.array_equal(b, ref[k][1])) for k in (None, 0)]
        dev.gemm_pair(None)
        rec["tflops two / rule"] = [round(4 * n ** 3 / rec["pair=0"] / 1e6, 1), round(4 * n ** 3 / rec["pair=-1"] / 1e6, 1)]
        print(json.dumps(rec), flush=True)

if os.environ.get("AB_DKDV", "1") == "1":
    Bn, H, S, dh = 32, 16, 1024, 64
    d, ld = H * dh, 3 * H * dh
    DS, PD = dev.zeros((Bn * H, S, S)), dev.zeros((Bn * H, S, S))
    c.lib.nk_fill(dev.h, DS.p, DS.size, 0.001); c.lib.nk_fill(dev.h, PD.p, PD.size, 0.002)
    Q, DO, GR = rand(dev, (Bn * S, ld), 5), rand(dev, (Bn * S, d), 6), dev.zeros((Bn * S, ld))
    dK, dV = GR.view_offset(d), GR.view_offset(2 * d)
    so, sq, po, pi = S * d, S * ld, H * S * S, S * S
    p0 = (1, 0, S, dh, S, DS, S, po, pi, Q, ld, sq, dh, 0.0, dK, ld, sq, dh)
    p1 = (1, 0, S, dh, S, PD, S, po, pi, DO, d, so, dh, 0.0, dV, ld, sq, dh)
    rec = {"case": "dkdv C5"}
    for mode in (0, 1, -1, 0, 1):
        dev.gemm_pair(mode)
        us = timeit(dev, lambda: c.sgemm_pair_batched(dev, Bn, H, p0, p1), 20) * 1e3
        key = f"pair={mode}"
        rec[key if key not in rec else key + " again"] = round(us, 1)
    dev.gemm_pair(None)
    print(json.dumps(rec), flush=True)
