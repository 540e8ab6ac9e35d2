"""The batched attention GEMMs of C5 (512 heads, S = 1024, dh = 64) under a forced tile (NK_GEMM_FORCE with a library built
with -DNK_AB_GEMM_FORCE) or the heuristic: scores / dP (K = 64, output S x S) and context / dV / dQ / dK (N = 64, K = S)."""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import timeit, rand  # noqa: E402

dev = c.Device(0)
BH, S, D = 512, 1024, 64
Q, K = rand(dev, (BH, S, D), 0), rand(dev, (BH, S, D), 1)
P = rand(dev, (BH, S, S), 2, 0, 1)
O = dev.zeros((BH, S, D))
SC = dev.zeros((BH, S, S))
out = {"force": os.environ.get("NK_GEMM_FORCE", "heuristic")}
f_scores = lambda: c.sgemm_batched(dev, 0, 1, S, S, D, 1.0, Q, D, S * D, 0, K, D, S * D, 0, 0.0, SC, S, S * S, 0, BH, 1)   # Q.K^T
f_ctx = lambda: c.sgemm_batched(dev, 0, 0, S, D, S, 1.0, P, S, S * S, 0, K, D, S * D, 0, 0.0, O, D, S * D, 0, BH, 1)       # P.V
f_dv = lambda: c.sgemm_batched(dev, 1, 0, S, D, S, 1.0, P, S, S * S, 0, K, D, S * D, 0, 0.0, O, D, S * D, 0, BH, 1)        # P^T.dO
flop = 2.0 * BH * S * S * D
for name, f in (("scores_NT_K64", f_scores), ("context_NN_N64", f_ctx), ("dV_TN_N64", f_dv)):
    timeit(dev, f, 3)
    ms = timeit(dev, f, 8)
    out[name] = [round(ms * 1e3, 1), round(flop / ms / 1e9, 1)]
print(json.dumps(out))
