"""Does the ORDER of the launches behind the C5 attention backward matter?  The backward half of the C5 step as a sequence of raw
C-ABI calls - attention backward (+ dK / dV), the packed projection's input-gradient GEMM (NN, K = 3d), its column reduction and
its weight-gradient GEMM (TN, M = 3d), the out-projection's two gradient GEMMs - in the tape's order and in three other orders,
each timed as a whole (us per sequence, same box, same buffers), plus the sum of the same launches timed one kind at a time.
    python benchmarks/c5_order.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import timeit, rand  # noqa: E402

B, S, H, dh = 32, 1024, 16, 64
d, n = H * dh, B * S
scale, p, seed = float(np.float32(0.125)), 0.1, 7
dev = c.Device(0)
rng = np.random.default_rng(0)
mk = lambda cols: dev.array(rng.random((n, cols), dtype=np.float32) - np.float32(0.5))
Q, K, V, G = mk(d), mk(d), mk(d), mk(d)
big = lambda: dev.zeros((B * H, S, S))
scores, pd, ds = big(), big(), big()
stats, out = dev.zeros((B * H, S, 2)), dev.zeros((n, d))
dQKV = mk(3 * d)                                  # packed projection gradient [dQ | dK | dV] (its values do not matter here)
dQ, dK, dV = dev.zeros((n, d)), dev.zeros((n, d)), dev.zeros((n, d))
bits = dev.zeros((B * H, S, S // 32))
X, dX = mk(d), dev.zeros((n, d))
Wqkv, dWqkv = dev.array(rng.random((3 * d, d), dtype=np.float32) - np.float32(0.5)), dev.zeros((3 * d, d))
Wo, dWo, dCtx = dev.array(rng.random((d, d), dtype=np.float32) - np.float32(0.5)), dev.zeros((d, d)), dev.zeros((n, d))
c.attention_fwd(dev, Q, K, V, scores, stats, bits, out, B, S, H, dh, scale, p, True, seed, 0)


def attn_bwd():      # + dK, dV (two batched TN products)
    c.attention_bwd(dev, dQ, dK, dV, ds, pd, G, out, scores, stats, bits, Q, K, V, B, S, H, dh, scale, p, True, (True, True, True))


def qkv_dx(): c.sgemm(dev, 0, 0, n, d, 3 * d, 1.0, dQKV, 3 * d, Wqkv, d, 0.0, dX, d)          # dX = dQKV . Wqkv
def qkv_dw(): c.sgemm(dev, 1, 0, 3 * d, d, n, 1.0, dQKV, 3 * d, X, d, 0.0, dWqkv, d)          # dW = dQKV^T . X
def out_dx(): c.sgemm(dev, 0, 0, n, d, d, 1.0, G, d, Wo, d, 0.0, dCtx, d)                      # out-projection input gradient
def out_dw(): c.sgemm(dev, 1, 0, d, d, n, 1.0, G, d, out, d, 0.0, dWo, d)                      # out-projection weight gradient


def dkdv():          # the two batched TN products nk_attention_bwd runs internally
    so, po, pi = S * d, H * S * S, S * S
    c.sgemm_batched(dev, 1, 0, S, dh, S, 1.0, ds, S, po, pi, Q, d, so, dh, 0.0, dK, d, so, dh, B, H)
    c.sgemm_batched(dev, 1, 0, S, dh, S, 1.0, pd, S, po, pi, G, d, so, dh, 0.0, dV, d, so, dh, B, H)


if os.environ.get("C5_ORDER_SPLIT") == "1":
    # with a library whose nk_attention_bwd leaves the dK / dV products to the caller (variant build, tools/sessions/r04_zz.sh): the
    # out-projection's weight-gradient GEMM BETWEEN the attention backward kernel and its dK / dV products
    rec = {}
    seqs = (("A: attention kernel, dK / dV, out_dw, qkv_dx, qkv_dw (the tape's order)", [out_dx, attn_bwd, dkdv, out_dw, qkv_dx, qkv_dw]),
            ("B: attention kernel, out_dw, dK / dV, qkv_dx, qkv_dw", [out_dx, attn_bwd, out_dw, dkdv, qkv_dx, qkv_dw]),
            ("D: attention kernel, dK / dV, qkv_dx, out_dw, qkv_dw", [out_dx, attn_bwd, dkdv, qkv_dx, out_dw, qkv_dw]),
            ("E: attention kernel, dK / dV, qkv_dx, qkv_dw, out_dw", [out_dx, attn_bwd, dkdv, qkv_dx, qkv_dw, out_dw]))
    for name, seq in seqs:
        timeit(dev, lambda: [f() for f in seq], 5)       # every order once before any is recorded
    for rep in range(4):                                  # alternating: four passes over the four legal orders
        for name, seq in seqs:
            rec.setdefault(name, []).append(round(timeit(dev, lambda: [f() for f in seq], 20) * 1e3, 1))
    print(json.dumps(rec))
    sys.exit(0)

orders = {
    "tape order: out_dx out_dw | attn_bwd | qkv_dx qkv_dw": [out_dx, out_dw, attn_bwd, qkv_dx, qkv_dw],
    "out_dw moved behind the attention backward": [out_dx, attn_bwd, out_dw, qkv_dx, qkv_dw],
    "qkv_dw before qkv_dx": [out_dx, out_dw, attn_bwd, qkv_dw, qkv_dx],
    "light first: out_dw then qkv_dw then qkv_dx": [out_dx, attn_bwd, out_dw, qkv_dw, qkv_dx],
}
rec = {}
alone = {f.__name__: timeit(dev, f, 10) * 1e3 for f in (out_dx, out_dw, attn_bwd, qkv_dx, qkv_dw)}
rec["each kind alone (us)"] = {k: round(v, 1) for k, v in alone.items()}
rec["sum of the kinds alone"] = round(sum(alone.values()), 1)
for rep in range(2):
    for name, seq in orders.items():
        us = timeit(dev, lambda: [f() for f in seq], 10) * 1e3
        rec[name + (" (again)" if rep else "")] = round(us, 1)
print(json.dumps(rec, indent=1))
