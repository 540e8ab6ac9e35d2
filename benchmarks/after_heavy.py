"""Does a projection GEMM run slower right after the traffic-heavy kernels of the C5 backward pass (fused attention backward,
dK / dV products)?  Per-launch HIP-event times of three NN projection GEMMs in different contexts, same process.
    python benchmarks/after_heavy.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuronika_amd import capi as c  # noqa: E402

B, S, H, dh = 32, 1024, 16, 64
dev = c.Device(0)
rng = np.random.default_rng(0)
mk = lambda shape: dev.array(rng.random(shape, dtype=np.float32) - np.float32(0.5))
Q, K, V, G = (mk((B * S, H * dh)) for _ in range(4))
W = mk((H * dh, H * dh))
dX = dev.zeros((B * S, H * dh))
big = lambda: dev.zeros((B * H, S, S))
scores, Pd, dS = big(), big(), big()
stats, out, dQ, dK, dV = dev.zeros((B * H, S, 2)), dev.zeros((B * S, H * dh)), dev.zeros((B * S, H * dh)), dev.zeros((B * S, H * dh)), dev.zeros((B * S, H * dh))
bits = dev.zeros((B * H, S, S // 32))
d, so, po, pi = H * dh, S * H * dh, H * S * S, S * S
scale = float(np.float32(0.125))
M = B * S


def fwd(): c.attention_fwd(dev, Q, K, V, scores, stats, bits, out, B, S, H, dh, scale, 0.1, True, 7, 0)
def bwd(): c.attention_bwd(dev, dQ, dK, dV, dS, Pd, G, out, scores, stats, bits, Q, K, V, B, S, H, dh, scale, 0.1, True, (True, True, True))
def dk(): c.sgemm_batched(dev, 1, 0, S, dh, S, 1.0, dS, S, po, pi, Q, d, so, dh, 0.0, dK, d, so, dh, B, H)
def dv(): c.sgemm_batched(dev, 1, 0, S, dh, S, 1.0, Pd, S, po, pi, G, d, so, dh, 0.0, dV, d, so, dh, B, H)
def nn(src): c.sgemm(dev, 0, 0, M, d, d, 1.0, src, d, W, d, 0.0, dX, d)


def timed(fns):
    evs = [dev.event() for _ in range(len(fns) + 1)]
    evs[0].record()
    for f, e in zip(fns, evs[1:]):
        f(); e.record()
    dev.sync()
    return [round(evs[i].elapsed_ms(evs[i + 1]) * 1e3, 1) for i in range(len(fns))]


fwd()
for _ in range(30):
    nn(dV)           # settle the clocks
res = {}
for rep in range(3):
    res.setdefault("nn_only x6", []).append(timed([lambda: nn(dV)] * 6))
    res.setdefault("bwd (with its dK / dV products), nn x6", []).append(timed([bwd] + [lambda: nn(dV)] * 6))
    res.setdefault("dk, dv, nn x6", []).append(timed([dk, dv] + [lambda: nn(dV)] * 6))
for k, v in res.items():
    print(json.dumps({k: v}))
