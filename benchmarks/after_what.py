"""What slows a GEMM that follows the attention backward - the cold Infinity Cache or the clocks?  The packed projection's
input-gradient GEMM (NN 32768 x 1024 x 3072) timed by HIP events INSIDE three repeated sequences: alone; behind a 4.3 GB streaming
copy (flushes the 256 MB cache, little arithmetic); behind the attention backward + dK / dV.  us per GEMM launch."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402

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
dQ, dK, dV = dev.zeros((n, d)), dev.zeros((n, d)), dev.zeros((n, d))
bits = dev.zeros((B * H, S, S // 32))
dQKV, dX = mk(3 * d), dev.zeros((n, d))
Wqkv = dev.array(rng.random((3 * d, d), dtype=np.float32) - np.float32(0.5))
c.attention_fwd(dev, Q, K, V, scores, stats, bits, out, B, S, H, dh, scale, p, True, seed, 0)


def gemm(): c.sgemm(dev, 0, 0, n, d, 3 * d, 1.0, dQKV, 3 * d, Wqkv, d, 0.0, dX, d)
def copy(): c.check(c.lib.nk_copy(dev.h, pd.p, ds.p, ds.size))                       # 2.1 GB read + 2.1 GB written
def attn(): c.attention_bwd(dev, dQ, dK, dV, ds, pd, G, out, scores, stats, bits, Q, K, V, B, S, H, dh, scale, p, True, (True, True, True))


db = dev.zeros((3 * d,))
def colsum(): c.unbroadcast_add(dev, db, dQKV, assign=True)                          # the packed bias gradient: 403 MB read, HBM-bound
def attn_colsum(): attn(); colsum()
def attn_copy(): attn(); c.check(c.lib.nk_copy(dev.h, pd.p, ds.p, ds.size // 8))      # + 0.5 GB of streaming (~90 us)
def fwd(): c.attention_fwd(dev, Q, K, V, scores, stats, bits, out, B, S, H, dh, scale, p, True, seed, 0)
def dkdv():
    so, po, pi = S * d, H * S * S, S * S
    c.sgemm_batched(dev, 1, 0, S, dh, S, 1.0, ds, S, po, pi, Q, d, so, dh, 0.0, dK, d, so, dh, B, H)
    c.sgemm_batched(dev, 1, 0, S, dh, S, 1.0, pd, S, po, pi, G, d, so, dh, 0.0, dV, d, so, dh, B, H)


rec = {}
for name, before in (("alone", None), ("behind a 4.3 GB copy", copy), ("behind attention backward + dK / dV", attn),
                     ("behind attention backward + dK / dV + the column sums (67 us of HBM-bound work)", attn_colsum),
                     ("behind attention backward + dK / dV + 0.5 GB of copy", attn_copy),
                     ("behind the attention forward", fwd), ("behind the dK / dV products alone", dkdv), ("alone (again)", None)):
    e0, e1 = dev.event(), dev.event()
    tot, reps = 0.0, 12
    for i in range(reps + 3):
        if before: before()
        e0.record(); gemm(); e1.record(); e1.sync()
        if i >= 3: tot += e0.elapsed_ms(e1)
    rec[name] = round(tot / reps * 1e3, 1)
print(json.dumps(rec))
