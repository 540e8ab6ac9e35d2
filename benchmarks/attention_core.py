"""C5 attention core, fused kernels vs the node-by-node device path they replace (same box, same buffers).
    python benchmarks/attention_core.py [B S H] [reps]
Prints one JSON line per variant: ms per call and TFLOP/s on the 4*B*H*S*S*dh algorithmic flop of each direction's two
MFMA products; both backward variants include the dK / dV products (4 products = 8*B*H*S*S*dh flop, reported on 4*...)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuronika_amd import capi as c  # noqa: E402


def main():
    B, S, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 1024, 16)
    reps = int(sys.argv[4]) if len(sys.argv) >= 5 else 10
    dh, p, seed = 64, float(os.environ.get("NK_ATT_P", "0.1")), 7
    scale = float(np.float32(0.125))
    dev = c.Device(0)
    rng = np.random.default_rng(0)
    mk = lambda: dev.array(rng.random((B * S, H * dh), dtype=np.float32) - np.float32(0.5))
    Q, K, V, G = mk(), mk(), mk(), mk()
    big = lambda: dev.zeros((B * H, S, S))
    scores, probs_d, dP, dS = big(), big(), big(), big()
    SP = c.attention_padded(S)   # the fused core's scratch tensors are whole 32 x 32 tiles (ragged S: its own, padded set)
    fbig = lambda: dev.zeros((B * H, SP, SP))
    f_scores, f_pd, f_ds = (scores, probs_d, dS) if SP == S else (fbig(), fbig(), fbig())
    stats, out, dQ = dev.zeros((B * H, SP, 2)), dev.zeros((B * S, H * dh)), dev.zeros((B * S, H * dh))
    dK, dV = dev.zeros((B * S, H * dh)), dev.zeros((B * S, H * dh))
    bits = dev.zeros((B * H, SP, SP // 32))
    d, so, po, pi = H * dh, S * H * dh, H * S * S, S * S
    flop = 4.0 * B * H * S * S * dh

    def fused_fwd():
        c.attention_fwd(dev, Q, K, V, f_scores, stats, bits, out, B, S, H, dh, scale, p, True, seed, 0)

    def fused_bwd():
        c.attention_bwd(dev, dQ, dK, dV, f_ds, f_pd, G, out, f_scores, stats, bits, Q, K, V, B, S, H, dh, scale, p, True, (True, True, True))

    def nodes_fwd():
        c.sgemm_batched(dev, 0, 1, S, S, dh, 1.0, Q, d, so, dh, K, d, so, dh, 0.0, scores, S, po, pi, B, H)
        c.scale_softmax_dropout_fwd(dev, scores, None, probs_d, None, scale, p, True, seed, 0)
        c.sgemm_batched(dev, 0, 0, S, dh, S, 1.0, probs_d, S, po, pi, V, d, so, dh, 0.0, out, d, so, dh, B, H)

    def nodes_bwd():
        c.sgemm_batched(dev, 0, 1, S, S, dh, 1.0, G, d, so, dh, V, d, so, dh, 0.0, dP, S, po, pi, B, H)
        c.scale_softmax_dropout_bwd_from_scores(dev, dS, dP, scores, None, scale, p, True, seed, 0, assign=True)
        c.sgemm_batched(dev, 0, 0, S, dh, S, 1.0, dS, S, po, pi, K, d, so, dh, 0.0, dQ, d, so, dh, B, H)
        c.sgemm_batched(dev, 1, 0, S, dh, S, 1.0, dS, S, po, pi, Q, d, so, dh, 0.0, dK, d, so, dh, B, H)
        c.sgemm_batched(dev, 1, 0, S, dh, S, 1.0, probs_d, S, po, pi, G, d, so, dh, 0.0, dV, d, so, dh, B, H)

    for name, fn in (("fused_fwd", fused_fwd), ("fused_bwd", fused_bwd), ("nodes_fwd", nodes_fwd), ("nodes_bwd", nodes_bwd),
                     ("fused_fwd", fused_fwd), ("fused_bwd", fused_bwd)):
        for _ in range(2):
            fn()
        dev.sync()
        e0, e1 = dev.event(), dev.event()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        dev.sync()
        ms = e0.elapsed_ms(e1) / reps
        print(json.dumps({"variant": name, "B": B, "S": S, "H": H, "ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 2)}), flush=True)


if __name__ == "__main__":
    main()
