"""Two C5 module steps (MultiheadAttention d=1024 h=16 S=1024 B=32, dropout 0.1, fwd + bwd) with nothing else around
them: the target of the PMC traffic passes (tools/traffic_pmc.sh) behind profiles/roofline_traffic.json."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import neuronika_amd  # noqa: E402

t = neuronika_amd.tape
dev = t.Device(0)
B, S, d, H = 32, 1024, 1024, 16
mha = t.nn.MultiheadAttention(dev, d, H, 0.1, 1)
X = t.from_ndarray(dev, np.random.default_rng(0).random((B * S, d), dtype=np.float32)).requires_grad()
G = t.from_ndarray(dev, np.random.default_rng(5).random((B * S, d), dtype=np.float32))
y = mha.forward(X, B)
leaves = [X] + [getattr(getattr(mha, n), w) for n in "qkvo" for w in ("weight", "bias")]
for _ in range(2):
    y.forward()
    y.no_grad(); y.with_grad()
    y.backward_from(G)
    for p in leaves:
        p.zero_grad()
dev.sync()
print("ok")
