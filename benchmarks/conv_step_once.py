"""Two C3 module steps (pad -> conv + bias, fwd + both backward passes) with nothing else around them: the target of the
PMC passes (`rocprofv3 --pmc FETCH_SIZE --kernel-trace`, then WRITE_SIZE in a second run) behind profiles/roofline_traffic.json."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import neuronika_amd  # noqa: E402

t = neuronika_amd.tape
dev = t.Device(0)
N = 128
conv = t.nn.Conv2d(dev, 64, 128, [3, 3], [1, 1], t.PaddingMode.zero(), [1, 1], [1, 1], 1)
X = t.from_ndarray(dev, np.random.default_rng(0).random((N, 64, 56, 56), dtype=np.float32)).requires_grad()
G = t.from_ndarray(dev, np.random.default_rng(2).random((N, 128, 56, 56), dtype=np.float32))
y = conv.forward(X)
for _ in range(2):
    y.forward()
    y.no_grad(); y.with_grad()
    y.backward_from(G)
    X.zero_grad(); conv.weight.zero_grad(); conv.bias.zero_grad()
dev.sync()
print("ok")
