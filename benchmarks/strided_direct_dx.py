"""Round 6: the direct (<= 16 channels per group) input gradient of STRIDED convolutions - depthwise / grouped / few-channel layers -
with whatever library NEURONIKA_HIP_LIB points at (run alternately against benchmarks/_ab/head.so: conv_direct_bwd_input_strided_kernel vs the general kernel)."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from neuronika_amd import capi as c
dev = c.Device(0)
rng = np.random.default_rng(0)
def time(fn, reps=20):
    e0, e1 = dev.event(), dev.event()
    for _ in range(3): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.sync()
    return e0.elapsed_ms(e1) / reps * 1e3
for name, xs, ws, g in (("depthwise 3x3 s2 256 ch @56", (128, 256, 56, 56), (256, 1, 3, 3), 256), ("grouped g=32 3x3 s2 128->128 @56", (128, 128, 56, 56), (128, 4, 3, 3), 32),
                        ("5x5 s2 8->16 @112", (64, 8, 112, 112), (16, 8, 5, 5), 1)):
    k = ws[2]; p = k // 2
    ho = (xs[2] + 2 * p - k) // 2 + 1
    W = dev.array(rng.random(ws, dtype=np.float32)); G = dev.array(rng.random((xs[0], ws[0], ho, ho), dtype=np.float32)); DX = dev.zeros(xs)
    fn = lambda: c.conv_bwd_input(dev, DX, G, W, (2, 2), (1, 1), g, assign=True, padding=(p, p))
    print(json.dumps({"shape": name, "bwd_input_us": [round(time(fn), 1) for _ in range(3)]}), flush=True)
