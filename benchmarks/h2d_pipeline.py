"""PCIe-inclusive rate of the input pipeline: C4-sized batches (4096 x 4096 f32 records + targets, 128 MiB per step)
streamed from page-locked host memory through data::DeviceLoader (copy stream, double buffered), alone and overlapped
with the C4 training step."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import neuronika_amd  # noqa: E402

nk = neuronika_amd.tape
dev = nk.Device(0)
H = B = 4096
NB = 4
rng = np.random.default_rng(0)
rec = rng.random((NB * B, H), dtype=np.float32)
lab = rng.random((NB * B, H), dtype=np.float32)
ds = nk.data.LabeledDataset(rec, lab)
loader = nk.data.DeviceLoader(dev, ds, B, True)
X, T = nk.zeros(dev, [B, H]), nk.zeros(dev, [B, H])


def rate(fn, n):
    for _ in range(3):
        fn()
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dev.sync()
    return (time.perf_counter() - t0) / n


def feed():
    if loader.next_into(X, T) == 0:
        loader.next_into(X, T)


t_feed = rate(feed, 24)
lins = [nk.nn.Linear(dev, H, H, s) for s in (1, 3, 5)]
loss = lins[2].forward(lins[1].forward(lins[0].forward(X).relu()).relu()).mse(T, nk.Reduction.Mean)
opt = nk.optim.SGD(1e-3)
for l in lins:
    opt.register(l.weight); opt.register(l.bias)


def step():
    loss.forward()
    loss.no_grad(); loss.with_grad()
    loss.backward(1.0)
    opt.step(); opt.zero_grad()


def fed_step():
    feed()
    step()


t_step = rate(step, 20)
t_both = rate(fed_step, 20)
mib = 2 * B * H * 4 / 2 ** 20
print(json.dumps({"batch_MiB": mib, "feed_only_ms": round(t_feed * 1e3, 3), "h2d_GBps": round(2 * B * H * 4 / t_feed / 1e9, 1),
                  "step_resident_ms": round(t_step * 1e3, 3), "step_with_fresh_batch_ms": round(t_both * 1e3, 3),
                  "samples_per_s_pcie_inclusive": round(B / t_both, 1)}))
