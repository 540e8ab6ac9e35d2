"""Launch-bound training step (the reference's quickstart MLP, 3 -> 5 -> 5 -> 1, batch 64): eager issue through the
tape vs replay of the same step captured into a hipGraph."""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import neuronika_amd  # noqa: E402

nk = neuronika_amd.tape
dev = nk.Device(0)
lins = [nk.nn.Linear(dev, 3, 5, 1), nk.nn.Linear(dev, 5, 5, 2), nk.nn.Linear(dev, 5, 1, 3)]
X, T = nk.rand(dev, [64, 3], 7), nk.rand(dev, [64, 1], 8)
loss = lins[2].forward(lins[1].forward(lins[0].forward(X).relu()).relu()).mse(T, nk.Reduction.Mean)
opt = nk.optim.SGD(0.01)
for l in lins:
    opt.register(l.weight); opt.register(l.bias)


def step():
    loss.forward()
    loss.no_grad(); loss.with_grad()
    loss.backward(1.0)
    opt.step()
    opt.zero_grad()


def rate(fn, n):
    for _ in range(20):
        fn()
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dev.sync()
    return n / (time.perf_counter() - t0)


eager = rate(step, 2000)
dev.graph_begin(); step(); g = dev.graph_end()
graph = rate(g.launch, 2000)
print(json.dumps({"workload": "C1 quickstart MLP training step (batch 64)", "eager_steps_per_s": round(eager, 1),
                  "hipgraph_steps_per_s": round(graph, 1), "speedup": round(graph / eager, 2)}))
