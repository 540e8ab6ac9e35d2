"""Fused attention-probabilities kernels at the C5 size: stored vs recomputed probabilities."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c
from benchmarks.microbench import timeit, rand
dev = c.Device(0)
shape = (512, 1024, 1024)
S, G = rand(dev, shape, 0, -4, 4), rand(dev, shape, 1, -1, 1)
P, O, D = dev.zeros(shape), dev.zeros(shape), dev.zeros(shape)
gb = 4 * 512 * 1024 * 1024 / 1e9
r = {}
r["fwd_store_ms"] = timeit(dev, lambda: c.scale_softmax_dropout_fwd(dev, S, P, O, None, 0.125, 0.1, True, 3, 0), 10)
r["fwd_nostore_ms"] = timeit(dev, lambda: c.scale_softmax_dropout_fwd(dev, S, None, O, None, 0.125, 0.1, True, 3, 0), 10)
r["bwd_probs_assign_ms"] = timeit(dev, lambda: c.scale_softmax_dropout_bwd(dev, D, G, P, None, 0.125, 0.1, True, 3, 0, assign=True), 10)
r["bwd_recomp_assign_ms"] = timeit(dev, lambda: c.scale_softmax_dropout_bwd_from_scores(dev, D, G, S, None, 0.125, 0.1, True, 3, 0, assign=True), 10)
r = {k: round(v, 4) for k, v in r.items()}
r["TBps"] = {"fwd_store": round(3 * gb / r["fwd_store_ms"], 2), "fwd_nostore": round(2 * gb / r["fwd_nostore_ms"], 2),
             "bwd_probs": round(3 * gb / r["bwd_probs_assign_ms"], 2), "bwd_recomp": round(3 * gb / r["bwd_recomp_assign_ms"], 2)}
print(json.dumps(r))
