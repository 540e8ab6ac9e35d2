"""Fixed cost of a GEMM launch next to its k-loop: 4096 x 4096 x K for K = 32 ... 4096, beta = 0 and 1 (where the
remaining 14 % of the 4096^3 launch go)."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neuronika_amd import capi as c
from benchmarks.microbench import timeit, rand
dev = c.Device(0)
n = 4096
out = {}
for K in (32, 64, 128, 256, 1024, 4096):
    A, B, C = rand(dev, (n, K), 0, 0, 1), rand(dev, (K, n), 1, 0, 1), dev.zeros((n, n))
    for beta in (0.0, 1.0):
        f = lambda: c.sgemm(dev, 0, 0, n, n, K, 1.0, A, K, B, n, beta, C, n)
        timeit(dev, f, 20)
        out[f"K{K}_b{int(beta)}"] = round(timeit(dev, f, 50) * 1e3, 1)
print(json.dumps(out))
