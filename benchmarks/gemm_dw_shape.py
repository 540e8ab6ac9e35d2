"""GEMM at the shape of the C3 kernel-gradient pass (dW = G . cols^T: M = Cout = 128, N = Cin*9 = 576, K = N*L = 401,408),
every operand layout: how much of the conv kernel's gap to the square GEMM is the SHAPE (deep split-K) and how much the gather."""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import timeit, rand  # noqa: E402

dev = c.Device(0)
out = {"lib": os.path.basename(c.LIB_PATH)}
K = 401408
for (ta, tb, M, N) in [(0, 1, 128, 576), (0, 1, 128, 640), (0, 1, 128, 512), (1, 0, 128, 576), (0, 0, 128, 576), (1, 1, 128, 576),
                       (0, 1, 576, 128), (0, 1, 640, 128)]:
    A = rand(dev, (K, M) if ta else (M, K), 0, 0, 1)
    B = rand(dev, (N, K) if tb else (K, N), 1, 0, 1)
    C = dev.zeros((M, N))
    lda, ldb = (M if ta else K), (K if tb else N)
    f = lambda: c.sgemm(dev, ta, tb, M, N, K, 1.0, A, lda, B, ldb, 1.0, C, N)
    timeit(dev, f, 3)
    ms = timeit(dev, f, 10)
    out[f"{'T' if ta else 'N'}{'T' if tb else 'N'}_{M}x{N}x{K}"] = [round(2.0 * M * N * K / ms / 1e9, 1), round(ms * 1e3, 1)]
    del A, B
print(json.dumps(out))
