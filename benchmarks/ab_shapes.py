"""GEMM shape sweep (mid sizes, attention and projection shapes) for A/B runs of the tile / split-K heuristic."""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import timeit, rand  # noqa: E402

dev = c.Device(0)
out = {"lib": os.path.basename(c.LIB_PATH)}
big = rand(dev, (4096, 4096), 9, 0, 1)
timeit(dev, lambda: c.sgemm(dev, 0, 0, 4096, 4096, 4096, 1.0, big, 4096, big, 4096, 0.0, dev.zeros((4096, 4096)), 4096), 30)
for (ta, tb, M, N, K) in [(0, 0, 1024, 1024, 1024), (0, 0, 2048, 2048, 2048), (0, 1, 2048, 2048, 2048), (1, 0, 2048, 2048, 2048),
                          (1, 0, 1024, 1024, 32768), (0, 1, 32768, 1024, 1024), (0, 0, 32768, 1024, 1024), (0, 0, 512, 512, 8192),
                          (0, 1, 64, 4096, 4096), (1, 0, 4096, 64, 4096), (0, 0, 3072, 3072, 3072), (0, 0, 1536, 1536, 1536), (0, 0, 2560, 2560, 2560), (0, 0, 5120, 5120, 1024),
                          (0, 0, 4096, 4096, 4096), (0, 1, 4096, 3072, 1024), (0, 0, 6144, 1024, 1024)]:
    A = rand(dev, (K, M) if ta else (M, K), 0, 0, 1)
    B = rand(dev, (N, K) if tb else (K, N), 1, 0, 1)
    C = dev.zeros((M, N))
    lda, ldb = (M if ta else K), (K if tb else N)
    f = lambda: c.sgemm(dev, ta, tb, M, N, K, 1.0, A, lda, B, ldb, 1.0, C, N)
    timeit(dev, f, 5)
    ms = timeit(dev, f, 20)
    out[f"{'T' if ta else 'N'}{'T' if tb else 'N'}_{M}x{N}x{K}"] = round(2.0 * M * N * K / ms / 1e9, 1)
print(json.dumps(out))
