"""GEMM on shapes that are not multiples of the tile / not 16-byte aligned (the guarded loader path) next to 4096^3."""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import timeit, rand  # noqa: E402

dev = c.Device(0)
out = {"lib": os.path.basename(c.LIB_PATH)}
for (ta, tb, M, N, K) in [(0, 0, 4096, 4096, 4096), (0, 0, 4000, 4000, 4000), (0, 1, 4000, 4000, 4000), (1, 0, 4000, 4000, 4000),
                          (0, 0, 4100, 4100, 4100), (0, 0, 4096, 4096, 4090), (0, 1, 4095, 4097, 4093), (0, 1, 1000, 1000, 1000),
                          (0, 1, 32768, 1000, 1024)]:
    A = rand(dev, (K, M) if ta else (M, K), 0, 0, 1)
    B = rand(dev, (N, K) if tb else (K, N), 1, 0, 1)
    C = dev.zeros((M, N))
    lda, ldb = (M if ta else K), (K if tb else N)
    f = lambda: c.sgemm(dev, ta, tb, M, N, K, 1.0, A, lda, B, ldb, 0.0, C, N)
    timeit(dev, f, 3)
    ms = timeit(dev, f, 10)
    out[f"{'T' if ta else 'N'}{'T' if tb else 'N'}_{M}x{N}x{K}"] = round(2.0 * M * N * K / ms / 1e9, 1)
    del A, B, C
print(json.dumps(out))
