"""A few 4096^3 GEMMs per layout with nothing else around them: the target of the PMC passes behind profiles/r01_sgemm_pmc.md."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import rand  # noqa: E402

dev = c.Device(0)
n = 4096
A, B, C = rand(dev, (n, n), 0, 0, 1), rand(dev, (n, n), 1, 0, 1), dev.zeros((n, n))
for _ in range(3):
    c.sgemm(dev, 0, 1, n, n, n, 1.0, A, n, B, n, 0.0, C, n)   # NT (Linear forward)
    c.sgemm(dev, 0, 0, n, n, n, 1.0, A, n, B, n, 0.0, C, n)   # NN (input gradient)
    c.sgemm(dev, 1, 0, n, n, n, 1.0, A, n, B, n, 0.0, C, n)   # TN (weight gradient)
dev.sync()
print("ok")
