"""4096^3 GEMMs per layout with a PADDED leading dimension (ld = 4096 + pad floats): are the fabric re-reads of the dense case
(NT 824 MB, NN 760, TN 556 per launch against 201 MB algorithmic, profiles/r03_traffic_pmc.md) L2 set / channel conflicts of the
16 KB row stride?  Target of a FETCH_SIZE pass:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d OUT -o r -- python benchmarks/gemm_once_ld.py 32"""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402

pad = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = c.Device(0)
n, ld = 4096, 4096 + pad
rng = np.random.default_rng(0)
A, B, C = dev.array(rng.random((n, ld), dtype=np.float32)), dev.array(rng.random((n, ld), dtype=np.float32)), dev.zeros((n, ld))
for _ in range(3):
    c.sgemm(dev, 0, 1, n, n, n, 1.0, A, ld, B, ld, 0.0, C, ld)   # NT
    c.sgemm(dev, 0, 0, n, n, n, 1.0, A, ld, B, ld, 0.0, C, ld)   # NN
    c.sgemm(dev, 1, 0, n, n, n, 1.0, A, ld, B, ld, 0.0, C, ld)   # TN
dev.sync()
e0, e1 = dev.event(), dev.event()
for name, ta, tb in (("NT", 0, 1), ("NN", 0, 0), ("TN", 1, 0)):
    e0.record()
    for _ in range(10):
        c.sgemm(dev, ta, tb, n, n, n, 1.0, A, ld, B, ld, 0.0, C, ld)
    e1.record(); dev.sync()
    print(name, "ld", ld, round(2.0 * n ** 3 * 10 / e0.elapsed_ms(e1) / 1e9, 1), "TFLOP/s", flush=True)
