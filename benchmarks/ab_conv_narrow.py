"""C3 kernel gradient (+ fused bias gradient): the uniform launch (5 column tiles of 128, the last half empty: 11 % of the MFMAs on
padding columns) against the mixed launch (4 wide tiles + a 64-wide one whose reduction is cut into fewer ranges) for several
prices of a narrow block's k-tile (NK_TUNE_CONV_NARROW, percent of a wide block's).  us per call incl. the reduce launch."""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import timeit, rand  # noqa: E402

dev = c.Device(0)
x = rand(dev, (128, 64, 58, 58), 2, 0, 1)
g = rand(dev, (128, 128, 56, 56), 4, -1, 1)
dw, db = dev.zeros((128, 64, 3, 3)), dev.zeros((128, 1, 1))
flop = 2 * 128 * 128 * 56 * 56 * 64 * 9
rec = {}
for cost in [int(a) for a in sys.argv[1:]] or [0, 50, 60, 65, 70, 75, 80, 90, 100, 0]:
    dev.conv_narrow(cost)
    us = timeit(dev, lambda: c.conv_bwd_kernel_bias(dev, dw, db, g, x, (1, 1), (1, 1), 1, assign=(True, True)), 30) * 1e3
    key = f"cost={cost}"
    rec[key if key not in rec else key + " again"] = [round(us, 1), round(flop / us / 1e6, 1)]
dev.conv_narrow(None)
print(json.dumps(rec))
