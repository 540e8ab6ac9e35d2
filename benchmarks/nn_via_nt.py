"""dX = G . W (NN: both operands row-major) against `W^T copy + NT` on the same box: is a transposed copy of the small
operand worth it?   python benchmarks/nn_via_nt.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import timeit, rand  # noqa: E402

dev = c.Device(0)
for M, N, K, beta in ((32768, 1024, 1024, 0.0), (32768, 1024, 1024, 1.0), (4096, 4096, 4096, 0.0), (4096, 4096, 4096, 1.0), (8192, 8192, 8192, 0.0), (2048, 2048, 2048, 0.0)):
    G, W, Wt, dX = rand(dev, (M, K), 0, 0, 1), rand(dev, (K, N), 1, 0, 1), dev.zeros((N, K)), dev.zeros((M, N))
    nn = lambda: c.sgemm(dev, 0, 0, M, N, K, 1.0, G, K, W, N, beta, dX, N)
    tr = lambda: c.transpose_fwd(dev, W, Wt)
    nt = lambda: c.sgemm(dev, 0, 1, M, N, K, 1.0, G, K, Wt, K, beta, dX, N)
    both = lambda: (tr(), nt())
    r = {"M": M, "N": N, "K": K, "beta": beta, "nn_us": [], "nt_us": [], "transpose_plus_nt_us": []}
    timeit(dev, nn, 30)   # clocks settle over the first ~50 ms of load: a first measurement reads 10 - 15 % slow
    for _ in range(3):    # interleaved, so that what is left of the drift hits every variant alike
        for name, f in (("nn_us", nn), ("nt_us", nt), ("transpose_plus_nt_us", both)):
            r[name].append(round(timeit(dev, f, 10) * 1e3, 1))
    r["transpose_us"] = round(timeit(dev, tr, 10) * 1e3, 1)
    print(json.dumps(r), flush=True)
