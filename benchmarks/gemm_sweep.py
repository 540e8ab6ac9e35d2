#!/usr/bin/env python3
"""K / MN sweeps of nk_sgemm to separate per-tile fixed cost from the k-loop rate."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuronika_amd import capi as c
from benchmarks.microbench import timeit, rand

dev = c.Device(0)
def run(M, N, K, ta=0, tb=0, beta=0.0, iters=10):
    A = rand(dev, (K, M) if ta else (M, K), 0); B = rand(dev, (N, K) if tb else (K, N), 1); C = dev.zeros((M, N))
    ms = timeit(dev, lambda: c.sgemm(dev, ta, tb, M, N, K, 1.0, A, A.shape[1], B, B.shape[1], beta, C, N), iters)
    print(json.dumps(dict(M=M, N=N, K=K, ta=ta, tb=tb, beta=beta, ms=round(ms, 4), tflops=round(2.0 * M * N * K / ms / 1e9, 1))), flush=True)

for _ in range(30): run(4096, 4096, 4096, iters=10) if _ == 29 else None
for K in ():
    run(4096, 4096, K)
for beta in (0.0, 1.0):
    run(4096, 4096, 4096, beta=beta)
for MN in (2048, 4096, 8192):
    run(MN, MN, 2048)
run(4096, 4096, 32)     # almost pure prologue + epilogue
run(8192, 8192, 32)
