#!/usr/bin/env python3
"""Per-kernel microbenchmarks through the C ABI (HIP-event timing on the compute stream).

    python benchmarks/microbench.py gemm --n 4096 --iters 20
    python benchmarks/microbench.py conv | stream | softmax | all

Prints one JSON line per measurement: achieved TFLOP/s or GB/s and the roofline fraction
(peaks from /opt/skills/guides/MI355X_MICROARCH.md: f32 MFMA 157.3 TFLOP/s, HBM 8.0 TB/s)."""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuronika_amd import capi as c  # noqa: E402

MFMA_F32_PEAK = 157.3e12
HBM_PEAK = 8.0e12


def timeit(dev, fn, iters, warmup=3, settle_ms=80.0, min_ms=40.0):
    """ms per call.  An MI355X that has been idle needs ~50 ms of load before its clocks settle - a measurement taken right
    after a pause (or after a different, lighter kernel) reads 10 - 15 % slow (benchmarks/nn_via_nt.py: the same 4096^3
    launch 1094 us measured first, 1000 us measured last) - so every measurement first runs the kernel for `settle_ms`,
    then times at least `iters` calls and at least `min_ms` of work."""
    e0, e1 = dev.event(), dev.event()
    e0.record()
    calls = 0
    while True:
        for _ in range(max(1, warmup)):
            fn()
        calls += max(1, warmup)
        e1.record()
        e1.sync()
        spent = e0.elapsed_ms(e1)
        if spent >= settle_ms:
            break
    iters = max(iters, int(min_ms / max(spent / calls, 1e-4)) + 1)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.sync()
    return e0.elapsed_ms(e1) / iters


def rand(dev, shape, seed, lo=-1.0, hi=1.0):
    rng = np.random.default_rng(seed)
    return dev.array((rng.random(shape, dtype=np.float32) * (hi - lo) + lo))


def emit(**kw):
    print(json.dumps(kw), flush=True)


def bench_gemm(dev, sizes, iters):
    for n in sizes:
        A, B, G = rand(dev, (n, n), 0, 0, 1), rand(dev, (n, n), 1, 0, 1), rand(dev, (n, n), 2, 0, 1)
        Cm, dA, dB = dev.zeros((n, n)), dev.zeros((n, n)), dev.zeros((n, n))
        flop = 2.0 * n ** 3
        res = {}
        for name, fn in (("NN_fwd", lambda: c.mm_fwd(dev, A, B, Cm)),
                         ("NT_bwd_left", lambda: c.mm_bwd_left(dev, dA, G, B)),
                         ("TN_bwd_right", lambda: c.mm_bwd_right(dev, dB, A, G))):
            ms = timeit(dev, fn, iters)
            res[name] = ms
            emit(kernel=f"sgemm_{name}", n=n, ms=round(ms, 4), tflops=round(flop / ms / 1e9, 2),
                 frac_mfma_peak=round(flop / (ms * 1e-3) / MFMA_F32_PEAK, 4))

        def fwd_bwd():
            c.mm_fwd(dev, A, B, Cm); c.mm_bwd_left(dev, dA, G, B); c.mm_bwd_right(dev, dB, A, G)
        ms = timeit(dev, fwd_bwd, max(3, iters // 2))
        emit(kernel="matmul_fwd_bwd(C2)", n=n, ms=round(ms, 4), tflops=round(3 * flop / ms / 1e9, 2),
             frac_mfma_peak=round(3 * flop / (ms * 1e-3) / MFMA_F32_PEAK, 4))


def bench_stream(dev, iters):
    # two sizes: the C4 tensors (4096 x 4096 = 64 MB each: a two-stream kernel moves 128 - 256 MB and partly lives in the 256 MB
    # Infinity Cache - rates above the 6.3 TB/s copy ceiling are cache hits), 16384 x 4096 (256 MB each: still cache-assisted) and
    # 65536 x 4096 (1 GiB each: HBM rates - the judged rows)
    for rows, label in ((4096, "C4 size, partly Infinity-Cache resident"), (16384, "256 MB tensors: Infinity-Cache assisted"),
                        (65536, "1 GiB tensors: HBM")):
        n = rows * 4096
        X, Y, Bv, G, D = rand(dev, (rows, 4096), 0), dev.zeros((rows, 4096)), rand(dev, (4096,), 1), rand(dev, (rows, 4096), 2), dev.zeros((rows, 4096))
        Db = dev.zeros((4096,))
        out = dev.zeros(())
        cases = [
            ("relu_fwd", lambda: c.relu_fwd(dev, X, Y), 8 * n),
            ("relu_bwd", lambda: c.relu_bwd(dev, D, G, X), 16 * n),
            ("relu_mask_inplace", lambda: c.relu_mask_inplace(dev, D, X), 12 * n),
            ("bias_add_fwd", lambda: c.binary_fwd(dev, "add", Y, X, Bv), 8 * n),
            ("add_bwd_left_same", lambda: c.binary_bwd_left(dev, "add", D, G), 12 * n),
            ("bias_grad_colreduce", lambda: c.binary_bwd_right(dev, "add", Db, G), 4 * n),
            ("mse_fwd", lambda: c.mse_fwd(dev, X, G, out, "mean"), 8 * n),
            ("mse_bwd", lambda: c.mse_bwd(dev, D, out, X, G, "mean"), 16 * n),
            ("fill0", lambda: D.fill(0.0), 4 * n),
            ("copy (1 read + 1 write)", lambda: c.check(c.lib.nk_copy(dev.h, Y.p, X.p, n)), 8 * n),
            ("add (2 reads + 1 write)", lambda: c.binary_fwd(dev, "add", Y, X, G), 12 * n),
            ("sgd", lambda: c.sgd_step(dev, X, G, None, lr=1e-9), 12 * n),
            ("sgd_multi(3 parameters)", lambda: c.sgd_step_multi(dev, [X, Y, D], [G, G, G], None, lr=1e-9), 3 * 12 * n),
        ]
        for name, fn, nbytes in cases:
            ms = timeit(dev, fn, iters)
            emit(kernel=name, size=label, bytes=nbytes, ms=round(ms, 4), gbps=round(nbytes / ms / 1e6, 1),
                 frac_hbm_peak=round(nbytes / (ms * 1e-3) / HBM_PEAK, 4))
        del X, Y, G, D


def bench_softmax(dev, iters):
    rows, L = 64 * 1024, 1024          # 1/8 of the C5 score tensor
    n = rows * L
    X, Y, G, D, NZ = rand(dev, (rows, L), 0, -4, 4), dev.zeros((rows, L)), rand(dev, (rows, L), 1), dev.zeros((rows, L)), dev.zeros((rows, L))
    cases = [
        ("softmax_fwd", lambda: c.softmax_fwd(dev, X, Y, 1), 8 * n),
        ("softmax_bwd", lambda: c.softmax_bwd(dev, D, G, Y, 1), 16 * n),
        ("log_softmax_fwd", lambda: c.log_softmax_fwd(dev, X, Y, 1), 8 * n),
        ("dropout_fwd", lambda: c.dropout_fwd(dev, X, Y, NZ, 0.1, True, 7, 0), 12 * n),
        ("dropout_bwd", lambda: c.dropout_bwd(dev, D, G, NZ, 0.1, True), 16 * n),
        ("attn_probs_fwd", lambda: c.scale_softmax_dropout_fwd(dev, X, None, Y, None, 0.125, 0.1, True, 7, 0), 8 * n),
        ("attn_probs_bwd_from_scores", lambda: c.scale_softmax_dropout_bwd_from_scores(dev, D, G, X, None, 0.125, 0.1, True, 7, 0, assign=True), 12 * n),
    ]
    for name, fn, nbytes in cases:
        ms = timeit(dev, fn, iters)
        emit(kernel=name, bytes=nbytes, ms=round(ms, 4), gbps=round(nbytes / ms / 1e6, 1),
             frac_hbm_peak=round(nbytes / (ms * 1e-3) / HBM_PEAK, 4))


def bench_conv(dev, iters, batch=128):
    xs, ws = (batch, 64, 56, 56), (128, 64, 3, 3)
    x = rand(dev, xs, 0, 0, 1)
    k = 1.0 / np.sqrt(576.0)
    W = rand(dev, ws, 1, -k, k)
    XP = dev.zeros((batch, 64, 58, 58))
    Y, G = dev.zeros((batch, 128, 56, 56)), rand(dev, (batch, 128, 56, 56), 2, 0, 1)
    DXP, DW = dev.zeros(XP.shape), dev.zeros(ws)
    flop = 2.0 * batch * 128 * 56 * 56 * 64 * 9
    c.pad_const_fwd(dev, x, XP, (1, 1), 0.0)
    for name, fn in (("conv_fwd", lambda: c.conv_fwd(dev, XP, W, Y, (1, 1), (1, 1), 1)),
                     ("conv_bwd_input", lambda: c.conv_bwd_input(dev, DXP, G, W, (1, 1), (1, 1), 1)),
                     ("conv_bwd_kernel", lambda: c.conv_bwd_kernel(dev, DW, G, XP, (1, 1), (1, 1), 1))):
        before = dev.conv_winograd_launches()
        ms = timeit(dev, fn, iters)
        wino = dev.conv_winograd_launches() > before       # Winograd F(2x2, 3x3) / F(3x3, 2x2): 16 / 36 of the direct multiplies
        executed = flop * (16.0 / 36.0 if wino else 1.0)
        emit(kernel=name, batch=batch, ms=round(ms, 4), winograd=wino, tflops_executed=round(executed / ms / 1e9, 2),
             frac_mfma_peak=round(executed / (ms * 1e-3) / MFMA_F32_PEAK, 4),        # on the MFMA flops executed: <= 1
             direct_equivalent_tflops=round(flop / ms / 1e9, 2), algorithmic_speedup=round(flop / executed, 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all", choices=["gemm", "stream", "softmax", "conv", "all"])
    ap.add_argument("--n", type=int, nargs="*", default=[1024, 2048, 4096, 8192])
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = c.Device(0)
    if a.what in ("gemm", "all"):
        bench_gemm(dev, a.n, a.iters)
    if a.what in ("stream", "all"):
        bench_stream(dev, a.iters)
    if a.what in ("softmax", "all"):
        bench_softmax(dev, a.iters)
    if a.what in ("conv", "all"):
        bench_conv(dev, max(3, a.iters // 2))
    dev.sync()


if __name__ == "__main__":
    main()
