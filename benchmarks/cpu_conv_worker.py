"""Worker of bench.py's C3 `cpu_baseline` (a module of its own so that spawned worker processes can import it by
name).  Test/bench infrastructure: it runs the CPU oracle, never the product."""
import time

import numpy as np


def conv_chunk(args):
    """One worker's share of the batch (the reference's rayon task, node/convolution/mod.rs:110-122): pad + conv
    forward + both backward passes of `n` samples with the oracle, one BLAS thread."""
    seed, n = args
    from threadpoolctl import threadpool_limits
    from oracle import neuronika_oracle as O
    with threadpool_limits(limits=1):
        x = np.random.default_rng(seed).random((n, 64, 56, 56), dtype=np.float32)
        k = 1.0 / np.sqrt(576.0)
        w = ((np.random.default_rng(1).random((128, 64, 3, 3), dtype=np.float32) * 2 - 1) * k).astype(np.float32)
        g = np.random.default_rng(2).random((n, 128, 56, 56), dtype=np.float32)
        t0 = time.perf_counter()
        xp = np.zeros((n, 64, 58, 58), np.float32)
        O.pad_constant_forward(x, xp, (1, 1), 0.0)
        y = np.zeros((n, 128, 56, 56), np.float32)
        O.convolution_forward(xp, w, y, (1, 1), (1, 1), 1)
        dxp, dw = np.zeros_like(xp), np.zeros_like(w)
        O.convolution_backward_input(dxp, g, w, (1, 1), (1, 1), 1)
        O.convolution_backward_kernel(dw, g, xp, (1, 1), (1, 1), 1)
        return time.perf_counter() - t0


