"""Sampled references for full-size 3 x 3 / stride 1 / pad 1 convolutions (C3 and the tape test beside it): the oracle is too slow
to replay 128 x 64 x 56 x 56 whole, so a test draws positions, restates each of them as ONE dot product - in f64 (the yardstick)
and in f32 (OpenBLAS `sdot`, the CPU restatement of the same sum) - and hands the three vectors to
`tolerance.assert_contraction`, which asserts SURVEY.md 8c(ii)'s bound with the operands' own maxima and records the margin.

Follows the direct sums of the reference's convolution (node/convolution/mod.rs:85-226: forward :110-122, input gradient :146-189,
kernel gradient :190-226); `xp` is the zero-padded input (node/pad/mod.rs:97-129)."""
import numpy as np


def _dot(a, b):
    """(f64 sum, f32 sum) of one contraction given its two f32 operand vectors"""
    a, b = np.ascontiguousarray(a, dtype=np.float32).ravel(), np.ascontiguousarray(b, dtype=np.float32).ravel()
    return float(np.dot(a.astype(np.float64), b.astype(np.float64))), np.float32(np.dot(a, b))


def forward(xp, w, bias, rng, count):
    """-> (index arrays, ref64, cpu32) of `count` outputs y[n, co, oh, ow] = sum(xp[n, :, oh:oh+3, ow:ow+3] * w[co]) (+ bias[co])"""
    N, _, HP, WP = xp.shape
    idx = (rng.integers(0, N, count), rng.integers(0, w.shape[0], count), rng.integers(0, HP - 2, count), rng.integers(0, WP - 2, count))
    r64, r32 = [], []
    for n, co, oh, ow in zip(*idx):
        d, s = _dot(xp[n, :, oh:oh + 3, ow:ow + 3], w[co])
        if bias is not None:
            d, s = d + float(bias.reshape(-1)[co]), np.float32(s + bias.reshape(-1)[co])
        r64.append(d); r32.append(s)
    return idx, np.array(r64), np.array(r32, np.float32)


def input_gradient(g, w, rng, count, padded):
    """-> (index arrays, ref64, cpu32) of `count` elements of the input gradient; padded: positions of the PADDED input
    (58 x 58 at C3), else of the unpadded one (the module's form: Pad backward = the centre slice)"""
    N, Cout, H, W = g.shape
    Cin = w.shape[1]
    ext = 2 if padded else 0
    idx = (rng.integers(0, N, count), rng.integers(0, Cin, count), rng.integers(0, H + ext, count), rng.integers(0, W + ext, count))
    r64, r32 = [], []
    for n, ci, ph, pw in zip(*idx):
        if not padded:
            ph, pw = ph + 1, pw + 1
        ga, wa = [], []
        for kh in range(3):
            for kw in range(3):
                oh, ow = ph - kh, pw - kw
                if 0 <= oh < H and 0 <= ow < W:
                    ga.append(g[n, :, oh, ow]); wa.append(w[:, ci, kh, kw])
        d, s = _dot(np.concatenate(ga), np.concatenate(wa)) if ga else (0.0, np.float32(0))
        r64.append(d); r32.append(s)
    return idx, np.array(r64), np.array(r32, np.float32)


def kernel_gradient(g, xp, rng, count):
    """-> (index arrays, ref64, cpu32) of `count` elements dw[co, ci, kh, kw] = sum(g[:, co] * xp[:, ci, kh:kh+H, kw:kw+W])"""
    N, Cout, H, W = g.shape
    Cin = xp.shape[1]
    idx = (rng.integers(0, Cout, count), rng.integers(0, Cin, count), rng.integers(0, 3, count), rng.integers(0, 3, count))
    r64, r32 = [], []
    for co, ci, kh, kw in zip(*idx):
        d, s = _dot(g[:, co], xp[:, ci, kh:kh + H, kw:kw + W])
        r64.append(d); r32.append(s)
    return idx, np.array(r64), np.array(r32, np.float32)
