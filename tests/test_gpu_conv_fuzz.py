"""Seeded random convolution geometries (1-3 spatial dims, channel counts that select the fast / generic / direct kernel
families, strides, dilations, groups, output widths around the quad boundaries, zero padding) through the C ABI against
the oracle: forward (with and without the fused bias), both backward passes in `+=` and first-write form, and the
backward-input with the module's zero padding folded in.  Tolerance: the contraction criterion of test_gpu_parity."""
import numpy as np
import pytest

from oracle import neuronika_oracle as O

pytestmark = pytest.mark.gpu


def capi():
    from neuronika_amd import capi as c
    return c


@pytest.fixture(scope="module")
def dev():
    return capi().Device(0)


def contraction_ok(gpu, cpu32, ref64, K, amax, bmax):
    """the one contraction policy of the suite: tests/tolerance.py"""
    from tolerance import assert_contraction
    assert_contraction("conv_fuzz", gpu, ref64, K, amax, bmax, cpu32=cpu32)


CHANNELS = [1, 2, 3, 4, 8, 16, 20, 24, 32, 64]   # per group: <= 16 direct, 17..31 generic, multiples of 32 fast


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    nd = int(rng.choice([1, 2, 2, 2, 3]))
    groups = int(rng.choice([1, 1, 2, 3]))
    cg, mg = int(rng.choice(CHANNELS)), int(rng.choice(CHANNELS))
    if nd == 3:
        cg, mg = min(cg, 32), min(mg, 32)
    k = [int(rng.integers(1, 4)) for _ in range(nd)]
    stride = [int(rng.choice([1, 1, 1, 2, 3])) for _ in range(nd)]
    dil = [int(rng.choice([1, 1, 2])) for _ in range(nd)]
    pad = [int(rng.integers(0, 3)) for _ in range(nd)]
    out = [int(rng.integers(1, 7)) for _ in range(nd)]
    out[-1] = int(rng.choice([1, 3, 4, 5, 7, 8, 9, 12]))                 # around the quad boundaries
    padded = [(o - 1) * s + d * (kk - 1) + 1 + int(rng.integers(0, s)) for o, s, d, kk in zip(out, stride, dil, k)]
    unp = [max(1, p - 2 * q) for p, q in zip(padded, pad)]
    padded = [u + 2 * q for u, q in zip(unp, pad)]
    if any(p < d * (kk - 1) + 1 for p, d, kk in zip(padded, dil, k)):
        return None
    n = int(rng.integers(1, 4))
    return dict(nd=nd, xs=(n, cg * groups) + tuple(unp), ws=(mg * groups, cg) + tuple(k), stride=tuple(stride),
                dil=tuple(dil), pad=tuple(pad), groups=groups)


@pytest.mark.parametrize("seed", range(160))
def test_conv_random_geometry(dev, seed):
    case = random_case(seed)
    if case is None:
        pytest.skip("degenerate geometry")
    c = capi()
    xs, ws, s, d, g, pad = case["xs"], case["ws"], case["stride"], case["dil"], case["groups"], case["pad"]
    rng = np.random.default_rng(seed)
    x = (rng.random(xs, dtype=np.float32) * 2 - 1)
    w = (rng.random(ws, dtype=np.float32) * 2 - 1)
    sl = (slice(None), slice(None)) + tuple(slice(q, q + u) for q, u in zip(pad, xs[2:]))   # centre block of the padded frame
    xp = np.zeros(xs[:2] + tuple(u + 2 * q for u, q in zip(xs[2:], pad)), np.float32)
    xp[sl] = x
    oshape = O.conv_out_shape(xp.shape, ws, s, d)
    go = (rng.random(oshape, dtype=np.float32) * 2 - 1)
    bias = (rng.random((ws[0],) + (1,) * case["nd"], dtype=np.float32) * 2 - 1)
    K = ws[1] * int(np.prod(ws[2:]))
    XP, W, G = dev.array(xp), dev.array(w), dev.array(go)
    # forward, plain and with the bias in the epilogue
    Y, YB = dev.full(oshape, 7.0), dev.full(oshape, 7.0)
    c.conv_fwd(dev, XP, W, Y, s, d, g)
    c.conv_fwd(dev, XP, W, YB, s, d, g, bias=dev.array(bias))
    y32 = np.zeros(oshape, np.float32); O.convolution_forward(xp, w, y32, s, d, g)
    y64 = np.zeros(oshape, np.float64); O.convolution_forward(xp.astype(np.float64), w.astype(np.float64), y64, s, d, g)
    contraction_ok(Y.numpy(), y32, y64, K, 1.0, 1.0)
    assert np.array_equal(YB.numpy(), (Y.numpy() + bias).astype(np.float32))
    # backward w.r.t. the padded input and the kernel: `+=` on random contents, first-write on stale contents
    dxp0, dw0 = rng.random(xp.shape, dtype=np.float32), rng.random(ws, dtype=np.float32)
    DXP, DW = dev.array(dxp0), dev.array(dw0)
    c.conv_bwd_input(dev, DXP, G, W, s, d, g)
    c.conv_bwd_kernel(dev, DW, G, XP, s, d, g)
    dxp32, dw32 = np.zeros(xp.shape, np.float32), np.zeros(ws, np.float32)
    O.convolution_backward_input(dxp32, go, w, s, d, g); O.convolution_backward_kernel(dw32, go, xp, s, d, g)
    dxp64, dw64 = np.zeros(xp.shape, np.float64), np.zeros(ws, np.float64)
    O.convolution_backward_input(dxp64, go.astype(np.float64), w.astype(np.float64), s, d, g)
    O.convolution_backward_kernel(dw64, go.astype(np.float64), xp.astype(np.float64), s, d, g)
    Kin = ws[0] // g * int(np.prod(ws[2:]))
    Kk = xs[0] * int(np.prod(oshape[2:]))
    A, B = dev.array(rng.random(xp.shape, dtype=np.float32)), dev.array(rng.random(ws, dtype=np.float32))
    c.conv_bwd_input(dev, A, G, W, s, d, g, assign=True)
    c.conv_bwd_kernel(dev, B, G, XP, s, d, g, assign=True)
    contraction_ok(A.numpy(), dxp32, dxp64, Kin, 1.0, 1.0)
    contraction_ok(B.numpy(), dw32, dw64, Kk, 1.0, 1.0)
    assert np.array_equal(DXP.numpy(), (dxp0 + A.numpy()).astype(np.float32))     # `+=` == old + first-write value
    assert np.array_equal(DW.numpy(), (dw0 + B.numpy()).astype(np.float32))
    # backward-input with the crop of the zero padding folded in: the centre block of the padded gradient, same bits
    C = dev.array(rng.random(xs, dtype=np.float32))
    c.conv_bwd_input(dev, C, G, W, s, d, g, assign=True, padding=pad)
    assert np.array_equal(C.numpy(), A.numpy()[sl])
