"""GPU parity tests: every C-ABI entry point of the hot path against the CPU oracle, on the
reference's own golden vectors and on seeded random inputs.  All calls go through the C ABI
(neuronika_amd.capi -> libneuronika_hip.so); there is no fallback path.

Tolerance policy:
  * reference fixtures: the reference's own |d| <= 4.88e-4 (utils.rs:500); exact equality for
    the integer-valued convolution fixtures and for data movement / masks;
  * random contractions of length K: the ONE bound of tests/tolerance.py (SURVEY.md 8c ii with the chain-length
    factor of DESIGN.md section 5), both f32 results measured against the f64 oracle;
  * elementwise / softmax: rtol 1e-5, atol 1e-6.
"""
import numpy as np
import pytest

from oracle import neuronika_oracle as O

pytestmark = pytest.mark.gpu

F16_EPSILON = 4.88e-4


def capi():
    from neuronika_amd import capi as c
    return c


def f32(v, shape=None):
    a = np.asarray(v, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


def rnd(seed, shape, lo=0.0, hi=1.0):
    a = np.asarray(np.random.default_rng(seed).random(shape, dtype=np.float32), dtype=np.float32)
    return np.asarray(a * np.float32(hi - lo) + np.float32(lo), dtype=np.float32).reshape(shape)


def close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def contraction_ok(gpu, cpu32, ref64, K, amax, bmax, label=None):
    """the one contraction policy of the suite: tests/tolerance.py"""
    from tolerance import assert_contraction
    assert_contraction(label, gpu, ref64, K, amax, bmax, cpu32=cpu32)


# ------------------------------------------------------------------------------ golden: conv
CONV = ["conv1d", "conv2d", "conv3d", "conv1d_strided", "conv2d_strided", "conv3d_strided",
        "conv1d_dilated", "conv2d_dilated", "conv3d_dilated",
        "grouped_conv1d", "grouped_conv2d", "grouped_conv3d"]


@pytest.mark.parametrize("name", CONV)
def test_conv_golden_exact(dev, golden, name):
    c = capi()
    case = next(k for k in golden["convolution"] if k["name"] == name)
    x = np.arange(case["input"]["arange"], dtype=np.float32).reshape(case["input"]["shape"])
    w = np.ones(case["kernel"]["ones"], dtype=np.float32)
    s, d, g = case["stride"], case["dilation"], case["groups"]
    oshape = O.conv_out_shape(x.shape, w.shape, s, d)
    X, W, Y = dev.array(x), dev.array(w), dev.full(oshape, 123.0)   # forward must OVERWRITE
    c.conv_fwd(dev, X, W, Y, s, d, g)
    assert np.array_equal(Y.numpy(), f32(case["output"], oshape)), case["cite"]
    G = dev.full(oshape, 1.0)
    DX, DW = dev.zeros(x.shape), dev.zeros(w.shape)
    c.conv_bwd_input(dev, DX, G, W, s, d, g)
    c.conv_bwd_kernel(dev, DW, G, X, s, d, g)
    assert np.array_equal(DX.numpy(), f32(case["input_grad"], x.shape)), case["cite"]
    assert np.array_equal(DW.numpy(), f32(case["kernel_grad"], w.shape)), case["cite"]
    c.conv_bwd_input(dev, DX, G, W, s, d, g)      # backward ACCUMULATES
    c.conv_bwd_kernel(dev, DW, G, X, s, d, g)
    assert np.array_equal(DX.numpy(), 2 * f32(case["input_grad"], x.shape))
    assert np.array_equal(DW.numpy(), 2 * f32(case["kernel_grad"], w.shape))


CONV_RANDOM = [
    # x shape, w shape, stride, dilation, groups
    ((2, 8, 10, 10), (16, 8, 3, 3), (1, 1), (1, 1), 1),
    ((3, 6, 11, 9), (8, 3, 3, 2), (2, 1), (1, 2), 2),
    ((2, 4, 20), (6, 4, 5), (3,), (2,), 1),
    ((1, 4, 6, 7, 8), (4, 2, 2, 3, 2), (1, 2, 1), (2, 1, 2), 2),
    ((4, 130, 9, 9), (140, 130, 3, 3), (1, 1), (1, 1), 1),       # > one 128 tile in M and K
    ((2, 64, 16, 16), (128, 64, 3, 3), (1, 1), (1, 1), 1),        # fast bwd-input (tap-major), generic fwd
    ((2, 64, 18, 18), (128, 64, 3, 3), (1, 1), (1, 1), 1),        # C3-shaped: fast fwd + fast bwd-input
    ((3, 64, 10, 14), (64, 32, 3, 3), (1, 1), (1, 1), 2),         # grouped fast paths (Cg = Mg = 32)
    ((2, 32, 12, 16), (96, 32, 3, 3), (1, 1), (2, 2), 1),         # dilated fast paths
    ((2, 32, 11, 14), (64, 32, 3, 3), (2, 1), (1, 1), 1),         # fast fwd (unit stride on W only), generic bwd-input
    ((2, 32, 40), (32, 32, 5), (1,), (1,), 1),                    # 1-d fast
    ((1, 32, 4, 6, 10), (32, 32, 2, 3, 3), (1, 1, 1), (1, 1, 1), 1),  # 3-d fast
    ((5, 32, 9, 9), (32, 32, 2, 2), (1, 1), (1, 1), 1),           # columns not a multiple of 128 / quads at the tail
    ((2, 32, 7, 13), (32, 32, 3, 5), (1, 1), (1, 1), 1),          # row length 13 (padded to 16), taps shifted by up to 4
    ((3, 32, 6, 11), (64, 32, 2, 3), (1, 1), (1, 3), 1),          # dilation 3 on the innermost axis, width 11
    ((2, 32, 5, 6, 7), (32, 32, 2, 2, 3), (1, 1, 1), (1, 2, 1), 1),   # 3-d, width 7
    ((2, 32, 23), (32, 32, 4), (1,), (2,), 1),                    # 1-d, width 23, dilation 2
    ((2, 32, 6, 7), (32, 32, 3, 3), (1, 1), (1, 1), 1),           # output 4 x 5: row-padded quads, 3 dummy columns per row
    ((2, 32, 5, 5), (32, 32, 3, 3), (1, 1), (1, 1), 1),           # output 3 x 3: rows shorter than a quad -> generic paths
    ((2, 32, 9, 12), (32, 32, 3, 3), (1, 2), (1, 1), 1),          # stride on the innermost axis: scalar gathers
    ((3, 32, 16, 16), (64, 32, 3, 3), (1, 1), (1, 1), 1),         # output 14 x 14 (ResNet-style), row padded to 16
    ((2, 64, 9, 9), (64, 64, 3, 3), (1, 1), (1, 1), 1),           # output 7 x 7, row padded to 8
    ((2, 32, 13, 14), (32, 32, 3, 3), (2, 2), (1, 1), 1),         # stride 2 x 2: four backward-input phases (4/2/2/1 taps)
    ((2, 32, 17, 19), (32, 32, 3, 3), (2, 2), (2, 2), 1),         # ... dilation 2: every tap in phase (0,0), three phases write zeros
    ((2, 32, 12, 11), (64, 32, 1, 1), (2, 2), (1, 1), 1),         # 1 x 1 kernel, stride 2 (projection shortcut)
    ((2, 32, 40), (32, 32, 5), (3,), (1,), 1),                    # 1-d stride 3
    ((1, 32, 9, 10, 11), (32, 32, 2, 3, 3), (2, 2, 2), (1, 1, 1), 1),  # 3-d, 8 phases
    ((2, 32, 21, 22), (32, 32, 5, 5), (4, 4), (1, 1), 1),         # 16 phases
    ((2, 32, 26, 27), (32, 32, 6, 6), (5, 5), (1, 1), 1),         # 25 phases > the table: generic kernel
    ((2, 64, 15, 17), (64, 32, 3, 3), (2, 2), (1, 1), 2),         # grouped, stride 2
    # 17..31 channels per group: the generic implicit-GEMM kernels (<= 16 goes to the direct kernels, multiples of 32 to the fast ones)
    ((2, 20, 10, 14), (24, 20, 3, 3), (1, 1), (1, 1), 1),         # output 8 x 12: vector gathers (QUADV)
    ((2, 20, 10, 13), (24, 20, 3, 3), (1, 1), (1, 1), 1),         # output 8 x 11: scalar gathers
    ((2, 20, 11, 13), (24, 20, 3, 3), (2, 2), (1, 1), 1),         # strided: divisibility tests in the backward-input gather
    ((2, 3, 20, 22), (64, 3, 5, 5), (2, 2), (1, 1), 1),           # stem-shaped: 3 input channels, 64 output channels
    ((3, 40, 9, 9), (40, 20, 3, 3), (1, 1), (1, 1), 2),           # grouped, 20 channels per group both ways
    ((2, 24, 30), (20, 24, 4), (2,), (2,), 1),                    # 1-d strided dilated
    # direct kernels (few channels per group)
    ((2, 8, 9, 10), (8, 1, 3, 3), (1, 1), (1, 1), 8),             # depthwise
    ((2, 8, 9, 10), (16, 1, 3, 3), (2, 1), (1, 2), 8),            # depthwise, channel multiplier 2, stride / dilation
    ((3, 12, 14), (6, 4, 5), (2,), (2,), 3),                      # 1-d grouped
    ((1, 6, 5, 6, 7), (6, 2, 2, 3, 2), (1, 2, 1), (2, 1, 2), 3),  # 3-d grouped
    ((2, 16, 12, 12), (16, 16, 3, 3), (1, 1), (1, 1), 1),         # 16 x 16 channels: the largest direct case
    ((2, 8, 26, 27), (8, 1, 3, 3), (1, 1), (1, 1), 8),            # planes >= 512 positions: four positions per thread
    ((2, 8, 49, 50), (16, 1, 3, 3), (2, 2), (1, 1), 8),           # ... strided
    ((2, 6, 28, 30), (6, 2, 5, 5), (1, 1), (1, 1), 3),            # 5 x 5 taps unrolled
    ((2, 6, 28, 30), (6, 2, 4, 2), (1, 1), (2, 1), 3),            # run-time kernel extents, dilation
    ((2, 8, 14, 18), (8, 1, 3, 3), (1, 1), (1, 1), 8),            # output 12 x 16: row-blocked direct forward (4 outputs per thread)
    ((2, 8, 30, 18), (16, 1, 3, 3), (2, 1), (2, 1), 8),           # ... strided / dilated rows
    ((2, 6, 12, 16), (6, 2, 5, 5), (1, 1), (1, 1), 3),            # ... 5 x 5
    ((2, 8, 18), (8, 1, 3), (1,), (1,), 8),                       # ... 1-d
]


@pytest.mark.parametrize("xs,ws,s,d,g", CONV_RANDOM)
def test_conv_random_vs_oracle(dev, xs, ws, s, d, g):
    c = capi()
    x, w = rnd(0, xs), rnd(1, ws, -1, 1)
    oshape = O.conv_out_shape(xs, ws, s, d)
    go = rnd(2, oshape)
    K = int(np.prod(ws[1:]))
    X, W, Y, G = dev.array(x), dev.array(w), dev.zeros(oshape), dev.array(go)
    c.conv_fwd(dev, X, W, Y, s, d, g)
    y32 = np.zeros(oshape, np.float32); O.convolution_forward(x, w, y32, s, d, g)
    y64 = np.zeros(oshape, np.float64); O.convolution_forward(x.astype(np.float64), w.astype(np.float64), y64, s, d, g)
    contraction_ok(Y.numpy(), y32, y64, K, 1.0, 1.0)
    dx0, dw0 = rnd(3, xs), rnd(4, ws)    # non-zero initial gradients: `+=`
    DX, DW = dev.array(dx0), dev.array(dw0)
    c.conv_bwd_input(dev, DX, G, W, s, d, g)
    c.conv_bwd_kernel(dev, DW, G, X, s, d, g)
    dx32, dw32 = dx0.copy(), dw0.copy()
    O.convolution_backward_input(dx32, go, w, s, d, g); O.convolution_backward_kernel(dw32, go, x, s, d, g)
    dx64, dw64 = dx0.astype(np.float64), dw0.astype(np.float64)
    O.convolution_backward_input(dx64, go.astype(np.float64), w.astype(np.float64), s, d, g)
    O.convolution_backward_kernel(dw64, go.astype(np.float64), x.astype(np.float64), s, d, g)
    contraction_ok(DX.numpy(), dx32, dx64, ws[0] // g * int(np.prod(ws[2:])), 1.0, 1.0)
    contraction_ok(DW.numpy(), dw32, dw64, xs[0] * int(np.prod(oshape[2:])), 1.0, 1.0)


@pytest.mark.parametrize("xs,ws,s,d,g", CONV_RANDOM)
def test_conv_kernel_and_bias_gradient_in_one_pass(dev, xs, ws, s, d, g):
    """nk_conv_bwd_kernel_bias: the kernel gradient is BIT-identical to nk_conv_bwd_kernel's (same pass), the bias gradient -
    AdditionBackwardRight of the module's (Cout,1,..) bias, summed on the way by the implicit-GEMM pass or by the reduction
    behind the scenes for the other kernel families - matches the oracle's un-broadcast sum; `+=` and first-write forms."""
    c = capi()
    x, w = rnd(0, xs), rnd(1, ws, -1, 1)
    oshape = O.conv_out_shape(xs, ws, s, d)
    go = rnd(2, oshape, -1, 1)
    X, G = dev.array(x), dev.array(go)
    dw0, db0 = rnd(4, ws), rnd(5, (ws[0],) + (1,) * (len(xs) - 2))
    ref_dw = dev.array(dw0)
    c.conv_bwd_kernel(dev, ref_dw, G, X, s, d, g)
    DW, DB = dev.array(dw0), dev.array(db0)
    c.conv_bwd_kernel_bias(dev, DW, DB, G, X, s, d, g)
    assert np.array_equal(DW.numpy(), ref_dw.numpy())
    axes = (0,) + tuple(range(2, len(oshape)))
    sum64 = go.astype(np.float64).sum(axis=axes).reshape(db0.shape)
    sum32 = np.zeros(db0.shape, np.float32); O.accumulate(sum32, go)
    n_terms = go.size // ws[0]
    err_gpu, err_cpu = np.abs(DB.numpy() - (db0 + sum64)).max(), np.abs(sum32 - sum64).max()
    assert err_gpu <= max(2 * err_cpu, 1e-6 * n_terms ** 0.5 * np.abs(go).max() + 2e-7 * np.abs(db0 + sum64).max()), (err_gpu, err_cpu)
    DW2, DB2 = dev.array(dw0), dev.array(db0)       # first-write form: the buffers' contents do not matter
    c.conv_bwd_kernel_bias(dev, DW2, DB2, G, X, s, d, g, assign=(True, True))
    ref2 = dev.array(dw0); c.conv_bwd_kernel(dev, ref2, G, X, s, d, g, assign=True)
    assert np.array_equal(DW2.numpy(), ref2.numpy())
    np.testing.assert_allclose(DB2.numpy(), DB.numpy() - db0, rtol=1e-5, atol=1e-5 * np.abs(sum64).max() + 1e-6)


@pytest.mark.parametrize("xs,ws,s,d", [
    ((4, 64, 14, 14), (128, 64, 3, 3), (1, 1), (1, 1)),      # C3-shaped: 576 columns = 4.5 tiles
    ((3, 64, 13, 18), (128, 64, 3, 3), (1, 1), (2, 1)),      # dilated rows, output rows of 16
    ((2, 192, 9, 12), (256, 192, 1, 1), (1, 1), (1, 1)),     # 1 x 1: 192 columns = 1.5 tiles, two tile rows
    ((16, 64, 30, 30), (128, 64, 3, 3), (1, 1), (1, 1)),     # reduction long enough for several k-tiles per split
    ((5, 64, 9, 9), (128, 64, 3, 3), (1, 1), (1, 1)),        # output rows of 7: row-padded quads (the last quad of a row re-reads one element)
    ((3, 64, 12, 11), (192, 64, 3, 3), (2, 1), (1, 1)),      # strided rows, 192 output channels: 64-row tiles, three tile rows
    ((4, 64, 14, 14), (64, 64, 3, 3), (1, 1), (1, 1)),       # 64 -> 64 channels (a ResNet stage-1 layer): 64-row tiles
])
@pytest.mark.parametrize("cost", [40, 70, 100])
def test_conv_kernel_gradient_mixed_launch(dev, xs, ws, s, d, cost):
    """The kernel gradient's mixed launch (conv_bwd_kernel_mixed_kernel: whole column tiles through 128-wide blocks, the last,
    half-empty one through 64-wide blocks with its reduction cut into fewer ranges; NK_TUNE_CONV_NARROW prices a narrow block)
    against the uniform launch and the oracle: dW and the fused bias gradient within the contraction policy, `+=` and first-write
    forms, for several prices (each gives other split counts for the two tile shapes)."""
    c = capi()
    x, go = rnd(0, xs), rnd(2, O.conv_out_shape(xs, ws, s, d), -1, 1)
    X, G = dev.array(x), dev.array(go)
    dw0, db0 = rnd(4, ws), rnd(5, (ws[0], 1, 1))
    dw32, dw64 = dw0.copy(), dw0.astype(np.float64)
    O.convolution_backward_kernel(dw32, go, x, s, d, 1)
    O.convolution_backward_kernel(dw64, go.astype(np.float64), x.astype(np.float64), s, d, 1)
    outs = {}
    for price in (0, cost):
        dev.conv_narrow(price)
        try:
            DW, DB = dev.array(dw0), dev.array(db0)
            c.conv_bwd_kernel_bias(dev, DW, DB, G, X, s, d, 1)
            DW2, DB2 = dev.full(ws, 7.0), dev.full(db0.shape, 7.0)
            c.conv_bwd_kernel_bias(dev, DW2, DB2, G, X, s, d, 1, assign=(True, True))
            outs[price] = (DW.numpy(), DB.numpy(), DW2.numpy(), DB2.numpy())
        finally:
            dev.conv_narrow(None)
    R = xs[0] * int(np.prod(go.shape[2:]))
    for price, (dw, db, dw2, db2) in outs.items():
        contraction_ok(dw, dw32, dw64, R, 1.0, 1.0)
        np.testing.assert_allclose(dw2, dw64 - dw0, rtol=1e-4, atol=2e-6 * R)
        np.testing.assert_allclose(db2.reshape(-1), go.astype(np.float64).sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-5 * R ** 0.5)
        np.testing.assert_allclose(db - db0, db2, rtol=1e-4, atol=1e-5 * R ** 0.5)
    np.testing.assert_allclose(outs[cost][0], outs[0][0], rtol=1e-4, atol=2e-6 * R)   # other split counts: other rounding, same sums


CONV_PADDED = [
    # unpadded x shape, w shape, padding, stride, dilation, groups
    ((2, 64, 16, 16), (128, 64, 3, 3), (1, 1), (1, 1), (1, 1), 1),     # C3-shaped, fast kernel
    ((3, 64, 10, 14), (64, 32, 3, 3), (1, 1), (1, 1), (1, 1), 2),      # grouped, fast kernel
    ((2, 32, 7, 13), (32, 32, 3, 5), (2, 1), (1, 1), (1, 1), 1),       # width 13 -> row padded to 16, asymmetric pads
    ((2, 32, 6, 9), (32, 32, 3, 3), (3, 4), (1, 1), (2, 2), 1),        # padding wider than the dilated kernel reach
    ((2, 32, 30), (32, 32, 5), (2,), (1,), (1,), 1),                   # 1-d fast
    ((1, 32, 4, 6, 10), (32, 32, 2, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 1),  # 3-d fast
    ((2, 8, 10, 10), (16, 8, 3, 3), (1, 1), (1, 1), (1, 1), 1),        # generic kernel
    ((3, 6, 11, 9), (8, 3, 3, 2), (2, 1), (2, 1), (1, 2), 2),          # generic, stride, dilation, groups
    ((2, 4, 20), (6, 4, 5), (3,), (3,), (2,), 1),
    ((1, 4, 6, 7, 8), (4, 2, 2, 3, 2), (0, 2, 1), (1, 2, 1), (2, 1, 2), 2),
    ((2, 32, 8, 8), (32, 32, 3, 3), (0, 0), (1, 1), (1, 1), 1),        # zero padding == the unpadded entry point
    ((2, 32, 12, 13), (32, 32, 3, 3), (1, 1), (2, 2), (1, 1), 1),      # stride 2 with the crop: phases start at odd coordinates
    ((2, 32, 12, 13), (32, 32, 3, 3), (2, 1), (2, 2), (1, 1), 1),
    ((2, 32, 14, 15), (64, 32, 7, 7), (3, 3), (2, 2), (1, 1), 1),      # 7 x 7 stride 2 pad 3 (stem-shaped)
    ((2, 32, 30), (32, 32, 4), (2,), (3,), (2,), 1),                   # 1-d stride 3 dilation 2
    ((1, 32, 6, 7, 9), (32, 32, 3, 3, 3), (1, 1, 1), (2, 1, 2), (1, 1, 1), 1),   # 3-d mixed strides
    ((2, 8, 9, 10), (8, 1, 3, 3), (1, 1), (1, 1), (1, 1), 8),          # depthwise (direct kernel) with the crop
    ((2, 8, 9, 10), (16, 1, 3, 3), (2, 1), (2, 2), (1, 1), 8),
    ((2, 20, 9, 10), (24, 20, 3, 3), (1, 2), (1, 1), (1, 1), 1),       # generic kernel, 20 -> 24 channels
    ((2, 20, 9, 10), (24, 20, 3, 3), (1, 1), (2, 1), (1, 1), 1),
    ((2, 8, 24, 25), (8, 1, 3, 3), (1, 1), (1, 1), (1, 1), 8),         # direct kernel, planes >= 512 positions
    ((2, 8, 47, 48), (8, 1, 3, 3), (1, 1), (2, 2), (1, 1), 8),
    ((2, 6, 26, 28), (6, 2, 5, 5), (2, 2), (1, 1), (1, 1), 3),
    ((2, 8, 12, 16), (8, 1, 3, 3), (1, 1), (1, 1), (1, 1), 8),         # row-blocked direct backward-input (width % 4 == 0)
    ((2, 8, 12, 16), (8, 1, 3, 3), (2, 2), (1, 1), (2, 1), 8),         # ... dilated rows, wider padding
    ((2, 6, 12, 16), (6, 2, 5, 5), (2, 2), (1, 1), (1, 1), 3),         # ... 5 x 5
    ((2, 8, 20), (8, 1, 3), (1,), (1,), (1,), 8),                      # ... 1-d
    ((2, 8, 12, 16), (16, 1, 3, 3), (0, 0), (1, 1), (1, 1), 8),        # ... no padding (borders masked)
]


@pytest.mark.parametrize("xs,ws,pad,s,d,g", CONV_PADDED)
def test_conv_bwd_input_padded_vs_oracle(dev, xs, ws, pad, s, d, g):
    """nk_conv_bwd_input_padded == ConvolutionBackwardInput on the padded shape followed by PadBackward (centre block),
    bit-identical to our own two-kernel form, `+=` and first-write variants."""
    c = capi()
    w = rnd(1, ws, -1, 1)
    ps = tuple(xs[:2]) + tuple(n + 2 * p for n, p in zip(xs[2:], pad))
    oshape = O.conv_out_shape(ps, ws, s, d)
    go = rnd(2, oshape)
    W, G = dev.array(w), dev.array(go)
    dx0 = rnd(3, xs)
    DX = dev.array(dx0)
    c.conv_bwd_input(dev, DX, G, W, s, d, g, padding=pad)
    # oracle: gradient of the padded input, centre block accumulated
    dxp32 = np.zeros(ps, np.float32); O.convolution_backward_input(dxp32, go, w, s, d, g)
    dxp64 = np.zeros(ps, np.float64); O.convolution_backward_input(dxp64, go.astype(np.float64), w.astype(np.float64), s, d, g)
    dx32 = dx0.copy(); O.pad_backward(dx32, dxp32, pad)
    dx64 = dx0.astype(np.float64); O.pad_backward(dx64, dxp64, pad)
    contraction_ok(DX.numpy(), dx32, dx64, ws[0] // g * int(np.prod(ws[2:])), 1.0, 1.0)
    # our two-kernel form: same k order per element -> identical bits
    DXP = dev.zeros(ps)
    c.conv_bwd_input(dev, DXP, G, W, s, d, g, assign=True)
    DX2 = dev.array(dx0)
    c.pad_bwd(dev, DX2, DXP, pad)
    assert np.array_equal(DX.numpy(), DX2.numpy())
    A = dev.array(rnd(5, xs))                     # stale contents must be overwritten
    c.conv_bwd_input(dev, A, G, W, s, d, g, assign=True, padding=pad)
    Z = dev.zeros(xs)
    c.conv_bwd_input(dev, Z, G, W, s, d, g, padding=pad)
    assert np.array_equal(A.numpy(), Z.numpy())


def test_conv_arg_errors(dev):
    c = capi()
    X, W, Y = dev.zeros((1, 2, 4, 4)), dev.zeros((1, 2, 5, 5)), dev.zeros((1, 1, 1, 1))
    with pytest.raises(c.NeuronikaHipError, match="kernel size can't be greater"):
        c.conv_fwd(dev, X, W, Y, (1, 1), (1, 1), 1)
    X, W = dev.zeros((3, 3, 10, 10)), dev.zeros((3, 3, 3, 3))
    with pytest.raises(c.NeuronikaHipError, match="not divisible by groups"):
        c.conv_fwd(dev, X, W, dev.zeros((3, 3, 8, 8)), (1, 1), (1, 1), 5)


# ------------------------------------------------------------------------------ golden: matmul
def test_mm_golden(dev, golden):
    c = capi()
    g = golden["nodes"]["mm_backward"]
    right = np.linspace(10.0, 18.0, 9, dtype=np.float32).reshape(3, 3)
    left = np.linspace(1.0, 9.0, 9, dtype=np.float32).reshape(3, 3)
    G = dev.full((3, 3), 1.0)
    DA, DB = dev.zeros((3, 3)), dev.zeros((3, 3))
    R, L = dev.array(right), dev.array(left)
    c.mm_bwd_left(dev, DA, G, R); close(DA.numpy(), f32(g["left_bwd"]["left_grad_once"], (3, 3)), 0, F16_EPSILON)
    c.mm_bwd_left(dev, DA, G, R); close(DA.numpy(), f32(g["left_bwd"]["left_grad_twice"], (3, 3)), 0, F16_EPSILON)
    c.mm_bwd_right(dev, DB, L, G); close(DB.numpy(), f32(g["right_bwd"]["right_grad_once"], (3, 3)), 0, F16_EPSILON)
    c.mm_bwd_right(dev, DB, L, G); close(DB.numpy(), f32(g["right_bwd"]["right_grad_twice"], (3, 3)), 0, F16_EPSILON)
    out = dev.full((3, 3), 7.0)
    c.mm_fwd(dev, L, R, out); close(out.numpy(), left @ right, 0, F16_EPSILON)   # overwrite, beta = 0


def test_mm_t_golden(dev, golden):
    c = capi()
    g = golden["nodes"]["mm_t_forward"]
    A, B = dev.array(f32(g["left"], g["left_shape"])), dev.array(f32(g["right"], g["right_shape"]))
    out = dev.full(g["out_shape"], -3.0)
    c.mm_t_fwd(dev, A, B, out)
    close(out.numpy(), f32(g["out"], g["out_shape"]), 0, F16_EPSILON)
    lit = golden["nodes"]["mm_t_backward"]["literals"]
    A, B, G = dev.array(f32(lit[2], (3, 3))), dev.array(f32(lit[3], (2, 3))), dev.array(f32(lit[4], (3, 2)))
    DA, DB = dev.zeros((3, 3)), dev.zeros((2, 3))
    c.mm_t_bwd_left(dev, DA, G, B); c.mm_t_bwd_right(dev, DB, G, A)
    close(DA.numpy(), f32(lit[6], (3, 3)), 0, F16_EPSILON); close(DB.numpy(), f32(lit[7], (2, 3)), 0, F16_EPSILON)
    c.mm_t_bwd_left(dev, DA, G, B); c.mm_t_bwd_right(dev, DB, G, A)
    close(DA.numpy(), f32(lit[8], (3, 3)), 0, F16_EPSILON); close(DB.numpy(), f32(lit[9], (2, 3)), 0, F16_EPSILON)


GEMM_SHAPES = [(3, 5, 7), (257, 131, 77), (128, 128, 32), (256, 384, 160), (64, 200, 1000), (128, 128, 4096),
               (1, 1, 1), (130, 4, 33)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_sgemm_random(dev, M, N, K, ta, tb):
    """Asymmetric random operands (a transposed C-write or operand would not pass)."""
    c = capi()
    a = rnd(10, (K, M) if ta else (M, K), -1, 1)
    b = rnd(11, (N, K) if tb else (K, N), -1, 1)
    c0 = rnd(12, (M, N))
    opa, opb = (a.T if ta else a), (b.T if tb else b)
    A, B, Cd = dev.array(a), dev.array(b), dev.array(c0)
    c.sgemm(dev, ta, tb, M, N, K, 1.0, A, a.shape[1], B, b.shape[1], 0.0, Cd, N)
    ref64 = opa.astype(np.float64) @ opb.astype(np.float64)
    contraction_ok(Cd.numpy(), opa @ opb, ref64, K, 1.0, 1.0)
    c.sgemm(dev, ta, tb, M, N, K, 0.5, A, a.shape[1], B, b.shape[1], 1.0, Cd, N)   # accumulate
    contraction_ok(Cd.numpy(), (opa @ opb) * 1.5, ref64 * 1.5, K, 1.5, 1.0)


HEUR_SHAPES = [  # one or more shapes per branch of the tile / split-K selection (nk_gemm.hip), all four layouts cycled
    (3072, 3072, 160), (3072, 1536, 96),      # >= 512 blocks of 128x128 but a ragged last wave -> smaller tiles
    (1536, 1536, 1536), (1024, 1024, 8192),   # < 512 blocks: 128x128 kept with split-K (>= 16 k-tiles per split)
    (512, 512, 4096), (256, 256, 2048), (768, 768, 768), (64, 2048, 4096), (2048, 64, 1024),   # 64x64 (+ split)
    (192, 320, 1100), (129, 65, 515), (1000, 1000, 1000), (33, 4097, 70), (4099, 31, 260),     # ragged / unaligned
    (2048, 2048, 96), (640, 640, 640), (1, 1, 1), (1, 700, 3), (700, 1, 5000)]


@pytest.mark.parametrize("idx", range(len(HEUR_SHAPES)))
def test_sgemm_heuristic_branches(dev, idx):
    """Every (tile, split-K) selection path gives the same product: checked against f64 at sampled rows/cols (full
    reference for the small ones), beta = 0 then beta = 1, layout cycling NN/NT/TN/TT with the shape index."""
    c = capi()
    M, N, K = HEUR_SHAPES[idx]
    ta, tb = (idx >> 1) & 1, idx & 1
    a = rnd(30 + idx, (K, M) if ta else (M, K), -1, 1)
    b = rnd(60 + idx, (N, K) if tb else (K, N), -1, 1)
    opa, opb = (a.T if ta else a).astype(np.float64), (b.T if tb else b).astype(np.float64)
    A, B, Cd = dev.array(a), dev.array(b), dev.full((M, N), 7.0)
    c.sgemm(dev, ta, tb, M, N, K, 1.0, A, a.shape[1], B, b.shape[1], 0.0, Cd, N)
    got = Cd.numpy().astype(np.float64)
    rows = np.unique(np.r_[0, M - 1, np.random.default_rng(idx).integers(0, M, 24)])
    cols = np.unique(np.r_[0, N - 1, np.random.default_rng(idx + 1).integers(0, N, 24)])
    from tolerance import assert_contraction
    a32, b32 = (a.T if ta else a), (b.T if tb else b)
    rows32, cols32 = a32[rows] @ b32, a32 @ b32[:, cols]                                 # the f32 CPU restatement (OpenBLAS)
    assert_contraction("sgemm_heuristic_branches", got[rows], opa[rows] @ opb, K, cpu32=rows32)
    assert_contraction("sgemm_heuristic_branches", got[:, cols], opa @ opb[:, cols], K, cpu32=cols32)
    c.sgemm(dev, ta, tb, M, N, K, -0.5, A, a.shape[1], B, b.shape[1], 1.0, Cd, N)       # accumulate: 0.5 * product
    got2 = Cd.numpy().astype(np.float64)
    assert_contraction("sgemm_heuristic_branches", got2[rows], 0.5 * (opa[rows] @ opb), K, cpu32=np.float32(0.5) * rows32, scale=0.5, epilogue=True)
    assert_contraction("sgemm_heuristic_branches", got2[:, cols], 0.5 * (opa @ opb[:, cols]), K, cpu32=np.float32(0.5) * cols32, scale=0.5, epilogue=True)


@pytest.mark.parametrize("n,m,o", [(64, 8192, 64), (256, 4096, 512), (1536, 1536, 1536), (3072, 128, 3072)])
def test_linear_fwd_split_and_small_tiles(dev, n, m, o):
    """nk_linear_fwd on shapes that take the split-K second pass (bias applied there) and the 64-wide tiles."""
    c = capi()
    x, w, b = rnd(1, (n, m), -1, 1), rnd(2, (o, m), -1, 1), rnd(3, (o,), -1, 1)
    X, W, Bv = dev.array(x), dev.array(w), dev.array(b)
    Y0, Y1, Y2 = dev.zeros((n, o)), dev.zeros((n, o)), dev.full((n, o), 3.0)
    c.mm_t_fwd(dev, X, W, Y0)
    c.binary_fwd(dev, "add", Y1, Y0, Bv)
    c.linear_fwd(dev, X, W, Bv, Y2)
    assert np.array_equal(Y1.numpy(), Y2.numpy())
    rows = np.random.default_rng(0).integers(0, n, 16)
    ref = x[rows].astype(np.float64) @ w.astype(np.float64).T + b
    from tolerance import assert_contraction
    assert_contraction("linear_fwd_split_and_small_tiles", Y2.numpy()[rows], ref, m, cpu32=x[rows] @ w.T + b, epilogue=True)


def test_sgemm_batched_strided(dev):
    """Two-level batch with the attention strides: Q_bh is a strided view of (B*S, H*dh)."""
    c = capi()
    B_, S, H, dh = 2, 96, 3, 32
    d = H * dh
    q, k = rnd(20, (B_ * S, d), -1, 1), rnd(21, (B_ * S, d), -1, 1)
    Q, Kd, Sc = dev.array(q), dev.array(k), dev.zeros((B_ * H, S, S))
    c.sgemm_batched(dev, 0, 1, S, S, dh, 1.0, Q, d, S * d, dh, Kd, d, S * d, dh, 0.0, Sc, S, H * S * S, S * S, B_, H)
    qh = q.reshape(B_, S, H, dh).transpose(0, 2, 1, 3).reshape(B_ * H, S, dh).astype(np.float64)
    kh = k.reshape(B_, S, H, dh).transpose(0, 2, 1, 3).reshape(B_ * H, S, dh).astype(np.float64)
    ref = qh @ kh.transpose(0, 2, 1)
    contraction_ok(Sc.numpy(), ref.astype(np.float32), ref, dh, 1.0, 1.0)


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("K,chunk,splits", [(64, 2, 1), (64, 5, 1), (96, 3, 1), (160, 16, 1), (256, 3, 2), (64, 64, 1)])
def test_sgemm_tile_chunks(dev, ta, tb, K, chunk, splits):
    """Short reductions: a block walks `chunk` consecutive tiles, prefetching the next tile's first k-tile behind the
    last MFMA block of the current one (nk_gemm.hip).  Must give the SAME BITS as one tile per block (same products,
    same k order) for every layout and tile shape, batch strides, alpha / beta, chunks that do not divide the tile
    count, and split-K slabs; and the result matches the f64 oracle."""
    import os
    c = capi()
    M, N, nb_o, nb_i = 512, 384, 3, 2
    nb = nb_o * nb_i
    a = rnd(40, (nb, K, M) if ta else (nb, M, K), -1, 1)
    b = rnd(41, (nb, N, K) if tb else (nb, K, N), -1, 1)
    c0 = rnd(42, (nb, M, N), -1, 1)
    A, B = dev.array(a), dev.array(b)
    lda, ldb = a.shape[2], b.shape[2]
    sa, sb, sc = a.shape[1] * lda, b.shape[1] * ldb, M * N
    outs = {}
    try:
        for tiles in ("2,2", "1,2"):
            for ch in (1, chunk):
                for alpha, beta in ((1.0, 0.0), (-1.5, 0.5)):
                    dev.gemm_force(f"{tiles},{splits},{ch}")
                    Cd = dev.array(c0)
                    c.sgemm_batched(dev, ta, tb, M, N, K, alpha, A, lda, nb_i * sa, sa, B, ldb, nb_i * sb, sb, beta, Cd, N, nb_i * sc, sc, nb_o, nb_i)
                    outs[(tiles, ch, alpha)] = Cd.numpy()
    finally:
        dev.gemm_force(None)
    opa = (a.transpose(0, 2, 1) if ta else a).astype(np.float64)
    opb = (b.transpose(0, 2, 1) if tb else b).astype(np.float64)
    ref = opa @ opb
    for tiles in ("2,2", "1,2"):
        for alpha, beta in ((1.0, 0.0), (-1.5, 0.5)):
            assert np.array_equal(outs[(tiles, 1, alpha)], outs[(tiles, chunk, alpha)]), (tiles, alpha)
            want = alpha * ref + beta * c0
            contraction_ok(outs[(tiles, chunk, alpha)], want.astype(np.float32), want, K, 1.5, 1.0)


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("tiles", ["1,1", "2,2", "1,2", "2,1"])
def test_sgemm_lookahead_loop_is_bit_identical(dev, ta, tb, tiles):
    """The two-k-tile look-ahead loop (taken per layout / tile shape from a k-tile threshold on, nk_gemm.hip) and the
    one-k-tile loop accumulate in the same order: forcing either for every reduction length from 1 to 7 k-tiles (all
    the loop's tail cases) and around the thresholds gives the same bits, for every layout and tile shape, with
    alpha / beta; and the result matches the f64 oracle."""
    import os
    c = capi()
    M, N = 256, 384
    try:
        for kt in (1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 47, 48, 49):
            K = 32 * kt
            a = rnd(60 + kt, (K, M) if ta else (M, K), -1, 1)
            b = rnd(61 + kt, (N, K) if tb else (K, N), -1, 1)
            c0 = rnd(62, (M, N), -1, 1)
            A, B = dev.array(a), dev.array(b)
            outs = {}
            for la in (1, 9999):
                dev.gemm_force(f"{tiles},1,1,8,{la}")
                Cd = dev.array(c0)
                c.sgemm(dev, ta, tb, M, N, K, -1.5, A, a.shape[1], B, b.shape[1], 0.5, Cd, N)
                outs[la] = Cd.numpy()
            assert np.array_equal(outs[1], outs[9999]), (kt,)
            if kt in (1, 7, 49):
                opa, opb = (a.T if ta else a).astype(np.float64), (b.T if tb else b).astype(np.float64)
                want = -1.5 * (opa @ opb) + 0.5 * c0
                contraction_ok(outs[1], want.astype(np.float32), want, K, 1.5, 1.0)
    finally:
        dev.gemm_force(None)


def test_gemm_override_variable_is_live(dev):
    """The schedule tests above flip the device handle's NK_TUNE_GEMM_FORCE override (nk_dev_tune) between calls of ONE process: it
    must take effect at once and go away again.  Split-K 4 sums in another order than the unsplit product - different bits on
    random data.  The library itself reads no environment variable: setting NK_GEMM_FORCE in the environment of a running
    process changes nothing."""
    import os
    c = capi()
    M = N = 256
    K = 1024
    a, b = rnd(70, (M, K), -1, 1), rnd(71, (N, K), -1, 1)
    A, B = dev.array(a), dev.array(b)
    outs = {}
    try:
        for force in ("2,2,1", "2,2,4", "2,2,1"):
            dev.gemm_force(force)
            Cd = dev.zeros((M, N))
            c.sgemm(dev, 0, 1, M, N, K, 1.0, A, K, B, K, 0.0, Cd, N)
            outs.setdefault(force, []).append(Cd.numpy())
    finally:
        dev.gemm_force(None)
    assert np.array_equal(outs["2,2,1"][0], outs["2,2,1"][1])
    assert not np.array_equal(outs["2,2,1"][0], outs["2,2,4"][0])
    os.environ["NK_GEMM_FORCE"] = "2,2,2"                     # the environment of a running process: not consulted
    try:                                                      # (the rules split this shape 4 ways: split 2 gives other bits)
        Cd = dev.zeros((M, N))
        c.sgemm(dev, 0, 1, M, N, K, 1.0, A, K, B, K, 0.0, Cd, N)
        rules = Cd.numpy()
        dev.gemm_force("2,2,2")
        c.sgemm(dev, 0, 1, M, N, K, 1.0, A, K, B, K, 0.0, Cd, N)
        assert not np.array_equal(rules, Cd.numpy())
    finally:
        os.environ.pop("NK_GEMM_FORCE", None)
        dev.gemm_force(None)


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_sgemm_kpair_blocks(dev, ta, tb):
    """k-pair blocks (nk_gemm.hip, KG = 2: 512 threads, two wave groups on the two halves of the reduction, accumulators added
    through LDS): group 0 + group 1 is the sum split-K 2's second pass forms, so the result must be BIT-identical to the
    forced split-K 2 launch of plain blocks - for every layout, reductions whose halves hit every tail case of the
    look-ahead loop, batches, alpha / beta; lock-step and skewed groups give the same bits; and it matches the f64 oracle."""
    import os
    c = capi()
    M, N, nb = 256, 384, 2
    try:
        for K in (256, 320, 384, 448, 512, 2048):
            a = rnd(80 + K, (nb, K, M) if ta else (nb, M, K), -1, 1)
            b = rnd(81 + K, (nb, N, K) if tb else (nb, K, N), -1, 1)
            c0 = rnd(82, (nb, M, N), -1, 1)
            A, B = dev.array(a), dev.array(b)
            lda, ldb = a.shape[2], b.shape[2]
            sa, sb, sc = a.shape[1] * lda, b.shape[1] * ldb, M * N
            for alpha, beta in ((1.0, 0.0), (-1.5, 0.5)):
                outs = {}
                for name, force, pair in (("split2", "2,2,2", "0"), ("pair", "2,2,1", "1"), ("pair_skewed", "2,2,1", "2"),
                                          ("pair_skewed_again", "2,2,1", "2"), ("split2_pair", "2,2,2", "2"),
                                          ("split2_64", "1,1,2", "0"), ("pair_64", "1,1,1", "1"), ("pair_64_skewed", "1,1,1", "2")):
                    if name == "split2_pair" and (K // 32) % 4 != 0:
                        continue
                    dev.gemm_force(force); dev.gemm_kpair(int(pair))
                    Cd = dev.array(c0)
                    c.sgemm_batched(dev, ta, tb, M, N, K, alpha, A, lda, 0, sa, B, ldb, 0, sb, beta, Cd, N, 0, sc, 1, nb)
                    outs[name] = Cd.numpy()
                assert np.array_equal(outs["pair"], outs["split2"]), (K, alpha)
                assert np.array_equal(outs["pair_skewed"], outs["split2"]), (K, alpha)
                assert np.array_equal(outs["pair_skewed_again"], outs["pair_skewed"]), (K, alpha)
                # 64 x 64 tiles (group 1 hands its accumulators to group 0): the same sums, whatever the tile shape
                assert np.array_equal(outs["split2_64"], outs["split2"]), (K, alpha)
                assert np.array_equal(outs["pair_64"], outs["split2"]) and np.array_equal(outs["pair_64_skewed"], outs["split2"]), (K, alpha)
                opa = (a.transpose(0, 2, 1) if ta else a).astype(np.float64)
                opb = (b.transpose(0, 2, 1) if tb else b).astype(np.float64)
                want = alpha * (opa @ opb) + beta * c0
                for name in ("pair_skewed", "split2_pair"):
                    if name in outs:
                        contraction_ok(outs[name], want.astype(np.float32), want, K, 1.5, 1.0)
    finally:
        dev.gemm_force(None); dev.gemm_kpair(None)


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_sgemm_is_the_device_order_model_bit_for_bit(dev, ta, tb):
    """The summation order of `sgemm_kernel` is a CONTRACT (nk_gemm.hip): every output is ONE f32 fma
    chain over the block's k range in the MFMA feeding order (inside each group of 8 k: k, k+4, k+1, k+5, ...; the MFMA
    itself is an exact fmaf chain), split-K adds the splits' chains in split order starting from 0, a k-pair block adds
    its two halves.  `oracle/device_order_sgemm.c` restates that order on the CPU and the device result must equal it BIT
    FOR BIT - every layout, every tile shape, both k-loops (two-k-tile look-ahead with each tail length, the one-k-tile loop
    for unaligned K and forced by the look-ahead threshold), split-K, k-pair blocks.  It is what makes the error analysis of
    tools/c4_tolerance_model.py (chain length vs. the parity bound) a statement about the device and not about a model."""
    import os
    from oracle.build_c import sgemm_device_order
    c = capi()
    M, N = 128, 256
    KC = 0
    try:
        for K, force, pair in ((2048, "2,2,1", "0"), (2080, "2,2,1", "0"), (2112, "2,2,1", "0"), (2144, "2,2,1", "0"), (4096, "2,2,1", "0"),
                               (6333, "2,2,1", "0"), (2100, "1,1,1", "0"), (4096, "2,2,1,1,8,1000", "0"), (4160, "1,2,1", "0"),
                               (4160, "2,1,1", "0"), (3200, "1,1,1", "0"), (8320, "2,2,2", "0"), (4096, "2,2,1", "2"), (4224, "2,2,1", "1"),
                               (8192, "2,2,2", "2"), (4096, "1,1,1", "2"), (2112, "1,1,1", "1"), (96, "2,2,1", "0"), (8, "2,2,1", "0")):
            a = rnd(90 + K, (K, M) if ta else (M, K), -1, 1)
            b = rnd(91 + K, (N, K) if tb else (K, N), -1, 1)
            opa, opb = np.ascontiguousarray(a.T if ta else a), np.ascontiguousarray(b.T if tb else b)
            dev.gemm_force(force); dev.gemm_kpair(int(pair))
            A, B, Cd = dev.array(a), dev.array(b), dev.full((M, N), np.nan)
            c.sgemm(dev, ta, tb, M, N, K, 1.0, A, a.shape[1], B, b.shape[1], 0.0, Cd, N)
            got = Cd.numpy()
            f = [int(v) for v in force.split(",")]
            parts = f[2] * (2 if pair != "0" else 1)       # chains per output: splits x k-pair halves, equal runs of whole k-tiles
            kt = -(-K // 32)
            per = -(-kt // f[2])
            bounds = []
            for sp in range(f[2]):
                k0, k1 = sp * per * 32, min(K, (sp + 1) * per * 32)
                if pair != "0":
                    h = (k1 - k0) // 2
                    bounds.append([(k0, k0 + h), (k0 + h, k1)])
                else:
                    bounds.append([(k0, k1)])
            want = None
            for sp in bounds:                               # second pass: s = 0; s += slab[k] in split order (f32)
                part = None
                for k0, k1 in sp:                           # k-pair: acc(first half) + acc(second half)
                    v = sgemm_device_order(opa[:, k0:k1], opb[k0:k1], KC)
                    part = v if part is None else part + v
                want = (np.zeros_like(part) + part) if want is None else want + part
            assert np.array_equal(got, want), (K, force, pair, parts, float(np.abs(got - want).max()))
            if K >= 2048 and f[2] == 1 and pair == "0":      # the check has teeth: chains of 1024 give other bits
                assert not np.array_equal(got, sgemm_device_order(opa, opb, 1024)), (K, force)
    finally:
        dev.gemm_force(None); dev.gemm_kpair(None)


def shared_chip_plan(M, N, K, busy, cus=256):
    """The host-side plan of sgemm_tail_kernel (nk_gemm.hip: gemm_tail_launch), restated: None when the launch stays plain,
    else (m_lo, n_lo, pieces, k-tiles per piece) of the rectangle that is cut along K."""
    tm, tn, kt = M // 128, N // 128, K // 32
    T, slots = tm * tn, 2 * cus - busy
    if slots < cus or T <= slots or T % slots == 0:
        return None
    g = 8 if tm % 8 == 0 else tm % 8
    tail = -(-(T % slots) // g) * g
    if tail > g * tn or tail >= T or slots // tail < 4:
        return None
    kts = max(4, -(-kt // (slots // tail)))
    pieces = -(-kt // kts)
    if pieces < 2:
        return None
    return (tm - g) * 128, (tn - tail // g) * 128, pieces, kts


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0)])
def test_sgemm_chained_launches(dev, ta, tb):
    """An unsplit plain-epilogue GEMM with K beyond one chain (GEMM_CHAIN_K = 2048 products; a k-pair block's halves are two chains)
    runs as consecutive launches over equal pieces of K, each on top of the last (nk_gemm.hip, gemm_impl).  BIT FOR BIT what the
    caller would get by issuing those launches himself with the knob at 0 (one chain per launch - whose own order is pinned by
    the device-order model above): alpha and beta on the first piece, beta = 1 after it; ragged K; a k-pair grid (2048 x 2048: one
    128 x 128 block per CU, so one launch reaches 4096); the knob's other lengths; a launch with a bias keeps ONE chain."""
    import ctypes
    c = capi()

    class View:
        def __init__(self, a, off): self.keep, self.p = a, ctypes.c_void_p(a.p.value + 4 * off)

    try:
        for (M, N, K, chain, reach) in ((2048, 4096, 4096, None, 2048), (2048, 4096, 4160, None, 2048), (4096, 2048, 6144, None, 2048),
                                        (2048, 2048, 8192, None, 4096), (2048, 4096, 3072, 1024, 1024), (2048, 4096, 2048, None, 2048)):
            a = rnd(70 + K, (K, M) if ta else (M, K), -1, 1)
            b = rnd(71 + K, (N, K) if tb else (K, N), -1, 1)
            c0 = rnd(72, (M, N), -1, 1)
            A, B = dev.array(a), dev.array(b)
            for alpha, beta in ((1.0, 0.0), (-0.5, 1.0)):
                dev.gemm_chain(chain)
                C1 = dev.array(c0)
                c.sgemm(dev, ta, tb, M, N, K, alpha, A, a.shape[1], B, b.shape[1], beta, C1, N)
                dev.gemm_chain(0)
                C2 = dev.array(c0)
                pieces = -(-K // reach)
                per = -(-(-(-K // pieces)) // 64) * 64
                k0 = 0
                while k0 < K:
                    kk = min(per, K - k0)
                    Av = View(A, k0 * a.shape[1] if ta else k0)
                    Bv = View(B, k0 if tb else k0 * b.shape[1])
                    c.sgemm(dev, ta, tb, M, N, kk, alpha, Av, a.shape[1], Bv, b.shape[1], beta if k0 == 0 else 1.0, C2, N)
                    k0 += kk
                assert np.array_equal(C1.numpy(), C2.numpy()), (M, N, K, chain, alpha, beta)
                if K > reach:       # ... and NOT what one chain gives (the cut is really taken)
                    C3 = dev.array(c0)
                    c.sgemm(dev, ta, tb, M, N, K, alpha, A, a.shape[1], B, b.shape[1], beta, C3, N)
                    assert not np.array_equal(C1.numpy(), C3.numpy())
        # an epilogue function acts on the whole sum: Linear forward (bias) is one chain whatever the knob says
        n, m, o = 2048, 4096, 4096
        x, w, bv = rnd(1, (n, m), -1, 1), rnd(2, (o, m), -1, 1), rnd(3, (o,), -1, 1)
        X, W, Bv = dev.array(x), dev.array(w), dev.array(bv)
        Y1, Y2 = dev.zeros((n, o)), dev.zeros((n, o))
        dev.gemm_chain(None); c.linear_fwd(dev, X, W, Bv, Y1)
        dev.gemm_chain(0); c.linear_fwd(dev, X, W, Bv, Y2)
        assert np.array_equal(Y1.numpy(), Y2.numpy())
    finally:
        dev.gemm_chain(None)


def test_sgemm_chained_launches_batched_and_copy_kernel(dev):
    """(i) The chain over K carries the two-level batch strides through its pieces: a batched unsplit GEMM with K = 4096 equals, bit for
    bit, the two batched calls over the halves of K issued by hand with the knob at 0 (beta = 1 on the second) - and not the one-chain
    launch.  (ii) `nk_copy` from 1 Mi floats on is a span-walk kernel (nk_common.h) instead of the runtime's blit: exact copies at sizes
    around the threshold, ragged lengths and misaligned pointers (those fall back to the blit)."""
    import ctypes
    c = capi()
    bo, bi, M, N, K = 2, 2, 1024, 2048, 4096                       # 4 x 128 tiles = 512 blocks: unsplit, K beyond one chain
    a, b = rnd(1, (bo * bi, M, K), -1, 1), rnd(2, (bo * bi, K, N), -1, 1)
    A, B, Cb = dev.array(a), dev.array(b), dev.full((bo * bi, M, N), np.nan)

    class View:
        def __init__(self, arr, off): self.keep, self.p = arr, ctypes.c_void_p(arr.p.value + 4 * off)

    def batched(Av, Bv, Cm, k, beta):
        c.sgemm_batched(dev, 0, 0, M, N, k, 1.0, Av, K, bi * M * K, M * K, Bv, N, bi * K * N, K * N, beta, Cm, N, bi * M * N, M * N, bo, bi)
    batched(A, B, Cb, K, 0.0)
    dev.gemm_chain(0)
    try:
        C1, C2 = dev.full((bo * bi, M, N), np.nan), dev.full((bo * bi, M, N), np.nan)
        batched(A, B, C1, K, 0.0)                                   # one chain of 4096
        batched(A, B, C2, K // 2, 0.0)                              # the pieces by hand
        batched(View(A, K // 2), View(B, (K // 2) * N), C2, K // 2, 1.0)
    finally:
        dev.gemm_chain(None)
    assert np.array_equal(Cb.numpy(), C2.numpy()) and not np.array_equal(Cb.numpy(), C1.numpy())
    for n, off_d, off_s in (((1 << 20), 0, 0), ((1 << 20) + 4, 0, 0), ((1 << 20) - 4, 0, 0), (5 * (1 << 20) + 8, 4, 8), ((1 << 21) + 3, 0, 0),
                            ((1 << 21), 1, 0), ((1 << 21), 0, 3), (3 << 20, 4, 4)):
        src = rnd(n, (n + 16,), -1, 1)
        S, D = dev.array(src), dev.full((n + 16,), 7.0)
        c.check(c.lib.nk_copy(dev.h, ctypes.c_void_p(D.p.value + 4 * off_d), ctypes.c_void_p(S.p.value + 4 * off_s), n))
        got = D.numpy()
        assert np.array_equal(got[off_d:off_d + n], src[off_s:off_s + n]), (n, off_d, off_s)
        assert np.all(got[:off_d] == 7.0) and np.all(got[off_d + n:] == 7.0), (n, off_d, off_s)


@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
@pytest.mark.parametrize("M,N,K,busy", [(4096, 4096, 256, 16), (2048, 4096, 1024, 8), (4096, 2048, 384, 32), (4096, 4096, 512, 40),
                                        (3200, 5120, 256, 16)])   # tile counts the rules keep at 128x128: 1024, 512, 512, 1024, 1000 (25 tile rows: a last group of one); >= 4 pieces possible
def test_gemm_shared_chip_schedule(dev, layout, M, N, K, busy):
    """nk_device_set_busy_slots(n): while an exchange holds n resident-block slots, a GEMM whose tiles no longer divide the free
    slots runs whole rounds of one tile per block and cuts the LEFT-OVER tiles - a rectangle at the end of the tile sequence -
    along K (sgemm_tail_kernel).  Checked, for the three layouts of a Linear layer's passes with their epilogues (NT: bias +
    ReLU forward; NN: the masked input gradient, beta = 1; TN: the weight gradient, beta = 1):
      * outside the rectangle the result is the plain launch's, bit for bit;
      * inside it equals the same product of the sub-matrices under split-K with the plan's piece length (pieces added in
        order, the launch's epilogue applied once) - bit for bit, through the forced split-K schedule of nk_sgemm;
      * the whole result is inside the contraction bound against f64;
      * run to run identical; busy = 0 afterwards gives the plain bits again."""
    from tolerance import assert_contraction
    c = capi()
    ta, tb = {"NT": (0, 1), "NN": (0, 0), "TN": (1, 0)}[layout]
    plan = shared_chip_plan(M, N, K, busy, dev.num_cus() if hasattr(dev, "num_cus") else 256)
    assert plan is not None, "the case is meant to take the shared-chip schedule"
    m_lo, n_lo, pieces, kts = plan
    a = rnd(1, (K, M) if ta else (M, K), -1, 1)
    b = rnd(2, (N, K) if tb else (K, N), -1, 1)
    c0 = rnd(3, (M, N), -1, 1)
    bias = rnd(4, (N,), -1, 1)
    mask = rnd(5, (M, N), -1, 1)
    A, B, Bi, Mk = dev.array(a), dev.array(b), dev.array(bias), dev.array(mask)

    def run(Ad, Bd, Cd, biasd, maskd, m, n):
        if layout == "NT":
            c.linear_relu_fwd(dev, Ad, Bd, biasd, Cd)                         # C = max(A . B^T + bias, 0)
        elif layout == "NN":
            c.linear_bwd_input_relu(dev, Cd, Ad, Bd, maskd)                   # C += (A . B) where mask > 0
        else:
            c.sgemm(dev, 1, 0, m, n, K, 1.0, Ad, m, Bd, n, 1.0, Cd, n)       # C += A^T . B
        return Cd.numpy()

    plain = run(A, B, dev.array(c0), Bi, Mk, M, N)
    try:
        dev.busy_slots(busy)
        shared = run(A, B, dev.array(c0), Bi, Mk, M, N)
        again = run(A, B, dev.array(c0), Bi, Mk, M, N)
    finally:
        dev.busy_slots(0)
    assert np.array_equal(shared, again)
    assert np.array_equal(run(A, B, dev.array(c0), Bi, Mk, M, N), plain)
    outside = np.ones((M, N), bool); outside[m_lo:, n_lo:] = False
    assert np.array_equal(shared[outside], plain[outside])
    assert not np.array_equal(shared[m_lo:, n_lo:], plain[m_lo:, n_lo:])      # the rectangle really is another chain order
    # the rectangle as a problem of its own, split-K forced to the plan's pieces (kts k-tiles each)
    opa, opb = (a.T if ta else a), (b.T if tb else b)
    sa, sb = opa[m_lo:], opb[:, n_lo:]
    sub_a = np.ascontiguousarray(sa.T if ta else sa)
    sub_b = np.ascontiguousarray(sb.T if tb else sb)
    assert -(-(K // 32) // pieces) == kts, "test case: the forced split must reproduce the plan's piece length"
    try:
        dev.gemm_force(f"2,2,{pieces}"); dev.gemm_kpair(0)
        sub = run(dev.array(sub_a), dev.array(sub_b), dev.array(c0[m_lo:, n_lo:]), dev.array(bias[n_lo:]), dev.array(mask[m_lo:, n_lo:]),
                  M - m_lo, N - n_lo)
    finally:
        dev.gemm_force(None); dev.gemm_kpair(None)
    assert np.array_equal(shared[m_lo:, n_lo:], sub)
    prod = opa.astype(np.float64) @ opb.astype(np.float64)
    if layout == "NT":
        want, cpu = np.maximum(prod + bias, 0), np.maximum(opa @ opb + bias, 0)
    elif layout == "NN":
        want, cpu = c0 + np.where(mask > 0, prod, 0), c0 + np.where(mask > 0, opa @ opb, 0).astype(np.float32)
    else:
        want, cpu = c0 + prod, c0 + opa @ opb
    assert_contraction("gemm_shared_chip_schedule", shared, want, K, cpu32=cpu, epilogue=True)


def test_gemm_shared_chip_schedule_leaves_other_launches_alone(dev):
    """Grids that divide the free slots, fit into one round, are split / batched / ragged / 64-wide, or whose left-over is more
    than a quarter of a round stay plain launches under nk_device_set_busy_slots: bit-identical results."""
    c = capi()
    try:
        for (M, N, K, ta, tb, busy) in ((1024, 1024, 512, 0, 1, 16), (2048, 2048, 256, 0, 0, 16), (4096, 3968, 128, 0, 1, 16),
                                        (1000, 3000, 300, 1, 0, 8), (3072, 3072, 256, 1, 1, 16), (4096, 4096, 64, 0, 1, 0),
                                        (2816, 2816, 128, 0, 1, 200), (4096, 4096, 256, 0, 1, 64)):
            a = rnd(1, (K, M) if ta else (M, K), -1, 1)
            b = rnd(2, (N, K) if tb else (K, N), -1, 1)
            A, B = dev.array(a), dev.array(b)
            outs = []
            for n in (0, busy):
                dev.busy_slots(n)
                Cd = dev.full((M, N), 0.5)
                c.sgemm(dev, ta, tb, M, N, K, 1.0, A, a.shape[1], B, b.shape[1], 1.0, Cd, N)
                outs.append(Cd.numpy())
            if ta and tb or shared_chip_plan(M, N, K, busy) is None or M % 128 or N % 128 or K % 32:
                assert np.array_equal(outs[0], outs[1]), (M, N, K, ta, tb, busy)
    finally:
        dev.busy_slots(0)
    with pytest.raises(RuntimeError):
        dev.busy_slots(-1)


def test_sgemm_large_rowsum_identity(dev):
    """Size-independent check at the BASELINE size (4096^2): (A.B).1 == A.(B.1)."""
    c = capi()
    n = 4096
    a, b = rnd(0, (n, n)), rnd(1, (n, n))
    A, B, Cd = dev.array(a), dev.array(b), dev.zeros((n, n))
    for ta, tb in ((0, 0), (0, 1), (1, 0)):
        c.sgemm(dev, ta, tb, n, n, n, 1.0, A, n, B, n, 0.0, Cd, n)
        opa = (a.T if ta else a).astype(np.float64)
        opb = (b.T if tb else b).astype(np.float64)
        got = Cd.numpy().astype(np.float64)
        want_rows = opa @ opb.sum(axis=1)
        np.testing.assert_allclose(got.sum(axis=1), want_rows, rtol=2e-6)
        want_cols = opa.sum(axis=0) @ opb
        np.testing.assert_allclose(got.sum(axis=0), want_cols, rtol=2e-6)
        # spot-check 64 entries exactly against f64 dot products
        rng = np.random.default_rng(5)
        from tolerance import abs_term
        for i, j in zip(rng.integers(0, n, 64), rng.integers(0, n, 64)):
            assert abs(got[i, j] - opa[i] @ opb[:, j]) <= abs_term(n, 1.0, 1.0)       # the unscaled term (sampled entries pass it)


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192])
def test_C2_matmul_fwd_bwd_every_benchmark_size(dev, n):
    """BASELINE config 2 at every size: the node entry points the C2 bench times (`nk_mm_fwd` NN, `nk_mm_bwd_left` NT `+=`,
    `nk_mm_bwd_right` TN `+=`) at N = 1024 / 2048 (small grids: 64x64 tiles, k-pair blocks), 4096 and 8192 (1 GiB of
    operands: tile count, XCD chunking, look-ahead path), each checked
      * through 96 sampled entries against f64 dot products of the operand rows / columns, under the suite's ONE contraction
        bound (tests/tolerance.py: the survey's, no factor) with err_cpu32 from OpenBLAS's f32 product of the same rows; at
        4096 / 8192 each product runs as 2 / 4 chained launches over K (chains of 2048, nk_gemm.hip GEMM_CHAIN_K) - margins
        recorded under the labels C2_<n>:<C|dA|dB>;
      * through the row-sum identity (A.B).1 == A.(B.1) over the whole result.
    Gradients start from a non-zero value so `+=` is exercised."""
    from tolerance import assert_contraction
    c = capi()
    a, b, g = rnd(0, (n, n)), rnd(1, (n, n)), rnd(2, (n, n))
    A, B, G = dev.array(a), dev.array(b), dev.array(g)
    Cm, dA, dB = dev.zeros((n, n)), dev.full((n, n), 0.5), dev.full((n, n), -0.25)
    c.mm_fwd(dev, A, B, Cm); c.mm_bwd_left(dev, dA, G, B); c.mm_bwd_right(dev, dB, A, G)
    rng = np.random.default_rng(5)
    ii, jj = rng.integers(0, n, 96), rng.integers(0, n, 96)
    ones = np.ones(n)
    for name, got_d, init, left, right, tl, tr in (("C", Cm, 0.0, a, b, False, False), ("dA", dA, 0.5, g, b, False, True),
                                                   ("dB", dB, -0.25, a, g, True, False)):
        got = got_d.numpy()
        lrows = (left[:, ii].T if tl else left[ii])                    # (96, n): row i of op(left)
        rcols = (right[jj] if tr else right[:, jj].T)                  # (96, n): column j of op(right)
        want = init + np.einsum("ek,ek->e", lrows.astype(np.float64), rcols.astype(np.float64))
        # the f32 CPU restatement: OpenBLAS sgemm of the sampled rows against the whole right operand, sampled columns taken
        full32 = np.ascontiguousarray(lrows) @ (right.T if tr else right)
        cpu32 = np.float32(init) + full32[np.arange(96), jj]
        assert_contraction(f"C2_{n}:{name}", got[ii, jj], want, n, float(np.abs(left).max()), float(np.abs(right).max()),
                           cpu32=cpu32)
        # (L.R).1 = L.(R.1): f64 on the host costs two matrix-vector products
        opr1 = (right.astype(np.float64).sum(axis=0) if tr else right.astype(np.float64) @ ones)
        want_rows = (left.astype(np.float64).T @ opr1 if tl else left.astype(np.float64) @ opr1) + init * n
        np.testing.assert_allclose(got.sum(axis=1, dtype=np.float64), want_rows, rtol=3e-6, err_msg=name)
        del got


@pytest.mark.parametrize("node", ["mm", "mm_t"])
@pytest.mark.parametrize("n,m,o", [(1024, 1024, 1024), (512, 1024, 256), (256, 256, 2048), (2048, 2048, 2048), (384, 640, 896),
                                   (100, 70, 50), (64, 64, 64)])
def test_mm_backward_as_one_call(dev, node, n, m, o):
    """nk_mm_bwd / nk_mm_t_bwd (MatrixMatrixMulBackward::backward, matrix_matrix_mul/mod.rs:121-126 and the MatMulT twin: both
    products of the node in one call, ONE launch of sgemm_pair_kernel when the pair is eligible) against the two single-product
    entry points: every output must be the same fma chain as in a launch of its own without k-pair blocks, so the forced one
    launch (mode 1) is BIT-identical to two launches under NK_TUNE_GEMM_KPAIR = 0; the rule (mode -1) equals either that or the
    two launches under the k-pair rule; ragged shapes fall back to two launches; `+=` on a non-zero gradient and the assign
    form; and the f64 oracle."""
    c = capi()
    a = rnd(1, (n, m))
    b = rnd(2, (m, o) if node == "mm" else (o, m))
    g = rnd(3, (n, o))
    A, B, G = dev.array(a), dev.array(b), dev.array(g)
    fn = c.mm_bwd if node == "mm" else c.mm_t_bwd
    left = c.mm_bwd_left if node == "mm" else c.mm_t_bwd_left

    def right(dB):
        if node == "mm": c.mm_bwd_right(dev, dB, A, G)
        else: c.mm_t_bwd_right(dev, dB, G, A)

    def run(pair_mode, kpair, assign):
        dev.gemm_pair(pair_mode); dev.gemm_kpair(kpair)
        try:
            dA, dB = dev.full(a.shape, 0.5), dev.full(b.shape, -0.25)
            if pair_mode is False:       # the two single-product entry points (`+=` only)
                left(dev, dA, G, B); right(dB)
            else:
                fn(dev, dA, dB, G, A, B, assign, assign)
            return dA.numpy(), dB.numpy()
        finally:
            dev.gemm_pair(None); dev.gemm_kpair(None)

    same = lambda x, y: all(np.array_equal(u, v) for u, v in zip(x, y))
    for assign in (False, True):
        two_rule, two_plain = run(0, None, assign), run(0, 0, assign)
        assert same(run(1, None, assign), two_plain)
        rule = run(-1, None, assign)
        assert same(rule, two_rule) or same(rule, two_plain)
    assert same(run(False, None, False), run(0, None, False))
    a64, b64, g64 = a.astype(np.float64), b.astype(np.float64), g.astype(np.float64)
    want_a = 0.5 + (g64 @ b64.T if node == "mm" else g64 @ b64)
    want_b = -0.25 + (a64.T @ g64 if node == "mm" else g64.T @ a64)
    got_a, got_b = run(-1, None, False)
    from tolerance import assert_contraction
    assert_contraction("mm_backward_as_one_call", got_a, want_a, o, np.abs(g).max(), np.abs(b).max(), epilogue=True)
    assert_contraction("mm_backward_as_one_call", got_b, want_b, n, np.abs(g).max(), np.abs(a).max(), epilogue=True)


@pytest.mark.parametrize("packed", [False, True])
def test_batched_pair_equals_two_batched_launches(dev, packed):
    """nk_sgemm_pair_batched on the geometry of the attention backward's dK / dV products (TN + TN over (sample, head), heads as
    column blocks): bit-identical to two nk_sgemm_batched launches, forced and by rule, also when the two outputs are column
    blocks of ONE buffer (the packed projection gradient: their address ranges interleave, their elements do not)."""
    c = capi()
    Bn, H, S, dh = 2, 4, 256, 64
    d = H * dh
    ld = 3 * d if packed else d
    ds, pd = rnd(1, (Bn * H, S, S), -1, 1), rnd(2, (Bn * H, S, S))
    q, do = rnd(3, (Bn * S, ld), -1, 1), rnd(4, (Bn * S, d), -1, 1)
    DS, PD, Q, DO = dev.array(ds), dev.array(pd), dev.array(q), dev.array(do)
    so, sq, po, pi = S * d, S * ld, H * S * S, S * S
    outs = []
    for mode in (0, 1, -1):
        dev.gemm_pair(mode)
        try:
            buf = dev.full((Bn * S, ld if packed else 2 * d), 0.25)   # packed: [ . | dK | dV ] column blocks; else two halves of the rows' columns
            if packed:
                dK, dV, ldo, sqo = buf.view_offset(d), buf.view_offset(2 * d), ld, sq
            else:
                dK, dV, ldo, sqo = buf.view_offset(0), buf.view_offset(d), 2 * d, S * 2 * d
            c.sgemm_pair_batched(dev, Bn, H, (1, 0, S, dh, S, DS, S, po, pi, Q, ld, sq, dh, 1.0, dK, ldo, sqo, dh),
                                 (1, 0, S, dh, S, PD, S, po, pi, DO, d, so, dh, 1.0, dV, ldo, sqo, dh))
            outs.append(buf.numpy())
        finally:
            dev.gemm_pair(None)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    got = outs[0].reshape(Bn, S, -1)
    k0 = d if packed else 0
    for bb in range(Bn):
        for h in range(H):
            wk = 0.25 + ds[bb * H + h].astype(np.float64).T @ q[bb * S:(bb + 1) * S, h * dh:(h + 1) * dh].astype(np.float64)
            wv = 0.25 + pd[bb * H + h].astype(np.float64).T @ do[bb * S:(bb + 1) * S, h * dh:(h + 1) * dh].astype(np.float64)
            close(got[bb, :, k0 + h * dh:k0 + (h + 1) * dh], wk, rtol=1e-5, atol=1e-6 * S)
            close(got[bb, :, k0 + d + h * dh:k0 + d + (h + 1) * dh], wv, rtol=1e-5, atol=1e-6 * S)


def test_mm_backward_one_call_keeps_order_on_aliased_gradients(dev):
    """x.mm(x): both products accumulate into ONE gradient buffer.  Two launches on a stream are ordered; one launch would run
    them concurrently - nk_sgemm_pair must notice the overlap and launch twice (forced mode 1 included)."""
    c = capi()
    n = 256
    a, g = rnd(1, (n, n)), rnd(2, (n, n))
    A, G = dev.array(a), dev.array(g)
    outs = []
    for mode in (0, 1, -1):
        dev.gemm_pair(mode)
        try:
            D = dev.full((n, n), 0.125)
            c.mm_bwd(dev, D, D, G, A, A)
            outs.append(D.numpy())
        finally:
            dev.gemm_pair(None)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    a64, g64 = a.astype(np.float64), g.astype(np.float64)
    from tolerance import assert_contraction
    assert_contraction("mm_backward_aliased", outs[0], 0.125 + g64 @ a64.T + a64.T @ g64, 2 * n, epilogue=True)   # two products of length n


# ------------------------------------------------------------------------------ binaries
BCAST = [((64, 96), (96,)), ((96,), (64, 96)), ((2, 2, 3), (1, 3)), ((1, 3), (2, 2, 3)), ((4, 8, 5, 6), (8, 1, 1)),
         ((7, 5), ()), ((33, 1), (1, 17)), ((2, 3, 1, 5, 2), (3, 4, 1, 2)), ((512, 1024), (512, 1024)), ((5, 7), (5, 7))]


@pytest.mark.parametrize("ls,rs", BCAST)
@pytest.mark.parametrize("op", ["add", "sub", "mul", "div"])
def test_binary_fwd_bwd(dev, op, ls, rs):
    c = capi()
    l, r = rnd(1, ls, 0.5, 1.5), rnd(2, rs, 0.5, 1.5)
    oshape = O.cobroadcast(ls, rs)
    L, R, OUT = dev.array(l), dev.array(r), dev.full(oshape, 9.0)
    c.binary_fwd(dev, op, OUT, L, R)
    want = np.zeros(oshape, np.float32); O.binary_forward(op, l, r, want)
    close(OUT.numpy(), want)
    g = rnd(3, oshape)
    G = dev.array(g)
    dl0, dr0 = rnd(4, ls), rnd(5, rs)
    DL, DR = dev.array(dl0), dev.array(dr0)
    c.binary_bwd_left(dev, op, DL, G, R)
    c.binary_bwd_right(dev, op, DR, G, L, R)
    dl, dr = np.array(dl0, dtype=np.float64), np.array(dr0, dtype=np.float64)   # 0-d stays an ndarray
    O.binary_backward_left(op, dl, g.astype(np.float64), l.astype(np.float64), r.astype(np.float64))
    O.binary_backward_right(op, dr, g.astype(np.float64), l.astype(np.float64), r.astype(np.float64))
    red_l = max(1, int(np.prod(oshape)) // max(1, l.size))
    red_r = max(1, int(np.prod(oshape)) // max(1, r.size))
    close(DL.numpy(), dl, rtol=1e-5, atol=2e-6 * red_l)
    close(DR.numpy(), dr, rtol=1e-5, atol=2e-5 * red_r)


def test_binary_reduction_golden(dev, golden):
    """addition/test.rs:110-124, multiplication/test.rs:121-139, division/test.rs:121-207."""
    c = capi()
    n = golden["nodes"]

    def pair(cs):
        v = [k["value"] for k in cs["constructors"] if k["shape"] == [3]]
        return v[-2], v[-1]

    G = dev.full((3, 3), 1.0)
    D = dev.zeros((3,))
    once, twice = pair(n["addition_backward_left_reduction"])
    c.binary_bwd_left(dev, "add", D, G); close(D.numpy(), np.full(3, once), 0, F16_EPSILON)
    c.binary_bwd_left(dev, "add", D, G); close(D.numpy(), np.full(3, twice), 0, F16_EPSILON)
    cs = n["multiplication_backward_left_reduction"]
    once, twice = pair(cs)
    R = dev.full((3, 3), cs["constructors"][0]["value"]); D = dev.zeros((3,))
    c.binary_bwd_left(dev, "mul", D, G, R); close(D.numpy(), np.full(3, once), 0, F16_EPSILON)
    c.binary_bwd_left(dev, "mul", D, G, R); close(D.numpy(), np.full(3, twice), 0, F16_EPSILON)
    cs = n["division_backward_right_reduction"]
    once, twice = pair(cs)
    L = dev.full((3, 3), cs["constructors"][1]["value"]); R = dev.full((3,), cs["constructors"][2]["value"])
    D = dev.zeros((3,))
    c.binary_bwd_right(dev, "div", D, G, L, R); close(D.numpy(), np.full(3, once), 0, F16_EPSILON)
    c.binary_bwd_right(dev, "div", D, G, L, R); close(D.numpy(), np.full(3, twice), 0, F16_EPSILON)


def test_binary_incompatible_shapes(dev):
    c = capi()
    with pytest.raises(c.NeuronikaHipError, match="incompatible shape"):
        c.binary_fwd(dev, "add", dev.zeros((2, 3)), dev.zeros((2, 3)), dev.zeros((2, 4)))


def test_bias_column_reduction_full_size(dev):
    """C4 bias gradient: (4096,4096) -> (4096); checked against f64 column sums."""
    c = capi()
    g = rnd(7, (4096, 4096), -1, 1)
    G, D = dev.array(g), dev.zeros((4096,))
    c.unbroadcast_add(dev, D, G)
    close(D.numpy(), g.astype(np.float64).sum(0), rtol=1e-5, atol=1e-3)


# ------------------------------------------------------------------------------ relu / sum / mean / mse
def test_relu(dev, golden):
    c = capi()
    lit = golden["nodes"]["relu_forward"]["literals"]
    X, Y = dev.array(f32(lit[0], (3, 3))), dev.full((3, 3), 5.0)
    c.relu_fwd(dev, X, Y); assert np.array_equal(Y.numpy(), f32(lit[1], (3, 3)))
    lit = golden["nodes"]["relu_backward"]["literals"]
    DX, X, G = dev.array(f32(lit[0])), dev.array(f32(lit[1])), dev.array(f32(lit[2]))
    c.relu_bwd(dev, DX, G, X); assert np.array_equal(DX.numpy(), f32(lit[4]))
    c.relu_bwd(dev, DX, G, X); assert np.array_equal(DX.numpy(), f32(lit[5]))
    x = rnd(1, (1000, 37), -1, 1); x[::7] = 0.0          # strict `>`: gradient at 0 is 0
    g, d0 = rnd(2, x.shape, -1, 1), rnd(3, x.shape)
    X, G, D, Y = dev.array(x), dev.array(g), dev.array(d0), dev.zeros(x.shape)
    c.relu_fwd(dev, X, Y); c.relu_bwd(dev, D, G, X)
    y = np.zeros_like(x); O.relu_forward(x, y); d = d0.copy(); O.relu_backward(d, g, x)
    assert np.array_equal(Y.numpy(), y) and np.array_equal(D.numpy(), d)     # bit-exact mask


def same_nonfinite(got, ref, rtol=1e-5, atol=1e-6):
    """NaN positions and the positions / signs of infinities bit for bit, finite values to the elementwise tolerance."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), (np.isnan(got).sum(), np.isnan(ref).sum())
    inf = np.isinf(ref)
    assert np.array_equal(np.isinf(got), inf) and np.array_equal(got[inf], ref[inf])
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=rtol, atol=atol)


def test_non_finite_inputs_follow_the_reference(dev):
    """NaN / +-inf through the nodes whose reference code takes a position on them:
    relu forward `o.max(0.)` (Rust `f32::max` ignores a NaN operand: NaN -> 0, relu/mod.rs:33-37), relu backward
    `((x > 0.) as usize as f32) * g` (0 * inf = NaN, :71-78), softmax / log-softmax with the max folded from `f32::MIN` by
    `x.max(y)` (a NaN or a +inf anywhere in a lane, or a lane of -inf only, makes the WHOLE lane NaN; a single -inf among
    finite values is an exact 0 / -inf, softmax/mod.rs:41-52, logsoftmax/mod.rs:41-52), division (x/0 = +-inf, 0/0 = NaN,
    and both backward formulas, division/mod.rs:39-50,90-99,139-149) and the squared error (NaN / inf - inf propagate
    into the scalar and into the gradient, squared_error/mod.rs:42-59,94-123)."""
    c = capi()
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    with np.errstate(all="ignore"):
        # ---- relu
        x = rnd(1, (64, 33), -1, 1)
        x[0, :4] = [nan, inf, -inf, -0.0]; x[5, 7] = nan; x[63, 32] = -inf
        g = rnd(2, x.shape, -1, 1)
        g[0, :4] = [1.0, inf, inf, nan]; g[9, 9] = inf; g[10, 10] = -inf; g[11, 11] = nan     # inf / NaN gradients on both sides of the mask
        x[9, 9], x[10, 10], x[11, 11] = -0.5, 0.5, -0.25
        X, G, Y = dev.array(x), dev.array(g), dev.full(x.shape, 7.0)
        c.relu_fwd(dev, X, Y)
        y = np.empty_like(x); O.relu_forward(x, y)
        assert not np.isnan(y).any() and y[0, 0] == 0 and y[0, 1] == inf and y[0, 2] == 0       # the reference's semantics, spelled out
        assert np.array_equal(Y.numpy(), y)
        for assign in (False, True):
            d0 = rnd(3, x.shape)
            D = dev.array(d0)
            c.relu_bwd(dev, D, G, X, assign=assign)
            d = np.zeros_like(d0) if assign else d0.copy()
            O.relu_backward(d, g, x)
            assert np.isnan(d[9, 9]) and d[10, 10] == -inf and np.isnan(d[0, 2])               # 0 * inf = NaN where the mask is 0
            assert np.array_equal(D.numpy(), d, equal_nan=True), assign
        # ---- softmax / log-softmax, both axes, lanes longer and shorter than a wave
        for shape, axis in (((12, 1024), 1), ((12, 40), 1), ((40, 12), 0), ((12, 3000), 1)):
            x = rnd(4, shape, -4, 4)
            lane = lambda i: (i, slice(None)) if axis == 1 else (slice(None), i)
            at = lambda i, j: (i, j) if axis == 1 else (j, i)
            x[lane(1)] = -inf                       # all -inf: max stays f32::MIN, exp = 0, 0 / 0
            x[at(2, 3)] = inf                       # one +inf: exp(inf - inf) = NaN poisons the sum
            x[at(3, 5)] = nan                       # one NaN: skipped by the max fold, NaN in the sum
            x[at(4, 7)] = -inf                      # one -inf among finite values: an exact zero / -inf
            x[at(5, 0)] = -inf; x[at(5, shape[axis] - 1)] = -inf
            x[at(6, 1)] = inf; x[at(6, 2)] = inf
            x[at(7, 2)] = nan; x[at(7, 4)] = inf
            X = dev.array(x)
            for fwd, ofwd, bwd, obwd in ((c.softmax_fwd, O.softmax_forward, c.softmax_bwd, O.softmax_backward),
                                         (c.log_softmax_fwd, O.log_softmax_forward, c.log_softmax_bwd, O.log_softmax_backward)):
                Y = dev.full(shape, 3.0)
                fwd(dev, X, Y, axis)
                y = np.empty(shape, np.float32); ofwd(x, y, axis)
                for i in (1, 2, 3, 6, 7):
                    assert np.isnan(y[lane(i)]).all()
                assert not np.isnan(y[lane(4)]).any() and not np.isnan(y[lane(0)]).any()
                same_nonfinite(Y.numpy(), y)
                g, d0 = rnd(5, shape, -1, 1), rnd(6, shape)
                G, D = dev.array(g), dev.array(d0)
                bwd(dev, D, G, Y, axis)
                d = d0.copy(); obwd(d, g, Y.numpy(), axis)      # from the device's own y: the backward formula on non-finite y
                same_nonfinite(D.numpy(), d, rtol=1e-5, atol=2e-6)
        # ---- division
        l, r = rnd(7, (33, 17), -1, 1), rnd(8, (33, 17), 0.5, 1.5)
        l[0, :6] = [1.0, -1.0, 0.0, inf, nan, inf]; r[0, :6] = [0.0, 0.0, 0.0, inf, 1.0, 0.0]
        l[1, :3] = [2.0, -0.0, 1.0]; r[1, :3] = [-0.0, 3.0, nan]
        L, R, OUT = dev.array(l), dev.array(r), dev.zeros(l.shape)
        c.binary_fwd(dev, "div", OUT, L, R)
        o = np.empty_like(l); O.binary_forward("div", l, r, o)
        assert o[0, 0] == inf and o[0, 1] == -inf and np.isnan(o[0, 2]) and np.isnan(o[0, 3]) and o[1, 0] == -inf
        assert np.array_equal(OUT.numpy(), o, equal_nan=True)
        g = rnd(9, l.shape, -1, 1); g[2, 2] = inf; g[2, 3] = nan
        G, DL, DR = dev.array(g), dev.zeros(l.shape), dev.zeros(l.shape)
        c.binary_bwd_left(dev, "div", DL, G, R); c.binary_bwd_right(dev, "div", DR, G, L, R)
        dl, dr = np.zeros_like(l), np.zeros_like(l)
        O.binary_backward_left("div", dl, g, l, r); O.binary_backward_right("div", dr, g, l, r)
        same_nonfinite(DL.numpy(), dl); same_nonfinite(DR.numpy(), dr, rtol=2e-6, atol=1e-7)
        # ---- squared error
        for bad in ((nan, 0.5), (inf, 0.5), (inf, inf), (-inf, inf)):
            x, t = rnd(10, (40, 50)), rnd(11, (40, 50))
            x[3, 4], t[3, 4] = bad
            X, T, out = dev.array(x), dev.array(t), dev.zeros(())
            for red in ("mean", "sum"):
                c.mse_fwd(dev, X, T, out, red)
                ref = np.zeros((), np.float32); O.squared_error_forward(x, t, ref, red)
                same_nonfinite(np.float32(out.item()), ref)
                D, Gs = dev.zeros(x.shape), dev.full((), 0.5)
                c.mse_bwd(dev, D, Gs, X, T, red)
                d = np.zeros_like(x); O.squared_error_backward(d, np.float32(0.5), x, t, red)
                same_nonfinite(D.numpy(), d, rtol=1e-6, atol=1e-12)


def test_sum_mean_mse(dev, golden):
    c = capi()
    n = golden["nodes"]
    out = dev.zeros(())
    X = dev.array(f32(n["sum_forward"]["literals"][0], (3, 3)))
    c.sum_fwd(dev, X, out); close(out.item(), n["sum_forward"]["scalars"][0], 0, F16_EPSILON)
    c.mean_fwd(dev, X, out); close(out.item(), n["mean_forward"]["scalars"][0], 0, F16_EPSILON)
    for op, fn in (("sum", c.sum_bwd), ("mean", c.mean_bwd)):
        lit = n[f"{op}_backward"]["literals"]
        D, G = dev.array(f32(lit[0], (10, 10))), dev.full((), n[f"{op}_backward"]["scalars"][0])
        fn(dev, D, G); close(D.numpy(), f32(lit[1], (10, 10)), 0, F16_EPSILON)
        fn(dev, D, G); close(D.numpy(), f32(lit[2], (10, 10)), 0, F16_EPSILON)
    for red in ("mean", "sum"):
        cs = n[f"squared_error_{red}"]; lit = cs["literals"]
        T, X = dev.array(f32(lit[0], (3, 3))), dev.array(f32(lit[1], (3, 3)))
        c.mse_fwd(dev, X, T, out, red); close(out.item(), cs["scalars"][0], 0, F16_EPSILON)
        D, G = dev.array(f32(lit[2], (3, 3))), dev.full((), cs["scalars"][1])
        c.mse_bwd(dev, D, G, X, T, red); close(D.numpy(), f32(lit[3], (3, 3)), 0, F16_EPSILON)
        c.mse_bwd(dev, D, G, X, T, red); close(D.numpy(), f32(lit[4], (3, 3)), 0, F16_EPSILON)
    # large random: tolerance against f64
    x, t = rnd(1, (4096, 1025)), rnd(2, (4096, 1025))
    X, T = dev.array(x), dev.array(t)
    c.sum_fwd(dev, X, out); close(out.item(), x.astype(np.float64).sum(), rtol=1e-6)
    c.mean_fwd(dev, X, out); close(out.item(), x.astype(np.float64).mean(), rtol=1e-6)
    c.mse_fwd(dev, X, T, out, "mean"); close(out.item(), ((x.astype(np.float64) - t) ** 2).mean(), rtol=1e-6)
    D, G = dev.zeros(x.shape), dev.full((), 0.5)
    c.mse_bwd(dev, D, G, X, T, "mean")
    d = np.zeros_like(x); O.squared_error_backward(d, np.float32(0.5), x, t, "mean")
    close(D.numpy(), d, rtol=1e-6, atol=1e-12)


# ------------------------------------------------------------------------------ softmax
@pytest.mark.parametrize("op", ["softmax", "logsoftmax"])
def test_softmax_golden(dev, golden, op):
    c = capi()
    fwd = c.softmax_fwd if op == "softmax" else c.log_softmax_fwd
    bwd = c.softmax_bwd if op == "softmax" else c.log_softmax_bwd
    for which in ("rows", "columns"):
        cs = golden["nodes"][f"{op}_forward_{which}"]
        X, Y = dev.array(f32(cs["input"], cs["shape"])), dev.full(cs["shape"], 3.0)
        fwd(dev, X, Y, cs["axis"]); close(Y.numpy(), f32(cs["out"], cs["shape"]), 0, F16_EPSILON)
        b = golden["nodes"][f"{op}_backward_{which}"]; lit = b["literals"]
        X, G, Y, D = dev.array(f32(lit[1], (3, 3))), dev.array(f32(lit[2], (3, 3))), dev.zeros((3, 3)), dev.array(f32(lit[0], (3, 3)))
        fwd(dev, X, Y, b["axis"])
        bwd(dev, D, G, Y, b["axis"]); close(D.numpy(), f32(lit[4], (3, 3)), 0, F16_EPSILON)
        bwd(dev, D, G, Y, b["axis"]); close(D.numpy(), f32(lit[5], (3, 3)), 0, F16_EPSILON)


@pytest.mark.parametrize("shape,axis", [((64, 1024), 1), ((33, 1000), 1), ((7, 3000), 1), ((5, 10), 1), ((300, 40), 0),
                                        ((6, 50, 12), 1), ((4, 8, 256), 2), ((3, 2052), 1)])
@pytest.mark.parametrize("op", ["softmax", "logsoftmax"])
def test_softmax_random(dev, op, shape, axis):
    c = capi()
    fwd = c.softmax_fwd if op == "softmax" else c.log_softmax_fwd
    bwd = c.softmax_bwd if op == "softmax" else c.log_softmax_bwd
    ofwd = O.softmax_forward if op == "softmax" else O.log_softmax_forward
    obwd = O.softmax_backward if op == "softmax" else O.log_softmax_backward
    x, g, d0 = rnd(1, shape, -4, 4), rnd(2, shape, -1, 1), rnd(3, shape)
    X, G, D, Y = dev.array(x), dev.array(g), dev.array(d0), dev.zeros(shape)
    fwd(dev, X, Y, axis)
    y = np.zeros(shape, np.float64); ofwd(x.astype(np.float64), y, axis)
    close(Y.numpy(), y, rtol=1e-5, atol=1e-6)
    if op == "softmax":   # index-like behaviour: arg-max of each lane is preserved exactly
        assert np.array_equal(Y.numpy().argmax(axis), x.argmax(axis))
    bwd(dev, D, G, Y, axis)
    d = d0.astype(np.float64); obwd(d, g.astype(np.float64), Y.numpy().astype(np.float64), axis)
    close(D.numpy(), d, rtol=1e-5, atol=2e-6)


# ------------------------------------------------------------------------------ dropout
def test_dropout(dev):
    c = capi()
    n = 100_003
    x = rnd(1, (n,), 0.1, 1.0)
    X, Y, NZ = dev.array(x), dev.full((n,), 5.0), dev.zeros((n,))
    c.dropout_fwd(dev, X, Y, NZ, 1.0, True)                  # dropout/test.rs:57-68
    assert np.array_equal(Y.numpy(), np.zeros(n, np.float32)) and np.array_equal(NZ.numpy(), np.zeros(n, np.float32))
    c.dropout_fwd(dev, X, Y, NZ, 0.0, True)                  # :71-85
    assert np.array_equal(Y.numpy(), x)
    c.dropout_fwd(dev, X, Y, NZ, 0.5, False)                 # eval mode: copy
    assert np.array_equal(Y.numpy(), x)
    # (p = 0.09 / 0.16 / 0.33: f32(1) - f32(p) != f32(1 - p) - the scale follows dropout/mod.rs:76 literally)
    for p, seed, off in ((0.5, 7, 0), (0.1, 123456789012345, 1 << 33), (0.09, 3, (1 << 32) - 2), (0.16, 4, 1), (0.33, 5, 2), (0.999, 6, 3)):
        c.dropout_fwd(dev, X, Y, NZ, p, True, seed, off)
        noise = O.dropout_noise(n, p, seed, off)
        assert np.array_equal(NZ.numpy(), noise)             # bit-exact keep/drop pattern
        y = np.zeros_like(x); O.dropout_forward(x, y, noise, p, True)
        assert np.array_equal(Y.numpy(), y)
        assert np.all(Y.numpy() <= x / (np.float32(1) - np.float32(p)) * (1 + 1e-6))   # :88-104 (<= 2x at p=.5)
        g, d0 = rnd(2, (n,)), rnd(3, (n,))
        G, D = dev.array(g), dev.array(d0)
        c.dropout_bwd(dev, D, G, NZ, p, True)
        d = d0.copy(); O.dropout_backward(d, g, noise, p, True)  # NOT divided by 1-p
        assert np.array_equal(D.numpy(), d)
    D = dev.zeros((n,)); G = dev.full((n,), 1.0)
    c.dropout_bwd(dev, D, G, NZ, 0.0, True); assert np.array_equal(D.numpy(), np.ones(n, np.float32))  # :133-158
    with pytest.raises(c.NeuronikaHipError, match="Wrong probability"):
        c.dropout_fwd(dev, X, Y, NZ, 1.5, True)


def test_fused_attention_probs_equals_three_nodes(dev):
    """nk_scale_softmax_dropout_* == Multiplication(scalar) -> Softmax(last) -> Dropout: the mask and the set of
    dropped elements bit for bit; the values to f32 rounding (the fused kernel multiplies by one reciprocal per row
    and by the host's 1/(1-p) where the reference nodes divide per element: at most one rounding apart per
    operation, far inside the stated rtol 1e-5), the backward likewise."""
    c = capi()
    rows, L = 96, 1024
    scale, p, seed, off = 0.125, 0.1, 77, 5
    s = rnd(1, (rows, L), -8, 8)
    S, SC = dev.array(s), dev.array(np.full((), scale, np.float32))
    SCALED, P1, O1, NZ = dev.zeros((rows, L)), dev.zeros((rows, L)), dev.zeros((rows, L)), dev.zeros((rows, L))
    c.binary_fwd(dev, "mul", SCALED, S, SC); c.softmax_fwd(dev, SCALED, P1, 1); c.dropout_fwd(dev, P1, O1, NZ, p, True, seed, off)
    P2, O2 = dev.zeros((rows, L)), dev.zeros((rows, L))
    c.scale_softmax_dropout_fwd(dev, S, P2, O2, None, scale, p, True, seed, off)
    np.testing.assert_allclose(P2.numpy(), P1.numpy(), rtol=3e-7, atol=0)      # <= 2 ulp: reciprocal-multiply vs divide
    np.testing.assert_allclose(O2.numpy(), O1.numpy(), rtol=5e-7, atol=0)
    assert np.array_equal(O2.numpy() == 0, O1.numpy() == 0)                    # exactly the same elements dropped
    NZ2, O3 = dev.zeros((rows, L)), dev.zeros((rows, L))
    c.scale_softmax_dropout_fwd(dev, S, P2, O3, NZ2, scale, p, True, seed, off)       # stored-mask form
    assert np.array_equal(NZ.numpy(), NZ2.numpy()) and np.array_equal(O3.numpy(), O2.numpy())
    assert np.array_equal(NZ.numpy().reshape(-1), O.dropout_noise(rows * L, p, seed, off))
    g, d0 = rnd(2, (rows, L), -1, 1), rnd(3, (rows, L))
    G = dev.array(g)
    dP, dSC, dS1 = dev.zeros((rows, L)), dev.zeros((rows, L)), dev.array(d0)
    c.dropout_bwd(dev, dP, G, NZ, p, True); c.softmax_bwd(dev, dSC, dP, P1, 1); c.binary_bwd_left(dev, "mul", dS1, dSC, SC)
    dS2, dS3 = dev.array(d0), dev.array(d0)
    c.scale_softmax_dropout_bwd(dev, dS2, G, P2, None, scale, p, True, seed, off)       # regenerated mask
    c.scale_softmax_dropout_bwd(dev, dS3, G, P2, NZ2, scale, p, True, seed, off)        # stored mask
    assert np.array_equal(dS2.numpy(), dS3.numpy())
    close(dS2.numpy(), dS1.numpy(), rtol=1e-6, atol=1e-7)
    # eval mode / p = 0 / p = 1 (dropout/test.rs:57-85 semantics carried through)
    P2ref = P2.numpy().copy()
    for pp, train in ((0.3, False), (0.0, True)):
        c.scale_softmax_dropout_fwd(dev, S, P2, O2, None, scale, pp, train, seed, off)
        assert np.array_equal(O2.numpy(), P2ref)                                # eval / p = 0: the probabilities pass through
    c.scale_softmax_dropout_fwd(dev, S, P2, O2, None, scale, 1.0, True, seed, off)
    assert not O2.numpy().any() and np.array_equal(P2.numpy(), P2ref)
    dS4 = dev.array(d0); c.scale_softmax_dropout_bwd(dev, dS4, G, P2, None, scale, 1.0, True, seed, off)
    assert np.array_equal(dS4.numpy(), d0)
    with pytest.raises(c.NeuronikaHipError, match="L % 4 == 0"):
        c.scale_softmax_dropout_fwd(dev, dev.zeros((4, 6)), dev.zeros((4, 6)), dev.zeros((4, 6)), None, 1.0, 0.0)


# ------------------------------------------------------------------------------ layout glue
@pytest.mark.parametrize("mode", ["reflective", "replicative"])
def test_pad_modes(dev, golden, mode):
    """Pad<Reflective|Replicative>: the reference's exact 1/2/3-d vectors, random batched cases vs the oracle,
    the mode-independent backward (pad/mod.rs:157-181) and the out-of-range reflective case."""
    c = capi()
    for fn in ("test_1d", "test_2d", "test_3d"):
        cs = golden["nodes"][f"pad_{mode}_{fn}"]
        base = np.arange(cs["arange"], dtype=np.float32).reshape([1, 1] + cs["base_shape"])
        X, Y = dev.array(base), dev.full([1, 1] + cs["padded_shape"], -7.0)
        c.pad_mode_fwd(dev, X, Y, cs["padding"], mode)
        assert np.array_equal(Y.numpy()[0, 0], f32(cs["expected"], cs["padded_shape"])), cs["cite"]
    for seed, shape, pad in [(1, (3, 5, 17), (4,)), (2, (2, 3, 9, 12), (3, 0)), (3, (2, 2, 4, 5, 6), (1, 2, 3)),
                             (4, (4, 64, 28, 28), (1, 1))]:
        x = rnd(seed, shape)
        oshape = list(shape[:2]) + [n + 2 * p for n, p in zip(shape[2:], pad)]
        y = np.zeros(oshape, np.float32); O.pad_mode_forward(x, y, pad, mode)
        X, Y = dev.array(x), dev.zeros(oshape)
        c.pad_mode_fwd(dev, X, Y, pad, mode)
        assert np.array_equal(Y.numpy(), y)
        g = rnd(seed + 10, oshape); d0 = rnd(seed + 20, shape)
        D, G = dev.array(d0), dev.array(g)
        c.pad_bwd(dev, D, G, pad)
        d = d0.copy(); O.pad_backward(d, g, pad)
        assert np.array_equal(D.numpy(), d)
    if mode == "reflective":
        X, Y = dev.array(rnd(5, (1, 1, 3))), dev.zeros((1, 1, 9))
        with pytest.raises(RuntimeError, match="reflective padding"):
            c.pad_mode_fwd(dev, X, Y, (3,), mode)


def test_pad_chunk_concat_transpose(dev, golden):
    c = capi()
    for mode in ("zero", "constant"):
        cs = golden["nodes"][f"pad_{mode}_test"]
        want = f32(cs["literals"][0], (7, 9))
        base = np.arange(25, dtype=np.float32).reshape(1, 1, 5, 5)
        X, Y = dev.array(base), dev.full((1, 1, 7, 9), -1.0)
        c.pad_const_fwd(dev, X, Y, (1, 2), float(want[0, 0]))
        assert np.array_equal(Y.numpy()[0, 0], want), cs["cite"]
        D = dev.array(base)
        c.pad_bwd(dev, D, Y, (1, 2))
        assert np.array_equal(D.numpy(), 2 * base)
    x = rnd(1, (3, 4, 9, 12)); y = np.zeros((3, 4, 11, 18), np.float32)
    X, Y = dev.array(x), dev.zeros(y.shape)
    c.pad_const_fwd(dev, X, Y, (1, 3), 0.25); O.pad_constant_forward(x, y, (1, 3), 0.25)
    assert np.array_equal(Y.numpy(), y)
    x3 = rnd(2, (2, 2, 3, 4, 5)); y3 = np.zeros((2, 2, 5, 4, 9), np.float32)
    X3, Y3 = dev.array(x3), dev.zeros(y3.shape)
    c.pad_const_fwd(dev, X3, Y3, (1, 0, 2), 0.0); O.pad_constant_forward(x3, y3, (1, 0, 2), 0.0)
    assert np.array_equal(Y3.numpy(), y3)

    lit = golden["nodes"]["chunk_forward_base_case"]["literals"]
    xin = np.linspace(-4.0, 4.0, 9, dtype=np.float32).reshape(3, 3)
    X = dev.array(xin)
    for i in range(3):
        Y = dev.zeros((1, 3)); c.chunk_fwd(dev, X, Y, i)
        assert np.array_equal(Y.numpy(), f32(lit[i], (1, 3)))
    lit = golden["nodes"]["chunk_backward_base_case"]["literals"]
    G = dev.full((1, 3), 1.0)
    for i in range(3):
        D = dev.zeros((3, 3))
        c.chunk_bwd(dev, D, G, i); assert np.array_equal(D.numpy(), f32(lit[2 * i], (3, 3)))
        c.chunk_bwd(dev, D, G, i); assert np.array_equal(D.numpy(), f32(lit[2 * i + 1], (3, 3)))
    x = rnd(3, (10, 64, 20))
    X = dev.array(x)
    for no in (0, 3, 7):
        Y = dev.zeros((5, 16, 8)); c.chunk_fwd(dev, X, Y, no)
        y = np.zeros((5, 16, 8), np.float32); O.chunk_forward(x, y, no)
        assert np.array_equal(Y.numpy(), y)
        D = dev.array(x); c.chunk_bwd(dev, D, Y, no)
        d = x.copy(); O.chunk_backward(d, y, no)
        assert np.array_equal(D.numpy(), d)

    parts = [rnd(4, (3, 5, 8)), rnd(5, (3, 2, 8)), rnd(6, (3, 9, 8))]
    out = np.zeros((3, 16, 8), np.float32); O.multi_concatenate_forward(parts, out, 1)
    P, OUT = [dev.array(p) for p in parts], dev.zeros(out.shape)
    c.concat_fwd(dev, P, OUT, 1); assert np.array_equal(OUT.numpy(), out)
    d0 = [rnd(7 + i, p.shape) for i, p in enumerate(parts)]
    DP = [dev.array(d) for d in d0]
    c.concat_bwd(dev, DP, OUT, 1)
    dd = [d.copy() for d in d0]; O.multi_concatenate_backward(dd, out, 1)
    for a, b in zip(DP, dd):
        assert np.array_equal(a.numpy(), b)

    lit = golden["nodes"]["transpose_forward"]["literals"]
    X, Y = dev.array(f32(lit[0], (3, 3))), dev.zeros((3, 3))
    c.transpose_fwd(dev, X, Y); assert np.array_equal(Y.numpy(), f32(lit[1], (3, 3)))
    for shape in ((70, 45), (3, 5, 7), (128, 256)):
        x = rnd(9, shape); X, Y = dev.array(x), dev.zeros(shape[::-1])
        c.transpose_fwd(dev, X, Y); assert np.array_equal(Y.numpy(), x.T)
        d0 = rnd(10, shape); D = dev.array(d0)
        c.transpose_bwd(dev, D, Y); assert np.array_equal(D.numpy(), d0 + x)

    B_, S, H, dh = 2, 5, 3, 8
    x = rnd(11, (B_ * S, H * dh))
    X, Yh = dev.array(x), dev.zeros((B_ * H, S, dh))
    c.split_heads_fwd(dev, X, Yh, B_, S, H, dh)
    want = x.reshape(B_, S, H, dh).transpose(0, 2, 1, 3).reshape(B_ * H, S, dh)
    assert np.array_equal(Yh.numpy(), want)
    # ... which is exactly chunks((S, dh)) in ndarray order (var.rs:401-417)
    for no in range(B_ * H):
        t = np.zeros((S, dh), np.float32); O.chunk_forward(x, t, no)
        assert np.array_equal(t, want[no])
    D = dev.array(x); c.split_heads_bwd(dev, D, Yh, B_, S, H, dh); assert np.array_equal(D.numpy(), 2 * x)
    Z = dev.zeros(x.shape); c.merge_heads_fwd(dev, Yh, Z, B_, S, H, dh); assert np.array_equal(Z.numpy(), x)
    Dh = dev.array(want); c.merge_heads_bwd(dev, Dh, Z, B_, S, H, dh); assert np.array_equal(Dh.numpy(), 2 * want)


# ------------------------------------------------------------------------------ pointwise unary (next row f-2)
UNARY_FIX = {"neg": "negation", "sqrt": "sqrt", "sigmoid": "sigmoid", "tanh": "tanh", "softplus": "softplus",
             "leaky_relu": "leaky_relu", "pow": "power"}


@pytest.mark.parametrize("op", ["neg", "exp", "ln", "sqrt", "sigmoid", "tanh", "softplus", "leaky_relu", "pow"])
def test_unary_fwd_bwd(dev, golden, op):
    c = capi()
    if op in UNARY_FIX:                                   # the reference's own vectors
        fw = golden["nodes"][f"{UNARY_FIX[op]}_forward"]
        e = fw["exp"][0] if fw["exp"] else 0
        X, Y = dev.array(f32(fw["literals"][0], (3, 3))), dev.full((3, 3), 9.0)
        c.unary_fwd(dev, op, X, Y, e)
        close(Y.numpy(), f32(fw["literals"][1], (3, 3)), 0, F16_EPSILON)
    for e in ((3, -3, 2) if op == "pow" else (0,)):       # random, vs the f64 oracle
        lo, hi = (0.2, 3.0) if op in ("ln", "sqrt", "pow") else (-3.0, 3.0)
        x, g, d0 = rnd(1, (257, 129), lo, hi), rnd(2, (257, 129), -1, 1), rnd(3, (257, 129))
        X, G, D, Y = dev.array(x), dev.array(g), dev.array(d0), dev.zeros(x.shape)
        c.unary_fwd(dev, op, X, Y, e)
        y64 = np.zeros(x.shape, np.float64); O.unary_forward(op, x.astype(np.float64), y64, e)
        close(Y.numpy(), y64, rtol=2e-6, atol=1e-7)
        keeps_out = op in O.UNARY_KEEPS_OUTPUT
        c.unary_bwd(dev, op, D, G, None if op == "neg" else (Y if keeps_out else X), e)
        d64 = d0.astype(np.float64)
        O.unary_backward(op, d64, g.astype(np.float64), (Y.numpy() if keeps_out else x).astype(np.float64), e)
        close(D.numpy(), d64, rtol=3e-6, atol=1e-6)
    if op == "leaky_relu":                                # the reference quirk: += 0.01, not 0.01*g
        X, G, D = dev.array(f32([-1.0, 2.0, -3.0, 0.0])), dev.array(f32([5.0, 5.0, 5.0, 5.0])), dev.zeros((4,))
        c.unary_bwd(dev, op, D, G, X)
        assert np.array_equal(D.numpy(), f32([0.01, 5.0, 0.01, 0.01]))


# ------------------------------------------------------------------------------ first-write (_assign) variants
def test_assign_variants_equal_accumulate_into_zeros(dev):
    """Every `_assign` entry point writes, without reading the destination, exactly what its `+=` twin leaves in an
    all-zero destination (the destination is pre-filled with NaN to prove it is not read)."""
    c = capi()
    nan = lambda shape: dev.full(shape, float("nan"))
    x, g, t = rnd(1, (67, 129), -1, 1), rnd(2, (67, 129), -1, 1), rnd(3, (67, 129), -1, 1)
    X, G, T = dev.array(x), dev.array(g), dev.array(t)
    A, Z = nan(x.shape), dev.zeros(x.shape)
    c.relu_bwd(dev, A, G, X, assign=True); c.relu_bwd(dev, Z, G, X)
    assert np.array_equal(A.numpy(), Z.numpy())
    gs = dev.array(np.array(0.7, np.float32))
    for fn in (c.sum_bwd, c.mean_bwd):
        A, Z = nan(x.shape), dev.zeros(x.shape)
        fn(dev, A, gs, assign=True); fn(dev, Z, gs)
        assert np.array_equal(A.numpy(), Z.numpy())
    for red in ("mean", "sum"):
        A, Z = nan(x.shape), dev.zeros(x.shape)
        c.mse_bwd(dev, A, gs, X, T, red, assign=True); c.mse_bwd(dev, Z, gs, X, T, red)
        assert np.array_equal(A.numpy(), Z.numpy())
    for op in ("add", "sub", "mul", "div"):                       # same shape, row-broadcast (column reduction), scalar
        for oshape in (x.shape, (129,), (67, 1), ()):
            o = rnd(20, oshape, 0.5, 1.5); Oo = dev.array(o)
            A, Z = nan(oshape), dev.zeros(oshape)
            c.binary_bwd_right(dev, op, A, G, X, Oo, assign=True); c.binary_bwd_right(dev, op, Z, G, X, Oo)
            assert np.array_equal(A.numpy(), Z.numpy()), (op, oshape)
            A, Z = nan(oshape), dev.zeros(oshape)
            c.binary_bwd_left(dev, op, A, G, X, assign=True); c.binary_bwd_left(dev, op, Z, G, X)
            assert np.array_equal(A.numpy(), Z.numpy()), (op, oshape)
    A, Z = nan((129,)), dev.zeros((129,))
    c.unbroadcast_add(dev, A, G, assign=True); c.unbroadcast_add(dev, Z, G)
    assert np.array_equal(A.numpy(), Z.numpy())
    for op in ("neg", "exp", "sigmoid", "tanh", "leaky_relu", "pow"):
        A, Z = nan(x.shape), dev.zeros(x.shape)
        c.unary_bwd(dev, op, A, G, X, 3, assign=True); c.unary_bwd(dev, op, Z, G, X, 3)
        assert np.array_equal(A.numpy(), Z.numpy())
    for shape, axis in (((67, 129), 1), ((67, 129), 0), ((5, 1024), 1), ((3, 7, 5), 1)):
        yy, gg = rnd(21, shape, 0.01, 1.0), rnd(22, shape, -1, 1)
        Yy, Gg = dev.array(yy), dev.array(gg)
        for fn in (c.softmax_bwd, c.log_softmax_bwd):
            A, Z = nan(shape), dev.zeros(shape)
            fn(dev, A, Gg, Yy, axis, assign=True); fn(dev, Z, Gg, Yy, axis)
            assert np.array_equal(A.numpy(), Z.numpy())
    for train, p in ((True, 0.3), (False, 0.3), (True, 0.0)):
        A, Z = nan(x.shape), dev.zeros(x.shape)
        c.dropout_bwd(dev, A, G, T, p, train, assign=True); c.dropout_bwd(dev, Z, G, T, p, train)
        assert np.array_equal(A.numpy(), Z.numpy())
    gt = rnd(23, (129, 67), -1, 1)
    A, Z = nan(x.shape), dev.zeros(x.shape)
    c.transpose_bwd(dev, A, dev.array(gt), assign=True); c.transpose_bwd(dev, Z, dev.array(gt))
    assert np.array_equal(A.numpy(), Z.numpy())
    gc = rnd(24, (67, 300), -1, 1)
    As, Zs = [nan((67, 129)), nan((67, 171))], [dev.zeros((67, 129)), dev.zeros((67, 171))]
    c.concat_bwd(dev, As, dev.array(gc), 1, assign=True); c.concat_bwd(dev, Zs, dev.array(gc), 1)
    assert all(np.array_equal(a.numpy(), z.numpy()) for a, z in zip(As, Zs))
    gp = rnd(4, (3, 4, 9, 14))
    A, Z = nan((3, 4, 7, 8)), dev.zeros((3, 4, 7, 8))
    c.pad_bwd(dev, A, dev.array(gp), (1, 3), assign=True); c.pad_bwd(dev, Z, dev.array(gp), (1, 3))
    assert np.array_equal(A.numpy(), Z.numpy())
    B, S, H, dh = 2, 10, 3, 8
    gh, gf = rnd(5, (B * H, S, dh)), rnd(6, (B * S, H * dh))
    A, Z = nan((B * S, H * dh)), dev.zeros((B * S, H * dh))
    c.split_heads_bwd(dev, A, dev.array(gh), B, S, H, dh, assign=True); c.split_heads_bwd(dev, Z, dev.array(gh), B, S, H, dh)
    assert np.array_equal(A.numpy(), Z.numpy())
    A, Z = nan((B * H, S, dh)), dev.zeros((B * H, S, dh))
    c.merge_heads_bwd(dev, A, dev.array(gf), B, S, H, dh, assign=True); c.merge_heads_bwd(dev, Z, dev.array(gf), B, S, H, dh)
    assert np.array_equal(A.numpy(), Z.numpy())
    for xs, ws, st, dl, gr in [((3, 4, 9, 8), (6, 2, 3, 3), (1, 1), (1, 1), 2), ((2, 32, 12, 12), (32, 32, 3, 3), (1, 1), (1, 1), 1),
                               ((2, 3, 11), (4, 3, 3), (2,), (1,), 1),
                               # stride phases of the fast input-gradient pass: none / three of four / three of four without a tap (those
                               # positions are zeroed by one memset on a first write and left alone by `+=`)
                               ((2, 32, 13, 14), (32, 32, 3, 3), (2, 2), (1, 1), 1), ((2, 32, 12, 11), (64, 32, 1, 1), (2, 2), (1, 1), 1),
                               ((2, 32, 17, 19), (32, 32, 3, 3), (2, 2), (2, 2), 1)]:
        xx, ww = rnd(9, xs, -1, 1), rnd(10, ws, -1, 1)
        osp = tuple((n - d * (k - 1) - 1) // s + 1 for n, k, s, d in zip(xs[2:], ws[2:], st, dl))
        gy = rnd(11, (xs[0], ws[0]) + osp, -1, 1)
        Xc, Wc, Gc = dev.array(xx), dev.array(ww), dev.array(gy)
        A, Z = nan(xs), dev.zeros(xs)
        c.conv_bwd_input(dev, A, Gc, Wc, st, dl, gr, assign=True); c.conv_bwd_input(dev, Z, Gc, Wc, st, dl, gr)
        assert np.array_equal(A.numpy(), Z.numpy())
        A, Z = nan(ws), dev.zeros(ws)
        c.conv_bwd_kernel(dev, A, Gc, Xc, st, dl, gr, assign=True); c.conv_bwd_kernel(dev, Z, Gc, Xc, st, dl, gr)
        assert np.array_equal(A.numpy(), Z.numpy())
        bias = rnd(12, (ws[0],) + (1,) * len(osp), -1, 1)
        Y0, Y1, Y2 = dev.zeros(gy.shape), dev.zeros(gy.shape), nan(gy.shape)
        c.conv_fwd(dev, Xc, Wc, Y0, st, dl, gr); c.binary_fwd(dev, "add", Y1, Y0, dev.array(bias))
        c.conv_fwd(dev, Xc, Wc, Y2, st, dl, gr, bias=dev.array(bias))
        assert np.array_equal(Y1.numpy(), Y2.numpy())                 # conv + bias in the epilogue == two nodes
    sc = rnd(7, (6, 40, 64), -2, 2)
    Sx, P, O_, Gs = dev.array(sc), dev.zeros(sc.shape), dev.zeros(sc.shape), dev.array(rnd(8, sc.shape, -1, 1))
    c.scale_softmax_dropout_fwd(dev, Sx, P, O_, None, 0.125, 0.2, True, 11, 5)
    A, Z = nan(sc.shape), dev.zeros(sc.shape)
    c.scale_softmax_dropout_bwd(dev, A, Gs, P, None, 0.125, 0.2, True, 11, 5, assign=True)
    c.scale_softmax_dropout_bwd(dev, Z, Gs, P, None, 0.125, 0.2, True, 11, 5)
    assert np.array_equal(A.numpy(), Z.numpy())
    # probabilities recomputed from the scores in the backward pass (forward called with probs = NULL): same bits
    O2 = dev.zeros(sc.shape)
    c.scale_softmax_dropout_fwd(dev, Sx, None, O2, None, 0.125, 0.2, True, 11, 5)
    assert np.array_equal(O2.numpy(), O_.numpy())
    for assign in (True, False):
        R = nan(sc.shape) if assign else dev.zeros(sc.shape)
        c.scale_softmax_dropout_bwd_from_scores(dev, R, Gs, Sx, None, 0.125, 0.2, True, 11, 5, assign=assign)
        assert np.array_equal(R.numpy(), Z.numpy())


@pytest.mark.parametrize("rows,L", [(512, 1024), (96, 2048), (300, 516), (64, 128)])
def test_attention_probs_paths_agree_at_benchmark_width(dev, rows, L):
    """At the row lengths the C5 bench runs (L = 1024: a different template instance from the L = 64 case above):
    probabilities stored by the forward and recomputed from the scores in the backward (what the bench runs) agree
    BIT for bit; the three reference nodes Multiplication(scalar) -> Softmax -> Dropout through the C ABI give the
    same keep/drop pattern exactly and the same values to f32 rounding; all match the oracle fed the same Philox mask."""
    c = capi()
    p, scale, seed, off = 0.1, 0.125, 77, 1000
    sc, g = rnd(7, (rows, L), -3, 3), rnd(8, (rows, L), -1, 1)
    Sx, Gs = dev.array(sc), dev.array(g)
    P, O1, O2 = dev.zeros(sc.shape), dev.zeros(sc.shape), dev.zeros(sc.shape)
    c.scale_softmax_dropout_fwd(dev, Sx, P, O1, None, scale, p, True, seed, off)
    c.scale_softmax_dropout_fwd(dev, Sx, None, O2, None, scale, p, True, seed, off)
    assert np.array_equal(O1.numpy(), O2.numpy())
    D1, D2 = dev.zeros(sc.shape), dev.zeros(sc.shape)
    c.scale_softmax_dropout_bwd(dev, D1, Gs, P, None, scale, p, True, seed, off)
    c.scale_softmax_dropout_bwd_from_scores(dev, D2, Gs, Sx, None, scale, p, True, seed, off, assign=False)
    assert np.array_equal(D1.numpy(), D2.numpy())
    # three reference nodes through the C ABI
    Ssc, Psm, Od, Nz = dev.zeros(sc.shape), dev.zeros(sc.shape), dev.zeros(sc.shape), dev.zeros(sc.shape)
    c.binary_fwd(dev, "mul", Ssc, Sx, dev.array(np.float32(scale).reshape(())))
    c.softmax_fwd(dev, Ssc, Psm, 1)
    c.dropout_fwd(dev, Psm, Od, Nz, p, True, seed, off)
    noise = O.dropout_noise(rows * L, p, seed, off).reshape(rows, L)
    assert np.array_equal(Nz.numpy(), noise)                                   # the mask, bit for bit
    assert np.array_equal(O1.numpy() != 0, (noise != 0) & (P.numpy() != 0))    # fused node drops exactly the same elements
    np.testing.assert_allclose(Psm.numpy(), P.numpy(), rtol=2e-6, atol=1e-30)
    np.testing.assert_allclose(Od.numpy(), O1.numpy(), rtol=2e-6, atol=1e-30)
    Dp, Dsm, Dsc = dev.zeros(sc.shape), dev.zeros(sc.shape), dev.zeros(sc.shape)
    c.dropout_bwd(dev, Dp, Gs, Nz, p, True)
    c.softmax_bwd(dev, Dsm, Dp, Psm, 1)
    c.binary_bwd_left(dev, "mul", Dsc, Dsm, dev.array(np.float32(scale).reshape(())))
    np.testing.assert_allclose(Dsc.numpy(), D1.numpy(), rtol=1e-5, atol=1e-7)
    # and the oracle (f64) on the same mask
    s64 = sc.astype(np.float64) * np.float64(np.float32(scale))
    pr = np.zeros_like(s64); O.softmax_forward(s64, pr, 1)
    od = np.zeros_like(pr); O.dropout_forward(pr, od, noise.astype(np.float64), p, True)
    np.testing.assert_allclose(O1.numpy(), od, rtol=2e-5, atol=1e-9)
    dpr = np.zeros_like(pr); O.dropout_backward(dpr, g.astype(np.float64), noise.astype(np.float64), p, True)
    dsm = np.zeros_like(pr); O.softmax_backward(dsm, dpr, pr, 1)
    np.testing.assert_allclose(D1.numpy(), dsm * np.float64(np.float32(scale)), rtol=1e-4, atol=2e-7)


# ------------------------------------------------------------------------------ fused Linear forward
@pytest.mark.parametrize("n,m,o", [(64, 3, 5), (4, 8, 1), (128, 128, 128), (300, 77, 200), (8, 4096, 64), (512, 256, 384)])
def test_linear_fwd_equals_mm_t_plus_bias(dev, n, m, o):
    """nk_linear_fwd (bias in the GEMM epilogue) is BIT-identical to nk_mm_t_fwd followed by the broadcast
    Addition node, on aligned, ragged and split-K shapes."""
    c = capi()
    x, w, b = rnd(1, (n, m), -1, 1), rnd(2, (o, m), -1, 1), rnd(3, (o,), -1, 1)
    X, W, Bv = dev.array(x), dev.array(w), dev.array(b)
    Y0, Y1, Y2 = dev.zeros((n, o)), dev.zeros((n, o)), dev.full((n, o), 3.0)
    c.mm_t_fwd(dev, X, W, Y0)
    c.binary_fwd(dev, "add", Y1, Y0, Bv)
    c.linear_fwd(dev, X, W, Bv, Y2)
    assert np.array_equal(Y1.numpy(), Y2.numpy())
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    from tolerance import assert_contraction
    assert_contraction("linear_fwd_equals_mm_t_plus_bias", Y2.numpy(), ref, m, cpu32=x @ w.T + b, epilogue=True)


# ------------------------------------------------------------------------------ loss criteria (next row f-4)
def _lin(spec):
    a, b, n, *shape = spec
    return np.linspace(a, b, int(n), dtype=np.float32).reshape(shape)


LOSS_ORACLE = {"mae": (O.mae_forward, O.mae_backward), "bce": (O.bce_forward, O.bce_backward),
               "bce_with_logits": (O.bce_with_logits_forward, O.bce_with_logits_backward)}


@pytest.mark.parametrize("red", ["mean", "sum"])
def test_loss_golden(dev, golden, red):
    c = capi(); n = golden["nodes"]
    for name, key in (("bce", "bce"), ("mae", "absolute_error")):
        cs = n[f"{key}_forward_base_case_{red}"]
        x, t = (_lin(sp) for sp in cs["linspace_start_stop_n_shape"])
        X, T, out = dev.array(x), dev.array(t), dev.zeros(())
        c.loss_fwd(dev, name, X, T, out, red)
        np.testing.assert_allclose(out.numpy(), cs["scalars"][-1], rtol=2e-6, atol=cs["tol"])
        cs = n[f"{key}_backward_base_case_{red}"]
        D, G = dev.zeros((3, 3)), dev.array(f32([cs["scalars"][0]]).reshape(()))
        c.loss_bwd(dev, name, D, G, X, T, red)
        want = f32(cs["literals"][0], (3, 3)) if cs["literals"] else np.full((3, 3), cs["from_elem"][0][0], np.float32)
        np.testing.assert_allclose(D.numpy(), want, rtol=2e-6, atol=cs["tol"])
    cs = n[f"nll_{red}"]; lit = cs["literals"]
    lx = np.zeros((3, 5), np.float32); O.log_softmax_forward(f32(lit[1], (3, 5)), lx, 1)
    X, T, out, G = dev.array(lx), dev.array(f32(lit[0])), dev.zeros(()), dev.array(np.ones((), np.float32))
    c.nll_fwd(dev, X, T, out, red); close(out.numpy(), cs["scalars"][0], 0, F16_EPSILON)
    D = dev.zeros((3, 5))
    c.nll_bwd(dev, D, G, T, red); close(D.numpy(), f32(lit[3], (3, 5)), 0, F16_EPSILON)
    c.nll_bwd(dev, D, G, T, red); close(D.numpy(), 2 * f32(lit[4], (3, 5)), 0, F16_EPSILON)
    cs = n[f"kldiv_{red}"]; lit = cs["literals"]
    X, T = dev.array(np.log(f32(lit[1], (2, 3)))), dev.array(f32(lit[0], (2, 3)))
    c.loss_fwd(dev, "kldiv", X, T, out, red); close(out.numpy(), cs["scalars"][0], 0, F16_EPSILON)
    D = dev.zeros((2, 3))
    c.loss_bwd(dev, "kldiv", D, G, None, T, red); close(D.numpy(), f32(lit[3], (2, 3)), 0, F16_EPSILON)
    cs = n[f"bce_with_logits_{red}"]; lit = cs["literals"]
    X, T = dev.array(f32(lit[1], (3, 3))), dev.array(f32(lit[0], (3, 3)))
    c.loss_fwd(dev, "bce_with_logits", X, T, out, red); close(out.numpy(), cs["scalars"][0], 0, 1e-3)
    D = dev.zeros((3, 3))
    c.loss_bwd(dev, "bce_with_logits", D, G, X, T, red); close(D.numpy(), f32(lit[3], (3, 3)), 0, F16_EPSILON)


@pytest.mark.parametrize("red", ["mean", "sum"])
@pytest.mark.parametrize("shape", [(7,), (33, 129), (4, 3, 50, 51), (1 << 20,)])
def test_loss_random(dev, red, shape):
    """f32 HIP vs the f64 oracle; forward tolerance covers the parallel (vs sequential) summation order."""
    c = capi()
    gval = np.float32(0.75)
    G = dev.array(np.array(gval, np.float32))
    for name in ("mae", "bce", "bce_with_logits", "kldiv"):
        if name in ("bce", "kldiv"):
            x, t = rnd(1, shape, 0.02, 0.98), rnd(2, shape, 0.0, 1.0)
            if name == "kldiv":
                t = np.where(t < 0.1, 0.0, t).astype(np.float32); x = np.log(x)
        else:
            x, t = rnd(1, shape, -4, 4), rnd(2, shape, 0.0, 1.0)
            if name == "mae":
                t.reshape(-1)[::5] = x.reshape(-1)[::5]          # exact ties: gradient 0 there
        d0 = rnd(3, shape, -1, 1)
        X, T, D, out = dev.array(x), dev.array(t), dev.array(d0), dev.zeros(())
        c.loss_fwd(dev, name, X, T, out, red)
        x64, t64, d64 = x.astype(np.float64), t.astype(np.float64), d0.astype(np.float64)
        if name == "kldiv":
            want = O.kldiv_forward(x64, t64, red); O.kldiv_backward(d64, float(gval), t64, red)
        else:
            fwd, bwd = LOSS_ORACLE[name]
            want = fwd(x64, t64, red); bwd(d64, float(gval), x64, t64, red)
        close(out.numpy(), want, rtol=2e-5, atol=1e-6)
        c.loss_bwd(dev, name, D, G, None if name == "kldiv" else X, T, red)
        close(D.numpy(), d64, rtol=3e-6, atol=1e-6)


@pytest.mark.parametrize("red", ["mean", "sum"])
@pytest.mark.parametrize("shape", [(64, 10), (5, 7, 6, 4), (4096, 1000)])
def test_nll_random(dev, red, shape):
    c = capi()
    x = np.log(rnd(1, shape, 0.01, 1.0))
    tshape = (shape[0],) + tuple(shape[2:])
    t = np.random.default_rng(5).integers(-1, shape[1] + 2, tshape).astype(np.float32)  # incl. out-of-range / negative
    t.reshape(-1)[::7] += 0.75                                                          # fraction is dropped
    d0 = rnd(3, shape, -1, 1)
    X, T, D, out, G = dev.array(x), dev.array(t), dev.array(d0), dev.zeros(()), dev.array(np.array(0.5, np.float32))
    c.nll_fwd(dev, X, T, out, red)
    close(out.numpy(), O.nll_forward(x.astype(np.float64), t, red), rtol=2e-5, atol=1e-6)
    c.nll_bwd(dev, D, G, T, red)
    d = d0.copy(); O.nll_backward(d, 0.5, t, red)
    assert np.array_equal(D.numpy(), d)


# ------------------------------------------------------------------------------ Linear + ReLU in the GEMM epilogues
@pytest.mark.parametrize("n,m,o", [(128, 128, 256), (96, 40, 64), (257, 131, 77), (64, 8192, 64), (256, 4096, 512), (1536, 1536, 1536),
                                   (3072, 128, 3072), (1, 1, 1)])
def test_linear_relu_fused_epilogues_equal_the_separate_nodes(dev, n, m, o):
    """nk_linear_relu_fwd == nk_linear_fwd + nk_relu_fwd and nk_linear_bwd_input_relu == nk_mm_t_bwd_left + nk_relu_bwd, BIT FOR
    BIT (the ReLU / the mask act on the f32 value the epilogue is about to store), on every tile / split-K path (the second
    pass applies them), `+=` and first-write forms, with NaN / inf in the operands (the reference's `o.max(0.)` and
    `((x > 0.) as f32) * g`, relu/mod.rs:33-37,71-78); and the in-place mask used as the fallback."""
    c = capi()
    x, w, b = rnd(1, (n, m), -1, 1), rnd(2, (o, m), -1, 1), rnd(3, (o,), -1, 1)
    if n * m > 16:
        x[1 % n, 3 % m] = np.nan; x[2 % n, 1 % m] = np.inf; w[5 % o, 2 % m] = -np.inf
    X, W, B = dev.array(x), dev.array(w), dev.array(b)
    Z, A1, A2 = dev.full((n, o), 3.0), dev.full((n, o), 5.0), dev.full((n, o), 7.0)
    c.linear_fwd(dev, X, W, B, Z); c.relu_fwd(dev, Z, A1)
    c.linear_relu_fwd(dev, X, W, B, A2)
    a = A1.numpy()
    assert np.array_equal(A2.numpy(), a) and not np.isnan(a).any()
    assert np.array_equal(a > 0, Z.numpy() > 0)                       # the output IS the mask of the pre-activation
    # backward of a FOLLOWING Linear(o -> p) whose input is `a`: dZ (+)= (a > 0) * (G . W2)
    pdim = 48 if o < 1000 else 256
    g, w2 = rnd(4, (n, pdim), -1, 1), rnd(5, (pdim, o), -1, 1)
    if n * pdim > 16:
        g[0, 0] = np.inf; g[n - 1, pdim - 1] = np.nan
    G, W2 = dev.array(g), dev.array(w2)
    d0 = rnd(6, (n, o), -1, 1)
    for assign in (False, True):
        GA = dev.zeros((n, o))
        c.sgemm(dev, 0, 0, n, o, pdim, 1.0, G, pdim, W2, o, 0.0, GA, o)        # nk_mm_t_bwd_left into a fresh gradient of a
        D1 = dev.array(d0); c.relu_bwd(dev, D1, GA, Z, assign=assign)          # ... then ReLUBackward on the INPUT z
        D2 = dev.array(d0); c.linear_bwd_input_relu(dev, D2, G, W2, A2, assign=assign)
        assert np.array_equal(D1.numpy(), D2.numpy(), equal_nan=True), assign
        GM = dev.array(GA.numpy()); c.relu_mask_inplace(dev, GM, A2)
        want = np.zeros_like(d0); O.relu_backward(want, GA.numpy(), Z.numpy())
        assert np.array_equal(GM.numpy(), want, equal_nan=True)
        c.relu_mask_inplace(dev, GM, A2)                                       # idempotent
        assert np.array_equal(GM.numpy(), want, equal_nan=True)


class _View:
    """`n` floats of a device array starting `off` floats in (pointer arithmetic on the host, as a Rust slice would)."""
    def __init__(self, a, off, n):
        import ctypes
        self.keep, self.p, self.size, self.shape = a, ctypes.c_void_p(a.p.value + 4 * off), n, (n,)


def test_sgd_step_multi_equals_one_launch_per_parameter(dev):
    """nk_sgd_step_multi: bit-identical to nk_sgd_step parameter by parameter - plain, momentum, Nesterov + dampening, with
    penalties (the gradient is penalised in place); sizes around the 4096-element chunk, a 4-byte-aligned parameter, more
    than eight parameters (two launches), an empty one."""
    c = capi()
    sizes = [4096 * 3, 4097, 5, 0, 1 << 20, 4096, 123457, 8191, 64, 12288, 1000]
    for kw in (dict(lr=0.05), dict(lr=0.05, momentum=0.9), dict(lr=0.01, momentum=0.8, dampening=0.1, nesterov=True),
               dict(lr=0.05, momentum=0.9, l1=1e-3, l2=2e-3), dict(lr=0.1, l2=1e-2)):
        ws = [rnd(10 + i, (sz + 1,), -1, 1) for i, sz in enumerate(sizes)]
        gs = [rnd(40 + i, (sz + 1,), -1, 1) for i, sz in enumerate(sizes)]
        vs = [rnd(70 + i, (sz + 1,), -1, 1) for i, sz in enumerate(sizes)]
        mom = kw.get("momentum", 0.0) > 0
        ref, got = [], []
        for which in (0, 1):
            W = [dev.array(w) for w in ws]; G = [dev.array(g) for g in gs]; V = [dev.array(v) for v in vs]
            # parameter 7 starts one float into its buffer: 4-byte aligned only
            view = lambda a, i: _View(a, 1 if i == 7 else 0, sizes[i])
            Wv, Gv, Vv = [view(a, i) for i, a in enumerate(W)], [view(a, i) for i, a in enumerate(G)], [view(a, i) for i, a in enumerate(V)]
            if which == 0:
                for w_, g_, v_ in zip(Wv, Gv, Vv):
                    c.sgd_step(dev, w_, g_, v_ if mom else None, **kw)
            else:
                c.sgd_step_multi(dev, Wv, Gv, Vv if mom else None, **kw)
            (ref if which == 0 else got).extend([a.numpy() for a in W] + [a.numpy() for a in G] + [a.numpy() for a in V])
        for r, g_ in zip(ref, got):
            assert np.array_equal(r, g_), kw


def test_round4_entry_points_reject_bad_arguments(dev):
    """Status codes, not crashes: nk_dev_tune (unknown knob, too many values, out-of-range modes), nk_sgd_step_multi (null tables),
    the fused Linear+ReLU forms (null bias / mask operand), nk_attention_qkv_* (null packed pointer)."""
    import ctypes as C
    c = capi()
    E = c.NeuronikaHipError
    with pytest.raises(E, match="unknown tuning knob"):
        dev.tune(99, [1])
    with pytest.raises(E, match="at most 6"):
        dev.tune(c.TUNE_GEMM_FORCE, [2, 2, 1, 1, 8, 16, 3])
    with pytest.raises(E, match="KPAIR"):
        dev.tune(c.TUNE_GEMM_KPAIR, 7)
    with pytest.raises(E, match="ATTENTION_OCC"):
        dev.tune(c.TUNE_ATTENTION_OCC, 3)
    dev.gemm_force(None); dev.gemm_kpair(None); dev.tune(c.TUNE_ATTENTION_OCC, None)
    with pytest.raises(E, match="null table"):
        c.check(c.lib.nk_sgd_step_multi(dev.h, 2, None, None, None, None, 0.1, 0.0, 0.0, 0, 0.0, 0.0))
    c.check(c.lib.nk_sgd_step_multi(dev.h, 0, None, None, None, None, 0.1, 0.0, 0.0, 0, 0.0, 0.0))        # nothing to do
    X, W, Y = dev.zeros((8, 8)), dev.zeros((8, 8)), dev.zeros((8, 8))
    with pytest.raises(E, match="null bias"):
        c.check(c.lib.nk_linear_relu_fwd(dev.h, X.p, W.p, None, Y.p, 8, 8, 8))
    with pytest.raises(E, match="null mask"):
        c.check(c.lib.nk_linear_bwd_input_relu(dev.h, Y.p, X.p, W.p, None, 8, 8, 8, 1))
    with pytest.raises(E, match="null pointer"):
        c.check(c.lib.nk_attention_qkv_fwd(dev.h, None, None, None, None, Y.p, 1, 8, 1, 64, 0.125, 0.0, 1, 0, 0))
    with pytest.raises(E, match="bad nk_comm_init_all"):
        c.check(c.lib.nk_comm_init_all(0, None, None))


# ------------------------------------------------------------------------------ GEMV / dot (next row f-4)
def test_gemv_dot_golden(dev, golden):
    c = capi(); n = golden["nodes"]
    lit = n["matrix_vector_mul_forward"]["literals"]
    A, x, y = dev.array(f32(lit[0], (3, 3))), dev.array(f32(lit[1])), dev.full((3,), 9.0)
    c.mv_fwd(dev, A, x, y); assert np.array_equal(y.numpy(), f32(lit[2]))
    lit = n["matrix_vector_mul_backward"]["literals"]
    dA, dx, A, x, g = (dev.array(f32(lit[0], (3, 3))), dev.array(f32(lit[1])), dev.array(f32(lit[2], (3, 3))),
                       dev.array(f32(lit[3])), dev.array(f32(lit[4])))
    for k in (6, 8):
        c.mv_bwd_left(dev, dA, g, x); c.mv_bwd_right(dev, dx, A, g)
        assert np.array_equal(dA.numpy(), f32(lit[k], (3, 3))) and np.array_equal(dx.numpy(), f32(lit[k + 1]))
    lit = n["vector_matrix_mul_forward"]["literals"]
    v, B, y = dev.array(f32(lit[0])), dev.array(f32(lit[1], (3, 3))), dev.full((3,), 9.0)
    c.vm_fwd(dev, v, B, y); assert np.array_equal(y.numpy(), f32(lit[2]))
    lit = n["vector_matrix_mul_backward"]["literals"]
    dv, dB, v, B, g = (dev.array(f32(lit[0])), dev.array(f32(lit[1], (3, 3))), dev.array(f32(lit[2])),
                       dev.array(f32(lit[3], (3, 3))), dev.array(f32(lit[4])))
    for k in (6, 8):
        c.vm_bwd_left(dev, dv, B, g); c.vm_bwd_right(dev, dB, v, g)
        assert np.array_equal(dv.numpy(), f32(lit[k])) and np.array_equal(dB.numpy(), f32(lit[k + 1], (3, 3)))
    cs = n["vector_vector_mul_forward"]
    l, r, out = dev.array(f32(cs["literals"][0])), dev.array(f32(cs["literals"][1])), dev.zeros(())
    c.vv_fwd(dev, l, r, out); assert out.numpy() == cs["scalars"][0]
    cs = n["vector_vector_mul_backward"]; lit = cs["literals"]
    dl, dr, l, r, g = dev.array(f32(lit[0])), dev.array(f32(lit[1])), dev.array(f32(lit[2])), dev.array(f32(lit[3])), dev.array(np.ones((), np.float32))
    for k in (4, 6):
        c.vv_bwd(dev, dl, r, g); c.vv_bwd(dev, dr, l, g)
        assert np.array_equal(dl.numpy(), f32(lit[k])) and np.array_equal(dr.numpy(), f32(lit[k + 1]))


@pytest.mark.parametrize("n,m", [(1, 1), (5, 3), (64, 1000), (1000, 64), (4099, 257), (3, 70001), (2048, 2048), (8192, 512)])
def test_gemv_random(dev, n, m):
    c = capi()
    a, x, g = rnd(1, (n, m), -1, 1), rnd(2, (m,), -1, 1), rnd(3, (n,), -1, 1)
    a64, x64, g64 = a.astype(np.float64), x.astype(np.float64), g.astype(np.float64)
    tol = dict(rtol=1e-5, atol=2e-6 * np.sqrt(max(n, m)))
    A, X, G = dev.array(a), dev.array(x), dev.array(g)
    Y = dev.full((n,), 5.0); c.mv_fwd(dev, A, X, Y); close(Y.numpy(), a64 @ x64, **tol)
    d0 = rnd(4, (n, m)); D = dev.array(d0); c.mv_bwd_left(dev, D, G, X); close(D.numpy(), d0 + np.outer(g64, x64), **tol)
    e0 = rnd(5, (m,)); E = dev.array(e0); c.mv_bwd_right(dev, E, A, G); close(E.numpy(), e0 + a64.T @ g64, **tol)
    # vector . matrix with B = a (n, m): v (n), result (m)
    Z = dev.full((m,), 5.0); c.vm_fwd(dev, G, A, Z); close(Z.numpy(), g64 @ a64, **tol)
    f0 = rnd(6, (n,)); F = dev.array(f0); c.vm_bwd_left(dev, F, A, X); close(F.numpy(), f0 + a64 @ x64, **tol)
    D = dev.array(d0); c.vm_bwd_right(dev, D, G, X); close(D.numpy(), d0 + np.outer(g64, x64), **tol)
    out = dev.zeros(()); c.vv_fwd(dev, X, dev.array(e0), out); close(out.numpy(), x64 @ e0.astype(np.float64), **tol)
    s = dev.array(np.array(-1.5, np.float32)); E = dev.array(e0); c.vv_bwd(dev, E, X, s); close(E.numpy(), e0 - 1.5 * x64, **tol)


# ------------------------------------------------------------------------------ optimizer (next row)
def test_optimizer_steps(dev):
    """neuronika-optim update rules on the device vs the oracle restatement, several steps each
    (state carried on the device), with and without penalties."""
    c = capi()
    n = 4097
    w0, g0 = rnd(1, (n,), -1, 1), rnd(2, (n,), -1, 1)
    w0[:3] = (0.0, -0.0, 0.5)                                   # Rust signum(+-0) = +-1 under L1

    def run(dev_step, ora_step, nstate, steps=3, tol=2e-6):
        W, ww = dev.array(w0), w0.copy()
        ST, st = [dev.zeros((n,)) for _ in range(nstate)], [np.zeros(n, np.float32) for _ in range(nstate)]
        for k in range(1, steps + 1):
            gk = (g0 * np.float32(1.0 / k)).astype(np.float32)
            G, gg = dev.array(gk), gk.copy()
            dev_step(W, G, ST, k); ora_step(ww, gg, st, k)
            close(W.numpy(), ww, tol, tol); close(G.numpy(), gg, 1e-6, 1e-7)
            for a, b in zip(ST, st):
                close(a.numpy(), b, tol, tol)

    run(lambda W, G, S, k: c.sgd_step(dev, W, G, None, lr=0.1, l1=0.02, l2=0.01),
        lambda w, g, s, k: O.sgd_step(w, g, 0.1, l1=0.02, l2=0.01), 0)
    run(lambda W, G, S, k: c.sgd_step(dev, W, G, S[0], lr=0.1, momentum=0.9, dampening=0.1, nesterov=True),
        lambda w, g, s, k: O.sgd_step(w, g, 0.1, s[0], 0.9, 0.1, True), 1)
    run(lambda W, G, S, k: c.sgd_step(dev, W, G, S[0], lr=0.1, momentum=0.5),
        lambda w, g, s, k: O.sgd_step(w, g, 0.1, s[0], 0.5), 1)
    run(lambda W, G, S, k: c.adam_step(dev, W, G, S[0], S[1], None, 1e-2, 0.9, 0.999, 1e-8, k, l2=0.01),
        lambda w, g, s, k: O.adam_step(w, g, s[0], s[1], 1e-2, 0.9, 0.999, 1e-8, k, l2=0.01), 2, tol=1e-5)
    run(lambda W, G, S, k: c.adam_step(dev, W, G, S[0], S[1], S[2], 1e-2, 0.9, 0.999, 1e-8, k),
        lambda w, g, s, k: O.adam_step(w, g, s[0], s[1], 1e-2, 0.9, 0.999, 1e-8, k, max_exp_avg_sq=s[2]), 3, tol=1e-5)
    run(lambda W, G, S, k: c.adagrad_step(dev, W, G, S[0], 1e-2, 0.1, 1e-10, k, l1=0.01),
        lambda w, g, s, k: O.adagrad_step(w, g, s[0], 1e-2, 0.1, 1e-10, k, l1=0.01), 1, tol=1e-5)
    for centered in (False, True):
        for mom in (0.0, 0.9):
            run(lambda W, G, S, k: c.rmsprop_step(dev, W, G, S[0], S[1] if centered else None, S[2] if mom else None,
                                                  1e-2, 0.99, 1e-8, mom),
                lambda w, g, s, k: O.rmsprop_step(w, g, s[0], 1e-2, 0.99, 1e-8, s[1] if centered else None,
                                                  s[2] if mom else None, mom), 3, tol=2e-5)


# ------------------------------------------------------------------------------ composed MLP step
def test_mlp_step_through_c_abi(dev):
    """C1-shaped plumbing: the tape order of `Linear -> ReLU -> Linear -> MSE` issued as raw C-ABI
    calls (the C++ tape mirror is covered in test_gpu_tape.py)."""
    c = capi()
    rng = np.random.default_rng(0)
    n, din, dh, dout = 64, 3, 5, 1
    x, t = rng.random((n, din), dtype=np.float32), rng.random((n, dout), dtype=np.float32)
    w1, b1 = (rng.random((dh, din), dtype=np.float32) - .5), (rng.random(dh, dtype=np.float32) - .5)
    w2, b2 = (rng.random((dout, dh), dtype=np.float32) - .5), (rng.random(dout, dtype=np.float32) - .5)
    loss, grads = O.mlp_step(x, t, [(w1, b1), (w2, b2)])
    X, T, W1, B1, W2, B2 = map(dev.array, (x, t, w1, b1, w2, b2))
    Z1, H1, A1, Z2, H2, LOSS = dev.zeros((n, dh)), dev.zeros((n, dh)), dev.zeros((n, dh)), dev.zeros((n, dout)), dev.zeros((n, dout)), dev.zeros(())
    c.mm_t_fwd(dev, X, W1, Z1); c.binary_fwd(dev, "add", H1, Z1, B1); c.relu_fwd(dev, H1, A1)
    c.mm_t_fwd(dev, A1, W2, Z2); c.binary_fwd(dev, "add", H2, Z2, B2); c.mse_fwd(dev, H2, T, LOSS, "mean")
    close(LOSS.item(), loss, 1e-5, 1e-6)
    GL = dev.full((), 1.0)
    dH2, dZ2, dB2, dW2, dA1, dH1, dZ1, dB1, dW1 = (dev.zeros(s) for s in ((n, dout), (n, dout), (dout,), (dout, dh), (n, dh), (n, dh), (n, dh), (dh,), (dh, din)))
    c.mse_bwd(dev, dH2, GL, H2, T, "mean")
    c.binary_bwd_left(dev, "add", dZ2, dH2); c.binary_bwd_right(dev, "add", dB2, dH2)
    c.mm_t_bwd_left(dev, dA1, dZ2, W2); c.mm_t_bwd_right(dev, dW2, dZ2, A1)
    c.relu_bwd(dev, dH1, dA1, H1)
    c.binary_bwd_left(dev, "add", dZ1, dH1); c.binary_bwd_right(dev, "add", dB1, dH1)
    c.mm_t_bwd_right(dev, dW1, dZ1, X)
    close(dW1.numpy(), grads[0][0], 1e-4, 1e-6); close(dB1.numpy(), grads[0][1], 1e-4, 1e-6)
    close(dW2.numpy(), grads[1][0], 1e-4, 1e-6); close(dB2.numpy(), grads[1][1], 1e-4, 1e-6)
