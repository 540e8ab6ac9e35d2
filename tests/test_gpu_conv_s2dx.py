"""The fused-phase input gradient of the 3 x 3 / stride-2 convolution (csrc/nk_conv_s2dx.h: the four stride phases of a tile in one block
walk) against the per-phase implicit-GEMM kernels it replaces by rule (NK_TUNE_CONV_S2DX = 0) and the oracle
(`convolution_backward_input`, node/convolution/mod.rs:146-189, 256-274): bit for bit on integer-valued data (every partial sum exact
in f32), inside the suite's contraction bound on random data; `+=` and first-write forms; the module's Zero padding folded in (1) and the
gradient of an already padded input (0); both block shapes; the geometries the kernel does not take stay with the per-phase kernels."""
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


def ints(seed, shape, lo, hi):
    return np.random.default_rng(seed).integers(lo, hi + 1, shape).astype(np.float32)


def rnd(seed, shape, lo=0.0, hi=1.0):
    a = np.random.default_rng(seed).random(shape, dtype=np.float32)
    return np.asarray(a * np.float32(hi - lo) + np.float32(lo), dtype=np.float32)


# N, Cin, Cout, H, W of the UNPADDED input (even), padding (both axes)
SHAPES = [
    (2, 64, 128, 8, 8, 1),        # narrow blocks (64 input channels), one block of super-pixels per sample and a half
    (3, 64, 64, 6, 10, 1),        # 45 super-pixels: a partly filled last block
    (2, 128, 256, 8, 8, 1),       # wide blocks (128 input channels), chunks of 32 reduction channels
    (1, 256, 128, 12, 4, 1),      # two wide channel blocks
    (2, 192, 48, 6, 6, 1),        # 192 = 3 x 64: narrow only; 48 reduction channels = three chunks of 16
    (2, 64, 128, 10, 10, 0),      # the gradient of an already padded input: neighbours (a - 1, b - 1) .. (a, b); output 4 x 4
    (3, 128, 64, 8, 12, 0),
    (5, 64, 64, 2, 2, 1),         # one super-pixel per sample: every neighbour but one out of range
    (4, 64, 128, 56, 56, 1),      # the layer of the shape table (3 x 3 s2 64 -> 128 at 56 x 56), four samples
]


def run(dev, go, w, dx0, xs, pad, mode):
    c = capi()
    dev.conv_s2dx(mode)
    try:
        G, Wd = dev.array(go), dev.array(w)
        DX, DXa = dev.array(dx0), dev.full(xs, np.nan)
        padding = (pad, pad) if pad else None
        c.conv_bwd_input(dev, DX, G, Wd, (2, 2), (1, 1), 1, padding=padding)
        c.conv_bwd_input(dev, DXa, G, Wd, (2, 2), (1, 1), 1, assign=True, padding=padding)
        return DX.numpy(), DXa.numpy()
    finally:
        dev.conv_s2dx(None)


def oracle_dx(go, w, xs, pad, dtype):
    N, Cin, H, W = xs
    dxp = np.zeros((N, Cin, H + 2 * pad, W + 2 * pad), dtype)
    O.convolution_backward_input(dxp, go.astype(dtype), w.astype(dtype), (2, 2), (1, 1), 1)
    return dxp[:, :, pad:pad + H, pad:pad + W]


@pytest.mark.parametrize("N,Cin,Cout,H,W,pad", SHAPES)
def test_s2dx_equals_the_per_phase_kernels_exactly_on_integer_data(dev, N, Cin, Cout, H, W, pad):
    xs = (N, Cin, H, W)
    Ho, Wo = (H + 2 * pad - 3) // 2 + 1, (W + 2 * pad - 3) // 2 + 1
    go, w, dx0 = ints(1, (N, Cout, Ho, Wo), -3, 3), ints(2, (Cout, Cin, 3, 3), -2, 2), ints(3, xs, -5, 5)
    fused, phases, again = run(dev, go, w, dx0, xs, pad, 1), run(dev, go, w, dx0, xs, pad, 0), run(dev, go, w, dx0, xs, pad, 1)
    want = oracle_dx(go, w, xs, pad, np.float32)
    for a, b, r in zip(fused, phases, again):
        assert np.array_equal(a, b) and np.array_equal(a, r)
    assert np.array_equal(fused[0], dx0 + want) and np.array_equal(fused[1], want)


@pytest.mark.parametrize("N,Cin,Cout,H,W,pad", SHAPES[:8])
def test_s2dx_random_inside_the_contraction_bound(dev, N, Cin, Cout, H, W, pad):
    from tolerance import assert_contraction
    xs = (N, Cin, H, W)
    Ho, Wo = (H + 2 * pad - 3) // 2 + 1, (W + 2 * pad - 3) // 2 + 1
    go, w, dx0 = rnd(1, (N, Cout, Ho, Wo)), rnd(2, (Cout, Cin, 3, 3), -1, 1), rnd(3, xs)
    fused, phases = run(dev, go, w, dx0, xs, pad, 1), run(dev, go, w, dx0, xs, pad, 0)
    d64, d32 = oracle_dx(go, w, xs, pad, np.float64), oracle_dx(go, w, xs, pad, np.float32)
    K = Cout * 4                                            # the longest phase: four taps
    for name, got in (("fused +=", fused[0] - dx0), ("fused =", fused[1]), ("per phase =", phases[1])):
        assert_contraction("s2dx:" + name, got, d64, K, 1.0, 1.0, cpu32=d32, epilogue=True)
    assert not np.array_equal(fused[1], phases[1]) or Cout <= 16      # two orders of summation


def _fuzz(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        out.append((int(rng.integers(1, 7)), int(rng.choice([64, 128, 192, 256])), int(rng.choice([16, 32, 48, 64, 96, 128])),
                    2 * int(rng.integers(1, 13)), 2 * int(rng.integers(1, 13)), int(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("N,Cin,Cout,H,W,pad", _fuzz(32, 20260930))
def test_s2dx_fuzz_equals_the_per_phase_kernels_on_integer_data(dev, N, Cin, Cout, H, W, pad):
    """Seeded random geometries (1 - 6 samples, 64 - 256 input channels, 16 - 128 output channels, even extents 2 - 24, padding 0 / 1): the
    fused-phase kernel (forced, whatever the size; narrow or wide blocks by the channel counts) against the per-phase kernels and the
    oracle on integer-valued data, bit for bit, `+=` and first write."""
    if H + 2 * pad < 3 or W + 2 * pad < 3:
        pytest.skip("no output position")
    xs = (N, Cin, H, W)
    Ho, Wo = (H + 2 * pad - 3) // 2 + 1, (W + 2 * pad - 3) // 2 + 1
    go, w, dx0 = ints(1, (N, Cout, Ho, Wo), -3, 3), ints(2, (Cout, Cin, 3, 3), -2, 2), ints(3, xs, -5, 5)
    fused, phases = run(dev, go, w, dx0, xs, pad, 1), run(dev, go, w, dx0, xs, pad, 0)
    want = oracle_dx(go, w, xs, pad, np.float32)
    for a, b in zip(fused, phases):
        assert np.array_equal(a, b)
    assert np.array_equal(fused[0], dx0 + want) and np.array_equal(fused[1], want)


def test_s2dx_rule_and_what_it_leaves_alone(dev):
    """By rule from one block per CU on; odd extents, other kernels / strides, groups, dilation, 40 channels and mixed padding stay with the
    per-phase (or direct) kernels even under the forced knob: the same bits as with the knob at 0."""
    c = capi()

    def dx_of(xs, ws, stride, pad, groups, dil, mode, seed=1):
        N, Cin = xs[0], xs[1]
        padded = tuple(xs[2 + i] + 2 * pad[i] for i in range(2))
        oshape = O.conv_out_shape((N, Cin) + padded, ws, stride, dil)
        go, w = rnd(seed, oshape), rnd(seed + 1, ws, -1, 1)
        dev.conv_s2dx(mode)
        try:
            D = dev.full(xs, np.nan)
            c.conv_bwd_input(dev, D, dev.array(go), dev.array(w), stride, dil, groups, assign=True, padding=pad if any(pad) else None)
            return D.numpy()
        finally:
            dev.conv_s2dx(None)

    small, large = (2, 64, 8, 8), (64, 64, 56, 56)
    ws = (128, 64, 3, 3)
    assert np.array_equal(dx_of(small, ws, (2, 2), (1, 1), 1, (1, 1), None), dx_of(small, ws, (2, 2), (1, 1), 1, (1, 1), 0))      # rule: per phase
    assert not np.array_equal(dx_of(small, ws, (2, 2), (1, 1), 1, (1, 1), 1), dx_of(small, ws, (2, 2), (1, 1), 1, (1, 1), 0))     # forced: fused
    assert np.array_equal(dx_of(large, ws, (2, 2), (1, 1), 1, (1, 1), None), dx_of(large, ws, (2, 2), (1, 1), 1, (1, 1), 1))      # rule: fused
    for xs, wk, st, pad, g, dil in (((2, 64, 7, 8), ws, (2, 2), (1, 1), 1, (1, 1)), ((2, 64, 8, 8), (128, 64, 5, 5), (2, 2), (2, 2), 1, (1, 1)),
                                    ((2, 64, 8, 8), ws, (2, 1), (1, 1), 1, (1, 1)), ((2, 128, 8, 8), ws, (2, 2), (1, 1), 2, (1, 1)),
                                    ((2, 64, 10, 10), ws, (2, 2), (2, 2), 1, (2, 2)), ((2, 40, 8, 8), (128, 40, 3, 3), (2, 2), (1, 1), 1, (1, 1)),
                                    ((2, 64, 8, 8), ws, (2, 2), (1, 0), 1, (1, 1))):
        assert np.array_equal(dx_of(xs, wk, st, pad, g, dil, 1), dx_of(xs, wk, st, pad, g, dil, 0)), (xs, wk, st, pad, g, dil)


@pytest.mark.parametrize("xs,ws,stride,pad,groups", [((2, 3, 20, 20), (8, 3, 7, 7), (2, 2), (3, 3), 1),     # the stem's geometry in small
                                                     ((3, 4, 9, 11), (6, 2, 3, 3), (2, 3), (1, 0), 2),       # groups, unequal strides
                                                     ((2, 2, 5, 6, 7), (4, 2, 2, 3, 1), (2, 1, 3), (0, 1, 0), 1),   # three dimensions
                                                     ((2, 8, 33), (16, 8, 5), (4,), (2,), 1),               # one dimension, stride 4
                                                     ((1, 3, 224, 224), (64, 3, 7, 7), (2, 2), (3, 3), 1)])  # the stem itself, one sample
def test_strided_direct_input_gradient_walks_residue_classes(dev, xs, ws, stride, pad, groups):
    """The direct (<= 16 channels per group) input gradient of a STRIDED convolution: a lane walks its own residue class of taps
    (`conv_direct_bwd_input_strided_kernel`, round 6: no division per tap - the 7 x 7 / stride-2 stem 3.8 ms -> see the shape table) -
    against the oracle on integer-valued data, exactly, `+=` and first write, with the module's padding folded and without."""
    c = capi()
    nd = len(xs) - 2
    dil = (1,) * nd
    padded = tuple(xs[:2]) + tuple(xs[2 + i] + 2 * pad[i] for i in range(nd))
    oshape = O.conv_out_shape(padded, ws, stride, dil)
    go, w, dx0 = ints(1, oshape, -3, 3), ints(2, ws, -2, 2), ints(3, xs, -5, 5)
    want_p = np.zeros(padded, np.float32)
    O.convolution_backward_input(want_p, go, w, stride, dil, groups)
    inner = (slice(None), slice(None)) + tuple(slice(pad[i], pad[i] + xs[2 + i]) for i in range(nd))
    G, Wd = dev.array(go), dev.array(w)
    D, Da = dev.array(dx0), dev.full(xs, np.nan)
    padding = pad if any(pad) else None
    c.conv_bwd_input(dev, D, G, Wd, stride, dil, groups, padding=padding)
    c.conv_bwd_input(dev, Da, G, Wd, stride, dil, groups, assign=True, padding=padding)
    assert np.array_equal(Da.numpy(), want_p[inner]) and np.array_equal(D.numpy(), dx0 + want_p[inner])
    DP = dev.full(padded, np.nan)                                   # the gradient of the padded input itself
    c.conv_bwd_input(dev, DP, G, Wd, stride, dil, groups, assign=True)
    assert np.array_equal(DP.numpy(), want_p)


# N, Cin, Cout, H, W of the UNPADDED input, padding (both axes)
FWD_SHAPES = [
    (2, 64, 128, 8, 8, 1),        # wide-capable (128 output channels) but few blocks: narrow by rule under the forced knob
    (3, 64, 64, 7, 10, 1),        # odd input extent: 4 x 5 outputs, a partly filled block
    (2, 128, 256, 9, 9, 0),       # no padding: 4 x 4 outputs
    (1, 32, 128, 12, 6, 1),       # one chunk of 32 (wide) / two of 16 (narrow)
    (2, 48, 64, 6, 6, 1),         # 48 input channels: narrow only (three chunks of 16)
    (5, 64, 64, 2, 2, 1),         # one output position per sample
    (4, 64, 128, 56, 56, 1),      # the layer of the shape table, four samples
]


@pytest.mark.parametrize("N,Cin,Cout,H,W,pad", FWD_SHAPES)
def test_s2_forward_tap_planes_equal_the_implicit_gemm_exactly_on_integer_data(dev, N, Cin, Cout, H, W, pad):
    """The stride-2 3 x 3 forward as nine tap products on staged tap planes (csrc/nk_conv_s2fwd.h) against the implicit-GEMM kernel
    (NK_TUNE_CONV_S2DX = 0) and the oracle on integer-valued data, bit for bit, with and without the fused bias, on the padded copy
    (`nk_conv_fwd` / `nk_conv_bias_fwd`) and - padding 1 - with the padding folded (`nk_conv_bias_fwd_padded`); both block shapes."""
    c = capi()
    x, w, b = ints(1, (N, Cin, H, W), -3, 3), ints(2, (Cout, Cin, 3, 3), -2, 2), ints(3, (Cout, 1, 1), -4, 4)
    xp = np.zeros((N, Cin, H + 2 * pad, W + 2 * pad), np.float32)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    oshape = O.conv_out_shape(xp.shape, w.shape, (2, 2), (1, 1))
    want = np.zeros(oshape, np.float32); O.convolution_forward(xp, w, want, (2, 2), (1, 1), 1)
    outs = {}
    for mode in (1, 0, 2, 3):
        dev.conv_s2dx(mode)
        try:
            XP, X, Wd, B = dev.array(xp), dev.array(x), dev.array(w), dev.array(b)
            Y, Yb, Yf = dev.full(oshape, np.nan), dev.full(oshape, np.nan), dev.full(oshape, np.nan)
            c.conv_fwd(dev, XP, Wd, Y, (2, 2), (1, 1), 1)
            c.conv_fwd(dev, XP, Wd, Yb, (2, 2), (1, 1), 1, bias=B)
            c.conv_fwd_padded(dev, X, Wd, Yf, (pad, pad), (2, 2), (1, 1), 1, bias=B)
            outs[mode] = [Y.numpy(), Yb.numpy(), Yf.numpy()]
        finally:
            dev.conv_s2dx(None)
    for mode in (1, 0, 2, 3):
        assert np.array_equal(outs[mode][0], want) and np.array_equal(outs[mode][1], want + b) and np.array_equal(outs[mode][2], want + b), mode


def test_s2_forward_random_inside_the_contraction_bound(dev):
    from tolerance import assert_contraction
    c = capi()
    N, Cin, Cout, H = 3, 128, 128, 20
    x, w = rnd(1, (N, Cin, H, H)), rnd(2, (Cout, Cin, 3, 3), -1, 1)
    oshape = O.conv_out_shape(x.shape, w.shape, (2, 2), (1, 1))
    y64 = np.zeros(oshape); O.convolution_forward(x.astype(np.float64), w.astype(np.float64), y64, (2, 2), (1, 1), 1)
    y32 = np.zeros(oshape, np.float32); O.convolution_forward(x, w, y32, (2, 2), (1, 1), 1)
    got = {}
    for mode in (1, 0):
        dev.conv_s2dx(mode)
        try:
            Y = dev.full(oshape, np.nan)
            c.conv_fwd(dev, dev.array(x), dev.array(w), Y, (2, 2), (1, 1), 1)
            got[mode] = Y.numpy()
        finally:
            dev.conv_s2dx(None)
        assert_contraction("s2 forward (%s)" % ("tap planes" if mode else "implicit GEMM"), got[mode], y64, Cin * 9, 1.0, 1.0, cpu32=y32)
    assert not np.array_equal(got[1], got[0])                        # two orders of summation
