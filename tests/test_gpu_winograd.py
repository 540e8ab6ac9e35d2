"""Winograd F(2x2, 3x3) forward and input gradient (nk_conv_winograd.h) against the direct implicit-GEMM kernels and the
oracle: 3 x 3 / stride 1 / dilation 1 / one group, even output extents, channel counts in whole MFMA blocks.

  * integer-valued data: the transforms are additions and halvings, every product and sum is exact in f32 - the Winograd
    result must EQUAL the direct kernels' bit for bit (and the oracle's), forward, input gradient (`+=` and assign, with and
    without the folded padding), bias;
  * random data: both paths inside the suite's contraction bound (tests/tolerance.py, K = Cin * 9) against the f64 oracle,
    margins recorded under winograd:*;
  * tile counts that do not fill the last block, several chunks of reduction channels, several blocks of output channels,
    image borders (the folded padding reads zeros outside the gradient);
  * the rule (by block count) and the knob NK_TUNE_CONV_WINOGRAD = 0 / 1;
  * run to run identical.

Second half of the file: the kernel gradient as F(3x3, 2x2) (nk_conv_winograd_dw.h) - its transform matrices against the nine direct
sums, dW (+= and =) and db bit-equal to the implicit-GEMM pass and the oracle on integer data over slice counts that leave ragged and
empty items, random data inside the contraction bound (K = N * Ho * Wo)."""
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


def rnd(seed, shape, lo=0.0, hi=1.0):
    a = np.random.default_rng(seed).random(shape, dtype=np.float32)
    return np.asarray(a * np.float32(hi - lo) + np.float32(lo), dtype=np.float32)


def ints(seed, shape, lo, hi):
    return np.random.default_rng(seed).integers(lo, hi + 1, shape).astype(np.float32)


# N, Cin, Cout, H, W of the (already padded) input; output (H - 2) x (W - 2)
SHAPES = [
    (2, 64, 128, 10, 10),      # 2 * 4 * 4 = 32 tiles: exactly one forward block, half an input-gradient block
    (3, 64, 128, 8, 14),       # 3 * 3 * 6 = 54 tiles: a partly filled last block both ways
    (1, 128, 128, 12, 12),     # forward: two chunks of 64 reduction channels; input gradient: 4 chunks, two channel blocks
    (2, 64, 256, 6, 20),       # forward: two blocks of 128 output channels; input gradient: 8 chunks of 32
    (5, 192, 128, 6, 6),       # 5 * 2 * 2 = 20 tiles; three forward chunks, three channel blocks of dX
    (2, 64, 128, 58, 58),      # the C3 plane (28 x 28 tiles per image), two samples
    (2, 64, 64, 10, 10),       # 64 output channels either way: the narrow blocks (two waves, chunks of 16) in both passes
    (2, 48, 64, 8, 12),        # forward: three chunks of 16 (narrow only: 48 is no multiple of 32); input gradient: 48 channels - direct
    (1, 32, 128, 12, 12),      # forward: ONE chunk of 32 (every item is a tile block's first and last); input gradient: 32 channels - direct
    (2, 128, 128, 10, 10),     # both block shapes possible in both passes (the third knob value picks)
    # round 6, the ODD instantiations: output extents that are not both even (border tiles with one row / one column)
    (2, 64, 128, 9, 9),        # 7 x 7 outputs (4 x 4 tiles per image, the last row and column half empty); dX of the 9 x 9 / 7 x 7 input alike
    (3, 64, 64, 9, 15),        # 7 x 13: narrow blocks both ways
    (1, 128, 128, 11, 8),      # 9 x 6: only the rows are odd
    (2, 64, 128, 8, 13),       # 6 x 11: only the columns are odd
    (4, 128, 256, 15, 15),     # 13 x 13, several tile blocks, two blocks of output channels
    (130, 64, 64, 5, 5),       # 3 x 3 outputs: 4 tiles per image, three of them border tiles
]


def run_all(dev, x, w, b, go, dx0, pad, mode, shape=None):
    c = capi()
    dev.conv_winograd(mode, None, shape)
    try:
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        X, Wd, B, G = dev.array(x), dev.array(w), dev.array(b), dev.array(go)
        Y, Yb = dev.full((N, Cout, H - 2, W - 2), 7.0), dev.full((N, Cout, H - 2, W - 2), 7.0)
        c.conv_fwd(dev, X, Wd, Y, (1, 1), (1, 1), 1)
        c.conv_fwd(dev, X, Wd, Yb, (1, 1), (1, 1), 1, bias=B)
        DX, DXa = dev.array(dx0), dev.full(x.shape, np.nan)
        c.conv_bwd_input(dev, DX, G, Wd, (1, 1), (1, 1), 1)
        c.conv_bwd_input(dev, DXa, G, Wd, (1, 1), (1, 1), 1, assign=True)
        up = (N, Cin, H - 2 * pad, W - 2 * pad)                     # gradient of the UNPADDED input of a zero Pad node
        DXp = dev.array(dx0[:, :, pad:H - pad, pad:W - pad].copy())
        DXpa = dev.full(up, np.nan)
        c.conv_bwd_input(dev, DXp, G, Wd, (1, 1), (1, 1), 1, padding=(pad, pad))
        c.conv_bwd_input(dev, DXpa, G, Wd, (1, 1), (1, 1), 1, assign=True, padding=(pad, pad))
        return [a.numpy() for a in (Y, Yb, DX, DXa, DXp, DXpa)]
    finally:
        dev.conv_winograd(None)


@pytest.mark.parametrize("N,Cin,Cout,H,W", SHAPES)
def test_winograd_equals_direct_exactly_on_integer_data(dev, N, Cin, Cout, H, W):
    x, w = ints(1, (N, Cin, H, W), -3, 3), ints(2, (Cout, Cin, 3, 3), -2, 2)
    b, go, dx0 = ints(3, (Cout, 1, 1), -4, 4), ints(4, (N, Cout, H - 2, W - 2), -3, 3), ints(5, (N, Cin, H, W), -5, 5)
    wino = run_all(dev, x, w, b, go, dx0, 1, 1)
    direct = run_all(dev, x, w, b, go, dx0, 1, 0)
    again = run_all(dev, x, w, b, go, dx0, 1, 1)
    narrow, wide = run_all(dev, x, w, b, go, dx0, 1, 1, 0), run_all(dev, x, w, b, go, dx0, 1, 1, 1)   # the two block shapes
    for name, a, d, r, nr, wd in zip(("y", "y+bias", "dx+=", "dx=", "dx(pad)+=", "dx(pad)="), wino, direct, again, narrow, wide):
        assert np.array_equal(a, d), name
        assert np.array_equal(a, r), name
        assert np.array_equal(a, nr) and np.array_equal(a, wd), name
    y = np.zeros((N, Cout, H - 2, W - 2), np.float32); O.convolution_forward(x, w, y, (1, 1), (1, 1), 1)
    assert np.array_equal(wino[0], y) and np.array_equal(wino[1], y + b)
    dx = dx0.copy(); O.convolution_backward_input(dx, go, w, (1, 1), (1, 1), 1)
    assert np.array_equal(wino[2], dx) and np.array_equal(wino[3], dx - dx0)
    assert np.array_equal(wino[4], dx[:, :, 1:H - 1, 1:W - 1]) and np.array_equal(wino[5], (dx - dx0)[:, :, 1:H - 1, 1:W - 1])


@pytest.mark.parametrize("N,Cin,Cout,H,W", SHAPES[:5] + SHAPES[6:15])
@pytest.mark.parametrize("pad", [0, 1, 2])
def test_winograd_random_inside_the_contraction_bound(dev, N, Cin, Cout, H, W, pad):
    from tolerance import assert_contraction
    if H - 2 * pad < 2 or W - 2 * pad < 2:
        pytest.skip("nothing left of the unpadded input")
    x, w = rnd(1, (N, Cin, H, W)), rnd(2, (Cout, Cin, 3, 3), -1, 1)
    b, go, dx0 = rnd(3, (Cout, 1, 1), -1, 1), rnd(4, (N, Cout, H - 2, W - 2)), rnd(5, (N, Cin, H, W))
    wino = run_all(dev, x, w, b, go, dx0, pad, 1)
    direct = run_all(dev, x, w, b, go, dx0, pad, 0)
    y64 = np.zeros((N, Cout, H - 2, W - 2)); O.convolution_forward(x.astype(np.float64), w.astype(np.float64), y64, (1, 1), (1, 1), 1)
    y32 = np.zeros((N, Cout, H - 2, W - 2), np.float32); O.convolution_forward(x, w, y32, (1, 1), (1, 1), 1)
    d64 = np.zeros(x.shape); O.convolution_backward_input(d64, go.astype(np.float64), w.astype(np.float64), (1, 1), (1, 1), 1)
    d32 = np.zeros(x.shape, np.float32); O.convolution_backward_input(d32, go, w, (1, 1), (1, 1), 1)
    inner = (slice(None), slice(None), slice(pad, H - pad), slice(pad, W - pad))
    wants = [(y64, y32, Cin * 9), (y64 + b, y32 + b, Cin * 9), (dx0 + d64, dx0 + d32, Cout * 9), (d64, d32, Cout * 9),
             (dx0[inner] + d64[inner], dx0[inner] + d32[inner], Cout * 9), (d64[inner], d32[inner], Cout * 9)]
    for name, got_w, got_d, (ref, cpu, K) in zip(("y", "y+bias", "dx+=", "dx=", "dx(pad)+=", "dx(pad)="), wino, direct, wants):
        assert_contraction("winograd:" + name, got_w, ref, K, 1.0, 1.0, cpu32=cpu, epilogue=True)
        assert_contraction("direct (same cases):" + name, got_d, ref, K, 1.0, 1.0, cpu32=cpu, epilogue=True)


def test_winograd_rule_and_knob(dev):
    """By rule the path is taken from an eighth of the CUs' worth of wide blocks on (an 18 x 18 plane, one sample: 2 forward blocks -
    direct; the C3 plane with 8 samples: 196 - Winograd); shapes it cannot take (stride 2, 5 x 5, groups, 40 channels) stay direct
    under the forced knob, an odd output extent (7 x 8) is taken since round 6.  Told apart by the bits on random data (the two orders of summation differ)."""
    c = capi()

    def fwd(x, w, s, g, mode):
        dev.conv_winograd(mode)
        try:
            X, Wd = dev.array(x), dev.array(w)
            oshape = O.conv_out_shape(x.shape, w.shape, s, (1, 1))
            Y = dev.zeros(oshape)
            c.conv_fwd(dev, X, Wd, Y, s, (1, 1), g)
            return Y.numpy()
        finally:
            dev.conv_winograd(None)

    w = rnd(2, (128, 64, 3, 3), -1, 1)
    small, large = rnd(1, (1, 64, 18, 18)), rnd(1, (8, 64, 58, 58))
    assert np.array_equal(fwd(small, w, (1, 1), 1, None), fwd(small, w, (1, 1), 1, 0))          # rule: direct
    assert not np.array_equal(fwd(small, w, (1, 1), 1, 1), fwd(small, w, (1, 1), 1, 0))         # forced: Winograd
    assert np.array_equal(fwd(large, w, (1, 1), 1, None), fwd(large, w, (1, 1), 1, 1))          # rule: Winograd
    odd = rnd(1, (2, 64, 9, 10))
    assert not np.array_equal(fwd(odd, w, (1, 1), 1, 1), fwd(odd, w, (1, 1), 1, 0))
    for x, wk, s, g in ((rnd(1, (2, 64, 11, 11)), w, (2, 2), 1),
                        (rnd(1, (2, 64, 12, 12)), rnd(2, (128, 64, 5, 5), -1, 1), (1, 1), 1),
                        (rnd(1, (2, 128, 10, 10)), rnd(2, (128, 64, 3, 3), -1, 1), (1, 1), 2),
                        (rnd(1, (2, 40, 10, 10)), rnd(2, (128, 40, 3, 3), -1, 1), (1, 1), 1)):
        assert np.array_equal(fwd(x, wk, s, g, 1), fwd(x, wk, s, g, 0))


# ---- the kernel gradient, F(3x3, 2x2) (nk_conv_winograd_dw.h) -------------------------------------------------------------------------
# N, Cin, Cout, H, W of the (already padded) input
DW_SHAPES = [
    (2, 64, 64, 10, 10),       # 32 tiles: 4 items, one (co, ci) block pair
    (3, 64, 128, 8, 14),       # 54 tiles (a partly filled last item), two co blocks
    (1, 128, 64, 12, 12),      # two ci blocks (the bias gradient is reported by the first only)
    (2, 128, 128, 6, 20),      # four block pairs
    (5, 64, 64, 6, 6),         # 20 tiles
    (6, 64, 128, 58, 58),      # the C3 plane: 4704 tiles over 128 slices (ragged slices, an all-empty item at the end of each)
    # round 6, the ODD instantiations: border tiles with one dY row / column
    (2, 64, 64, 9, 9),         # 7 x 7 outputs
    (3, 64, 128, 9, 15),       # 7 x 13
    (1, 128, 64, 11, 8),       # 9 x 6: only the rows are odd
    (2, 64, 128, 8, 13),       # 6 x 11: only the columns are odd
    (5, 128, 128, 15, 15),     # 13 x 13
    (40, 64, 64, 5, 5),        # 3 x 3 outputs: 4 tiles per image, three of them border tiles
]


def run_dw(dev, x, go, dw0, db0, mode):
    c = capi()
    dev.conv_winograd(1 if mode else 0, None, None, mode)
    try:
        X, G = dev.array(x), dev.array(go)
        DW, DWa, DWb, DB = dev.array(dw0), dev.full(dw0.shape, np.nan), dev.array(dw0), dev.array(db0)
        DWc, DBc = dev.full(dw0.shape, np.nan), dev.full(db0.shape, np.nan)
        c.conv_bwd_kernel(dev, DW, G, X, (1, 1), (1, 1), 1)
        c.conv_bwd_kernel(dev, DWa, G, X, (1, 1), (1, 1), 1, assign=True)
        c.conv_bwd_kernel_bias(dev, DWb, DB, G, X, (1, 1), (1, 1), 1)
        c.conv_bwd_kernel_bias(dev, DWc, DBc, G, X, (1, 1), (1, 1), 1, assign=(True, True))
        return [a.numpy() for a in (DW, DWa, DWb, DB, DWc, DBc)]
    finally:
        dev.conv_winograd(None)


def test_kernel_gradient_transform_matrices():
    """F(3x3, 2x2): A^T [(G g G^T) . (B^T d B)] A equals the nine direct sums for a 2x2 tile of dY on a 4x4 patch (the matrices of
    nk_conv_winograd_dw.h, restated here)."""
    AT = np.array([[1, 1, 1, 0], [0, 1, -1, 0], [0, 1, 1, 1]], float)
    G = np.array([[1, 0], [.5, .5], [.5, -.5], [0, 1]], float)
    BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, -1, 0, 1]], float)
    rng = np.random.default_rng(0)
    g, d = rng.integers(-4, 5, (2, 2)).astype(float), rng.integers(-4, 5, (4, 4)).astype(float)
    want = np.array([[sum(g[a, b] * d[a + k, b + l] for a in range(2) for b in range(2)) for l in range(3)] for k in range(3)])
    assert np.array_equal(AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T, want)


@pytest.mark.parametrize("N,Cin,Cout,H,W", DW_SHAPES)
def test_winograd_kernel_gradient_equals_direct_exactly_on_integer_data(dev, N, Cin, Cout, H, W):
    x, go = ints(1, (N, Cin, H, W), -3, 3), ints(4, (N, Cout, H - 2, W - 2), -2, 2)
    dw0, db0 = ints(5, (Cout, Cin, 3, 3), -5, 5), ints(6, (Cout, 1, 1), -5, 5)
    wino, direct, again = run_dw(dev, x, go, dw0, db0, 1), run_dw(dev, x, go, dw0, db0, 0), run_dw(dev, x, go, dw0, db0, 1)
    for name, a, d, r in zip(("dw+=", "dw=", "dw+= (with db)", "db+=", "dw= (with db)", "db="), wino, direct, again):
        assert np.array_equal(a, d), name
        assert np.array_equal(a, r), name
    dw = dw0.copy(); O.convolution_backward_kernel(dw, go, x, (1, 1), (1, 1), 1)
    assert np.array_equal(wino[0], dw) and np.array_equal(wino[1], dw - dw0)
    assert np.array_equal(wino[3], db0 + go.sum(axis=(0, 2, 3)).reshape(db0.shape))


@pytest.mark.parametrize("N,Cin,Cout,H,W", DW_SHAPES[:5])
def test_winograd_kernel_gradient_random_inside_the_contraction_bound(dev, N, Cin, Cout, H, W):
    from tolerance import assert_contraction
    x, go = rnd(1, (N, Cin, H, W)), rnd(4, (N, Cout, H - 2, W - 2))
    dw0, db0 = rnd(5, (Cout, Cin, 3, 3)), rnd(6, (Cout, 1, 1))
    wino, direct = run_dw(dev, x, go, dw0, db0, 1), run_dw(dev, x, go, dw0, db0, 0)
    d64 = np.zeros(dw0.shape); O.convolution_backward_kernel(d64, go.astype(np.float64), x.astype(np.float64), (1, 1), (1, 1), 1)
    d32 = np.zeros(dw0.shape, np.float32); O.convolution_backward_kernel(d32, go, x, (1, 1), (1, 1), 1)
    K = N * (H - 2) * (W - 2)
    for name, w, d, ref, cpu in (("dw+=", wino[0], direct[0], dw0 + d64, dw0 + d32), ("dw=", wino[1], direct[1], d64, d32)):
        assert_contraction("winograd:" + name, w, ref, K, 1.0, 1.0, cpu32=cpu)
        assert_contraction("direct (same cases):" + name, d, ref, K, 1.0, 1.0, cpu32=cpu)
    b64 = go.astype(np.float64).sum(axis=(0, 2, 3)).reshape(db0.shape)
    np.testing.assert_allclose(wino[5], b64, rtol=1e-5, atol=1e-6 * K)


# ---- the Conv module's Zero padding folded into the forward and the kernel gradient ---------------------------------------------------
# N, Cin, Cout, H, W of the UNPADDED input, padding per axis
FOLD_SHAPES = [
    (2, 64, 64, 8, 8, (1, 1)),        # every tile touches a border or a corner; the first tile of sample 0 / channel 0 is the shifted load
    (3, 64, 128, 6, 12, (1, 1)),      # two co blocks (forward: the wide blocks), 54 tiles
    (1, 128, 64, 10, 10, (1, 1)),     # two ci blocks
    (2, 64, 64, 2, 2, (1, 1)),        # ONE tile per sample: left and right, top and bottom border at once
    (2, 64, 128, 8, 10, (1, 0)),      # rows only (output 8 x 8)
    (2, 64, 64, 10, 8, (0, 1)),       # columns only
    (4, 64, 128, 56, 56, (1, 1)),     # the C3 plane
    # round 6: odd output extents (ODD + FOLD: with padding 1 the last tile's patch columns 2 AND 3 / rows 2 and 3 lie beyond the image)
    (2, 64, 64, 7, 7, (1, 1)),
    (3, 64, 128, 7, 13, (1, 1)),
    (2, 64, 128, 7, 10, (1, 0)),      # output 7 x 8
    (2, 64, 64, 9, 7, (0, 1)),        # output 7 x 7
    (4, 128, 128, 13, 13, (1, 1)),
    (6, 64, 64, 3, 3, (1, 1)),        # 3 x 3 outputs
]


@pytest.mark.parametrize("N,Cin,Cout,H,W,pad", FOLD_SHAPES)
def test_folded_padding_equals_the_padded_copy_bit_for_bit(dev, N, Cin, Cout, H, W, pad):
    """nk_conv_bias_fwd_padded / nk_conv_bwd_kernel_bias_padded on the UNPADDED input against nk_conv_bias_fwd /
    nk_conv_bwd_kernel_bias on the zero-padded copy (both through the Winograd kernels): the same arithmetic on zeros that are read
    instead of stored - equal bit for bit on RANDOM data, `+=` and `=`, with and without the bias terms."""
    c = capi()
    x, w, b = rnd(1, (N, Cin, H, W), -1, 1), rnd(2, (Cout, Cin, 3, 3), -1, 1), rnd(3, (Cout, 1, 1), -1, 1)
    Ho, Wo = H + 2 * pad[0] - 2, W + 2 * pad[1] - 2
    go, dw0, db0 = rnd(4, (N, Cout, Ho, Wo), -1, 1), rnd(5, (Cout, Cin, 3, 3)), rnd(6, (Cout, 1, 1))
    xp = np.zeros((N, Cin, H + 2 * pad[0], W + 2 * pad[1]), np.float32)
    xp[:, :, pad[0]:pad[0] + H, pad[1]:pad[1] + W] = x
    dev.conv_winograd(1, None, None, 1)
    try:
        X, XP, Wd, B, G = dev.array(x), dev.array(xp), dev.array(w), dev.array(b), dev.array(go)
        got, want = [], []
        for bias in (None, B):
            Y1, Y2 = dev.full((N, Cout, Ho, Wo), np.nan), dev.full((N, Cout, Ho, Wo), np.nan)
            c.conv_fwd_padded(dev, X, Wd, Y1, pad, (1, 1), (1, 1), 1, bias=bias)
            c.conv_fwd(dev, XP, Wd, Y2, (1, 1), (1, 1), 1, bias=bias)
            got.append(Y1.numpy()); want.append(Y2.numpy())
        for assign in (False, True):
            D1, D2 = dev.array(dw0), dev.array(dw0)
            E1, E2 = dev.array(db0), dev.array(db0)
            c.conv_bwd_kernel_padded(dev, D1, G, X, pad, (1, 1), (1, 1), 1, db=E1, assign=(assign, assign))
            c.conv_bwd_kernel_bias(dev, D2, E2, G, XP, (1, 1), (1, 1), 1, assign=(assign, assign))
            got += [D1.numpy(), E1.numpy()]; want += [D2.numpy(), E2.numpy()]
        D3 = dev.array(dw0)
        c.conv_bwd_kernel_padded(dev, D3, G, X, pad, (1, 1), (1, 1), 1)     # no bias gradient
        got.append(D3.numpy()); want.append(want[2])
    finally:
        dev.conv_winograd(None)
    for i, (a, r) in enumerate(zip(got, want)):
        assert np.array_equal(a, r), i
    y = np.zeros((N, Cout, Ho, Wo), np.float32); O.convolution_forward(xp, w, y, (1, 1), (1, 1), 1)
    np.testing.assert_allclose(got[0], y, rtol=0, atol=1e-6 * Cin * 9 * 2)


def test_padding_folds_query_and_fallbacks(dev):
    """nk_conv_padding_folds follows the rules in force (block counts, the knob).  The `_padded` entries fold where their kernels can and
    are otherwise the two nodes they stand for - Pad::forward (node/pad/mod.rs:97-129) into a scratch region of the handle, then the
    convolution entry (convolution/mod.rs:331-355) - for what no kernel folds (stride 2, 5 x 5, groups, padding 2, 40 channels, no padding
    at all, one spatial dimension) and for a foldable geometry under the knob's "never": the bits of pad -> nk_conv_bias_fwd /
    nk_conv_bwd_kernel_bias, never NK_ERR_UNSUPPORTED (round 5 refused; a knob change between graph build and forward() panicked)."""
    c = capi()
    big, small = (48, 64, 56, 56), (2, 64, 8, 8)
    assert c.conv_padding_folds(dev, big, (1, 1), (128, 64, 3, 3), (1, 1), (1, 1))
    assert not c.conv_padding_folds(dev, small, (1, 1), (128, 64, 3, 3), (1, 1), (1, 1))          # below the block-count rules
    dev.conv_winograd(1, None, None, 1)
    try:
        assert c.conv_padding_folds(dev, small, (1, 1), (128, 64, 3, 3), (1, 1), (1, 1))
    finally:
        dev.conv_winograd(None)
    dev.conv_winograd(0)
    try:
        assert not c.conv_padding_folds(dev, big, (1, 1), (128, 64, 3, 3), (1, 1), (1, 1))
    finally:
        dev.conv_winograd(None)
    cases = [(small, (1, 1), (128, 64, 3, 3), (2, 2), 1, None), (small, (2, 2), (128, 64, 5, 5), (1, 1), 1, None),
             ((2, 128, 8, 8), (1, 1), (128, 64, 3, 3), (1, 1), 2, None), (small, (2, 2), (128, 64, 3, 3), (1, 1), 1, None),
             ((2, 40, 8, 8), (1, 1), (128, 40, 3, 3), (1, 1), 1, None), (small, (0, 0), (128, 64, 3, 3), (1, 1), 1, None),
             ((3, 8, 20), (2,), (16, 8, 5), (1,), 1, None), ((2, 4, 6, 7, 8), (1, 0, 2), (6, 4, 3, 1, 2), (1, 1, 1), 1, None),
             ((4, 64, 16, 16), (1, 1), (128, 64, 3, 3), (1, 1), 1, 0)]               # foldable, under the knob's "never"
    for xs, pad, ws, st, g, knob in cases:
        nd = len(xs) - 2
        dil = (1,) * nd
        if knob is None:
            assert not c.conv_padding_folds(dev, xs, pad, ws, st, dil, g)
        x, w, b = rnd(1, xs, -1, 1), rnd(2, ws, -1, 1), rnd(3, (ws[0],) + (1,) * nd, -1, 1)
        pshape = tuple(xs[:2]) + tuple(xs[2 + i] + 2 * pad[i] for i in range(nd))
        xp = np.zeros(pshape, np.float32)
        xp[(slice(None), slice(None)) + tuple(slice(pad[i], pad[i] + xs[2 + i]) for i in range(nd))] = x
        oshape = O.conv_out_shape(pshape, ws, st, dil)
        go, dw0, db0 = rnd(4, oshape, -1, 1), rnd(5, ws), rnd(6, b.shape)
        X, XP, Wd, B, G = dev.array(x), dev.array(xp), dev.array(w), dev.array(b), dev.array(go)
        if knob is not None:
            dev.conv_winograd(knob)
        try:
            before = dev.conv_winograd_launches()
            for bias in (None, B):
                Y1, Y2 = dev.full(oshape, np.nan), dev.full(oshape, np.nan)
                c.conv_fwd_padded(dev, X, Wd, Y1, pad, st, dil, g, bias=bias)
                c.conv_fwd(dev, XP, Wd, Y2, st, dil, g, bias=bias)
                assert np.array_equal(Y1.numpy(), Y2.numpy()), (xs, pad)
                if bias is None:
                    plain = Y1.numpy()
            for assign in (False, True):
                D1, D2, E1, E2 = dev.array(dw0), dev.array(dw0), dev.array(db0), dev.array(db0)
                c.conv_bwd_kernel_padded(dev, D1, G, X, pad, st, dil, g, db=E1, assign=(assign, assign))
                c.conv_bwd_kernel_bias(dev, D2, E2, G, XP, st, dil, g, assign=(assign, assign))
                assert np.array_equal(D1.numpy(), D2.numpy()) and np.array_equal(E1.numpy(), E2.numpy()), (xs, pad, assign)
            assert dev.conv_winograd_launches() == before
        finally:
            if knob is not None:
                dev.conv_winograd(None)
        y = np.zeros(oshape, np.float32); O.convolution_forward(xp, w, y, st, dil, g)
        from tolerance import assert_contraction
        y64 = np.zeros(oshape, np.float64); O.convolution_forward(xp.astype(np.float64), w.astype(np.float64), y64, st, dil, g)
        assert_contraction("padded entry, fallback", plain, y64, int(np.prod(ws[1:])), cpu32=y)


def test_winograd_non_finite_input_stays_inside_the_tiles_that_see_it(dev):
    """Non-finite inputs: the transforms add and subtract patch elements, so a +inf in the input reaches the outputs
    as +-inf or - where two infinite terms meet in an add tree - as NaN, where the direct kernels give +-inf.  Pinned here: the SET of
    non-finite outputs is a superset of the direct form's and stays inside the 2 x 2 tiles whose 4 x 4 patch contains the element (in
    this case it is the same 3 x 3 window), every other output is untouched, and NK_TUNE_CONV_WINOGRAD = 0 restores the direct values."""
    c = capi()
    N, Cin, Cout, H = 2, 64, 128, 12
    x, w = rnd(1, (N, Cin, H, H), -1, 1), rnd(2, (Cout, Cin, 3, 3), -1, 1)
    x[1, 5, 6, 7] = np.inf                                   # padded-input coordinates (the caller's tensor)
    out = {}
    for mode in (1, 0):
        dev.conv_winograd(mode)
        try:
            X, Wd, Y = dev.array(x), dev.array(w), dev.zeros((N, Cout, H - 2, H - 2))
            c.conv_fwd(dev, X, Wd, Y, (1, 1), (1, 1), 1)
            out[mode] = Y.numpy()
        finally:
            dev.conv_winograd(None)
    direct_bad = ~np.isfinite(out[0])
    want_direct = np.zeros_like(direct_bad); want_direct[1, :, 4:7, 5:8] = True       # outputs whose 3 x 3 window covers (6, 7)
    assert np.array_equal(direct_bad, want_direct)
    wino_bad = ~np.isfinite(out[1])
    tiles = np.zeros_like(wino_bad); tiles[1, :, 4:8, 4:8] = True                     # tiles (2, 2), (2, 3), (3, 2), (3, 3): patches rows / cols 4..9
    assert np.all(wino_bad[direct_bad]) and not np.any(wino_bad & ~tiles)             # a superset of the direct pattern, confined to those tiles
    print("non-finite outputs per channel: direct", int(direct_bad[1, 0].sum()), "winograd", int(wino_bad[1, 0].sum()), "of the tiles' 16")
    np.testing.assert_allclose(out[1][~tiles], out[0][~tiles], rtol=0, atol=1e-6 * Cin * 9 * 2)
    assert np.all(np.isfinite(out[1][0]))                               # the other sample is untouched


def _fuzz_geometries(n, seed, any_extent=False):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        N = int(rng.integers(1, 6))
        Cin, Cout = int(rng.choice([64, 128, 192])), int(rng.choice([64, 128, 192, 256]))
        Ho, Wo = 2 * int(rng.integers(1, 13)), 2 * int(rng.integers(1, 13))
        if any_extent:      # round 6: odd extents too (the ODD instantiations of the forward / input-gradient kernel; 1 is below its minimum)
            Ho, Wo = int(rng.integers(1, 26)), int(rng.integers(1, 26))
        pad = (int(rng.integers(0, 2)), int(rng.integers(0, 2)))
        if Ho + 2 - 2 * pad[0] < 1 or Wo + 2 - 2 * pad[1] < 1:
            continue
        out.append((N, Cin, Cout, Ho, Wo, pad))
    return out


@pytest.mark.parametrize("N,Cin,Cout,Ho,Wo,pad", _fuzz_geometries(36, 20260925) + _fuzz_geometries(40, 20260930, any_extent=True))
def test_winograd_fuzz_all_three_passes_equal_the_direct_kernels_on_integer_data(dev, N, Cin, Cout, Ho, Wo, pad):
    """Seeded random geometries (1 - 5 samples, 64 - 256 channels, even output extents 2 - 24 and - the second set - any extent 1 - 25,
    zero padding 0 / 1 per axis): forward + bias, input gradient with the padding folded, kernel gradient + bias gradient - the Winograd
    kernels (forced, whatever the size) against the
    implicit-GEMM / direct kernels on integer-valued data, bit for bit; with padding also the folded entry points against the padded copy."""
    c = capi()
    H, W = Ho + 2 - 2 * pad[0], Wo + 2 - 2 * pad[1]                    # unpadded input
    x, w, b = ints(1, (N, Cin, H, W), -3, 3), ints(2, (Cout, Cin, 3, 3), -2, 2), ints(3, (Cout, 1, 1), -4, 4)
    go = ints(4, (N, Cout, Ho, Wo), -2, 2)
    xp = np.zeros((N, Cin, H + 2 * pad[0], W + 2 * pad[1]), np.float32)
    xp[:, :, pad[0]:pad[0] + H, pad[1]:pad[1] + W] = x
    res = {}
    for mode in (1, 0):
        dev.conv_winograd(mode, None, None, mode)
        try:
            XP, Wd, B, G = dev.array(xp), dev.array(w), dev.array(b), dev.array(go)
            Y, DX = dev.full((N, Cout, Ho, Wo), np.nan), dev.full((N, Cin, H, W), np.nan)
            DW, DB = dev.full(w.shape, np.nan), dev.full(b.shape, np.nan)
            c.conv_fwd(dev, XP, Wd, Y, (1, 1), (1, 1), 1, bias=B)
            c.conv_bwd_input(dev, DX, G, Wd, (1, 1), (1, 1), 1, assign=True, padding=pad)
            c.conv_bwd_kernel_bias(dev, DW, DB, G, XP, (1, 1), (1, 1), 1, assign=(True, True))
            res[mode] = [a.numpy() for a in (Y, DX, DW, DB)]
            if mode == 1 and pad != (0, 0):
                X = dev.array(x)
                Y2, DW2, DB2 = dev.full((N, Cout, Ho, Wo), np.nan), dev.full(w.shape, np.nan), dev.full(b.shape, np.nan)
                c.conv_fwd_padded(dev, X, Wd, Y2, pad, (1, 1), (1, 1), 1, bias=B)
                c.conv_bwd_kernel_padded(dev, DW2, G, X, pad, (1, 1), (1, 1), 1, db=DB2, assign=(True, True))
                res["fold"] = [a.numpy() for a in (Y2, DW2, DB2)]
        finally:
            dev.conv_winograd(None)
    for name, a, d in zip(("y", "dx", "dw", "db"), res[1], res[0]):
        assert np.array_equal(a, d), name
    if "fold" in res:
        for name, a, d in zip(("y", "dw", "db"), res["fold"], (res[0][0], res[0][2], res[0][3])):
            assert np.array_equal(a, d), "folded " + name
