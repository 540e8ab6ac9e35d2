"""Parity at BASELINE.json's FULL sizes through size-independent properties and sampled f64
references (the oracle is too slow to replay C3/C4/C5 whole, so each test checks what can be
checked exactly or against a cheap f64 restatement of a slice)."""
import numpy as np
import pytest

from oracle import neuronika_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nk():
    import neuronika_amd
    return neuronika_amd.tape


def rnd(seed, shape, lo=0.0, hi=1.0):
    a = np.random.default_rng(seed).random(shape, dtype=np.float32)
    return np.asarray(a * np.float32(hi - lo) + np.float32(lo), dtype=np.float32).reshape(shape)


def test_C3_conv_full_size(dev):
    """C3: x 128x64x56x56, w 128x64x3x3, zero padding 1.  (1) THE CALLS bench.py TIMES - the module's folded entries on the
    UNPADDED x (`nk_conv_bias_fwd_padded`, `nk_conv_bwd_input_padded_assign`, `nk_conv_bwd_kernel_bias_padded`) - and the
    padded-copy entries beside them: sampled outputs / input gradients / kernel gradients against f64 direct sums through
    `assert_contraction` with the operands' own maxima (max|w| = 1/24), margins recorded as `C3_full_size:*`;
    (2) exact linearity: conv(x, 2w) == 2 conv(x, w) bit for bit (power-of-two scaling commutes with every rounding);
    (3) folded == padded copy bit for bit at the full size (forward, kernel and bias gradient; the input gradients tile differently)."""
    from neuronika_amd import capi as c
    import conv_samples as S
    from tolerance import assert_contraction
    N, Cin, Cout, H = 128, 64, 128, 56
    x = rnd(0, (N, Cin, H, H))
    k = 1 / np.sqrt(Cin * 9)
    w = rnd(1, (Cout, Cin, 3, 3), -k, k)
    b = rnd(4, (Cout, 1, 1), -k, k)
    g = rnd(2, (N, Cout, H, H))
    xmax, wmax, gmax = float(np.abs(x).max()), float(np.abs(w).max()), float(np.abs(g).max())
    KF, KI, KW = Cin * 9, Cout * 9, N * H * H
    X, W, G, Bv = dev.array(x), dev.array(w), dev.array(g), dev.array(b)
    xp = np.zeros((N, Cin, H + 2, H + 2), np.float32); xp[:, :, 1:-1, 1:-1] = x
    rng = np.random.default_rng(3)
    took0 = dev.conv_winograd_launches()

    # ---- (1a) the benchmark's own calls, on the unpadded x --------------------------------------------------------------
    assert c.conv_padding_folds(dev, x.shape, (1, 1), w.shape, (1, 1), (1, 1), 1)
    YF = dev.full((N, Cout, H, H), np.nan)
    c.conv_fwd_padded(dev, X, W, YF, (1, 1), (1, 1), (1, 1), 1, bias=Bv)            # nk_conv_bias_fwd_padded
    yf = YF.numpy()
    idx, r64, r32 = S.forward(xp, w, b, rng, 64)
    assert_contraction("C3_full_size:y (nk_conv_bias_fwd_padded)", yf[idx], r64, KF, xmax, wmax, cpu32=r32, epilogue=True)
    DXF = dev.full(x.shape, np.nan)
    c.conv_bwd_input(dev, DXF, G, W, (1, 1), (1, 1), 1, assign=True, padding=(1, 1))   # nk_conv_bwd_input_padded_assign
    dxf = DXF.numpy()
    idx, r64, r32 = S.input_gradient(g, w, rng, 64, padded=False)
    assert_contraction("C3_full_size:dx (nk_conv_bwd_input_padded_assign)", dxf[idx], r64, KI, gmax, wmax, cpu32=r32)
    DWF, DBF = dev.full(w.shape, np.nan), dev.full(b.shape, np.nan)
    c.conv_bwd_kernel_padded(dev, DWF, G, X, (1, 1), (1, 1), (1, 1), 1, db=DBF, assign=(True, True))   # nk_conv_bwd_kernel_bias_padded
    dwf, dbf = DWF.numpy(), DBF.numpy()
    idx, r64, r32 = S.kernel_gradient(g, xp, rng, 16)
    assert_contraction("C3_full_size:dw (nk_conv_bwd_kernel_bias_padded)", dwf[idx], r64, KW, gmax, xmax, cpu32=r32)
    assert dev.conv_winograd_launches() - took0 == 3                  # by rule these ARE the Winograd kernels bench.py times
    # the same entries in their `+=` forms onto a non-zero start (the tape's later backward nodes)
    dx0, dw0, db0 = rnd(7, x.shape, -1, 1), rnd(8, w.shape, -1, 1), rnd(9, b.shape, -1, 1)
    DX2 = dev.array(dx0)
    c.conv_bwd_input(dev, DX2, G, W, (1, 1), (1, 1), 1, padding=(1, 1))
    np.testing.assert_allclose(DX2.numpy() - dx0, dxf, rtol=0, atol=2e-7 * (1 + np.abs(dxf).max()))   # one f32 add onto |dx0| <= 1
    DW2, DB2 = dev.array(dw0), dev.array(db0)
    c.conv_bwd_kernel_padded(dev, DW2, G, X, (1, 1), (1, 1), (1, 1), 1, db=DB2)
    np.testing.assert_allclose(DW2.numpy() - dw0, dwf, rtol=0, atol=2e-7 * (1 + np.abs(dwf).max()))
    DX4 = dev.full(x.shape, np.nan)
    c.conv_bwd_input(dev, DX4, G, W, (1, 1), (1, 1), 1, assign=True, padding=(1, 1))
    assert np.array_equal(dxf, DX4.numpy())                           # run-to-run deterministic

    # ---- (1b) the two-node form: Pad node, then the convolution entries on the padded copy ------------------------------
    XP, Y = dev.zeros((N, Cin, H + 2, H + 2)), dev.zeros((N, Cout, H, H))
    c.pad_const_fwd(dev, X, XP, (1, 1), 0.0)
    c.conv_fwd(dev, XP, W, Y, (1, 1), (1, 1), 1)
    y = Y.numpy()
    idx, r64, r32 = S.forward(xp, w, None, rng, 64)
    assert_contraction("C3_full_size:y (nk_conv_fwd on the padded copy)", y[idx], r64, KF, xmax, wmax, cpu32=r32)
    W2, Y2 = dev.array(2 * w), dev.zeros(y.shape)
    c.conv_fwd(dev, XP, W2, Y2, (1, 1), (1, 1), 1)
    assert np.array_equal(Y2.numpy(), 2 * y)                          # (2)
    YB = dev.full(y.shape, np.nan)
    c.conv_fwd(dev, XP, W, YB, (1, 1), (1, 1), 1, bias=Bv)            # nk_conv_bias_fwd
    assert np.array_equal(YB.numpy(), y + b.reshape(1, Cout, 1, 1))   # one f32 add per element on top of the same tile sums
    assert np.array_equal(YB.numpy(), yf)                             # (3) folded == padded copy
    del Y2, YB, YF

    DXP, DW = dev.zeros(xp.shape), dev.zeros(w.shape)
    c.conv_bwd_input(dev, DXP, G, W, (1, 1), (1, 1), 1)
    c.conv_bwd_kernel(dev, DW, G, XP, (1, 1), (1, 1), 1)
    dxp, dw = DXP.numpy(), DW.numpy()
    idx, r64, r32 = S.input_gradient(g, w, rng, 48, padded=True)
    assert_contraction("C3_full_size:dx (nk_conv_bwd_input on the padded copy)", dxp[idx], r64, KI, gmax, wmax, cpu32=r32)
    idx, r64, r32 = S.kernel_gradient(g, xp, rng, 12)
    assert_contraction("C3_full_size:dw (nk_conv_bwd_kernel on the padded copy)", dw[idx], r64, KW, gmax, xmax, cpu32=r32)
    DX = dev.zeros(x.shape)
    c.pad_bwd(dev, DX, DXP, (1, 1))
    assert np.array_equal(DX.numpy(), dxp[:, :, 1:-1, 1:-1])
    # (the input gradient of the padded copy walks 29 x 29 tiles whose origins lie one element off the folded form's 28 x 28: the same
    #  products, other add trees - equal to twice the bound each side has just been held to, not bit for bit)
    from tolerance import abs_term
    assert np.abs(dxf - dxp[:, :, 1:-1, 1:-1]).max() <= 2 * abs_term(KI, gmax, wmax)
    assert np.array_equal(dwf, dw)                                    # (3)

    # bias gradient summed on the way by the kernel-gradient pass (`nk_conv_bwd_kernel_bias`): same dW bits, db by its own bound
    DW3, DB3 = dev.array(dw0), dev.array(db0)
    c.conv_bwd_kernel_bias(dev, DW3, DB3, G, XP, (1, 1), (1, 1), 1)
    DWr = dev.array(dw0)
    c.conv_bwd_kernel(dev, DWr, G, XP, (1, 1), (1, 1), 1)
    assert np.array_equal(DW3.numpy(), DWr.numpy())                   # the same pass: kernel gradient bit-identical
    assert np.array_equal(DW3.numpy(), DW2.numpy()) and np.array_equal(DB3.numpy(), DB2.numpy())   # (3) in the `+=` form
    g64 = g.astype(np.float64)
    db64 = g64.sum(axis=(0, 2, 3)).reshape(b.shape)                   # AdditionBackwardRight: un-broadcast sum over N, H, W
    db32 = np.zeros(b.shape, np.float32); O.accumulate(db32, g)
    err_gpu, err_cpu = np.abs(DB2.numpy().astype(np.float64) - (db0 + db64)).max(), np.abs(db32 - db64).max()
    from conftest import record_margin
    db_bound = 1e-6 * (N * H * H) ** 0.5 * np.abs(g).max() + 1e-7 * np.abs(db64).max()   # a sum, not a product: sqrt growth + one ulp class
    record_margin("C3_full_size:db", err_gpu, err_cpu, db_bound)
    assert err_gpu <= max(2 * err_cpu, db_bound), (err_gpu, err_cpu)
    np.testing.assert_allclose(dbf, DB2.numpy() - db0, rtol=0, atol=2e-7 * np.abs(db64).max())


def tag_of(spelling):
    return "C4_full_size" if spelling == "reference_words" else "C4_full_size(node by node)"


@pytest.mark.parametrize("spelling", ["reference_words", "node_by_node"])
def test_C4_mlp_full_size(nk, spelling):
    """C4 on one GPU: Linear(4096,4096)x3 + ReLU, batch 4096, MSE mean, backward(1.0): loss and every weight / bias gradient
    against an f64 restatement (OpenBLAS on the host), next to the f32 restatement measured the same way.

    The graph is written in the reference's words, `lin.forward(x).relu()` (neuronika-nn/src/lib.rs:441-447,
    vardiff.rs:282-288), twice: "reference_words" lets the host mirror's graph-build peephole turn them into the Linear+ReLU
    nodes - THE graph bench.py times - and "node_by_node" switches the peephole off (a ReLU node over each Linear's output).

    Two things separate any two f32 evaluations of this network, and the test keeps them apart:
      * ReLU mask flips - a pre-activation within rounding of 0 lands on different sides in different summation orders, and
        one flipped unit moves gradient entries by O(|g|), hundreds of times the summation error (tools/c4_tolerance_model.py:
        3-6 flips per layer between ANY two f32 orders, err 1e-8 against 1e-10).  The masks are index-like behaviour: the
        device's masks are checked against the f64 pre-activations (a bounded number of flips, only where |z| is rounding
        noise), then (i) imposed on both host restatements, so that what is left is summation order, and (ii) NOT imposed -
        every evaluation with its own masks - with the flips' first-order effect as an explicit allowance;
      * summation order - the device sums the K = 4096 weight-gradient contractions as two chained launches (f32 fma chains of
        2048, nk_gemm.hip GEMM_CHAIN_K) and the fused Linear+ReLU forward / masked input gradient as ONE chain of 4096 (their
        epilogue functions act on the whole sum); the reference's sgemm and OpenBLAS sum in blocks of a few hundred.  The bound
        is the suite's one contraction policy, tests/tolerance.py - the survey's 1e-6 K |a| |b|, no factor."""
    from conftest import record_margin
    from tolerance import assert_contraction, abs_term
    dev = nk.Device(0)
    n = 4096
    x, t = rnd(100, (n, n)), rnd(200, (n, n))
    lins = [nk.nn.Linear(dev, n, n, s) for s in (1, 3, 5)]
    X, T = nk.from_ndarray(dev, x), nk.from_ndarray(dev, t)
    was = nk.nn.set_relu_peephole(spelling == "reference_words")
    try:
        a1 = lins[0].forward(X).relu()
        a2 = lins[1].forward(a1).relu()
        loss = lins[2].forward(a2).mse(T, nk.Reduction.Mean)
    finally:
        nk.nn.set_relu_peephole(was)
    assert loss.history_len() == (4 if spelling == "reference_words" else 6)
    loss.forward(); loss.backward(1.0)
    m1, m2 = a1.data() > 0, a2.data() > 0          # strict `>` on the input == `> 0` on max(z, 0)

    def reference(dt, masks):
        W = [(l.weight.data().astype(dt), l.bias.data().astype(dt)) for l in lins]
        h0, tt = x.astype(dt), t.astype(dt)
        z1 = h0 @ W[0][0].T + W[0][1]; k1 = (z1 > 0) if masks is None else masks[0]; a1 = np.where(k1, z1, 0).astype(dt)
        z2 = a1 @ W[1][0].T + W[1][1]; k2 = (z2 > 0) if masks is None else masks[1]; a2 = np.where(k2, z2, 0).astype(dt)
        z3 = a2 @ W[2][0].T + W[2][1]
        ls = ((z3 - tt) ** 2).mean(dtype=dt)
        g3 = 2 * (z3 - tt) / dt(z3.size)
        p2 = g3 @ W[2][0]; g2 = p2 * k2             # p: the gradient in front of the mask
        p1 = g2 @ W[1][0]; g1 = p1 * k1
        bounds = [np.abs(g).max() * np.abs(a).max() for g, a in ((g1, h0), (g2, a1), (g3, a2))]
        return ls, [(g1.T @ h0, g1.sum(0)), (g2.T @ a1, g2.sum(0)), (g3.T @ a2, g3.sum(0))], bounds, (z1, z2), (p1, p2), (a1, a2)

    l64, g64, ab, z64, _, a64 = reference(np.float64, (m1, m2))
    l32, g32, _, _, _, a32 = reference(np.float32, (m1, m2))
    np.testing.assert_allclose(loss.item(), l64, rtol=2e-6)
    # the forward as the benchmark runs it (Linear+ReLU nodes: bias and ReLU in the GEMM epilogue, one chain of 4096): sampled rows
    # of both hidden activations, the device's masks on every side
    rows = np.random.default_rng(11).integers(0, n, 64)
    for name, act, k, amax in (("a1", a1.data(), 0, float(np.abs(x).max())), ("a2", a2.data(), 1, float(np.abs(a1.data()).max()))):
        assert_contraction(tag_of(spelling) + f":{name} (Linear+ReLU forward, K = 4096, one chain)", act[rows], a64[k][rows], n, amax,
                           float(np.abs(lins[k].weight.data()).max()), cpu32=a32[k][rows], epilogue=True)
    # the device's masks against the f64 pre-activations: they may differ only where |z| is below the rounding error of a
    # K = 4096 f32 contraction (the flips seen are at |z| ~ 1e-7)
    for m, z, amax in ((m1, z64[0], 1.0), (m2, z64[1], float(np.abs(a1.data()).max()))):
        flipped = m != (z > 0)
        assert flipped.sum() <= 64, int(flipped.sum())
        if flipped.any():
            assert np.abs(z[flipped]).max() <= abs_term(n, amax, 1.0 / np.sqrt(n)), float(np.abs(z[flipped]).max())
    tag = tag_of(spelling)
    for lin, (dw64, db64), (dw32, db32), gab in zip(lins, g64, g32, ab):
        # (i) the device's masks on every side: summation order alone.  K = 4096 as two chained launches (chains of 2048)
        err_gpu, err_cpu = np.abs(lin.weight.grad() - dw64).max(), np.abs(dw32 - dw64).max()
        assert_contraction(tag + ":dW (K = 4096, chains of 2048)", lin.weight.grad(), dw64, n, gab, 1.0, cpu32=dw32)
        assert_contraction(tag + ":db", lin.bias.grad(), db64, n, gab, 1.0, cpu32=db32)
    # (ii) every evaluation with its OWN masks (the reference run by itself): the same bound plus the first-order effect of the
    # flipped units - a flip at (sample i, unit u) of layer l switches the gradient entry p_l[i, u] on or off; that moves row u
    # of dW_l by |p_l[i, u]| * |a_(l-1)[i, :]| and, through W_l, the entries of the earlier weight gradient by at most
    # |p_l[i, u]| * max|W_l[u, :]| * max|x|.  (Entries no flip reaches - dW3, db3 - are inside the plain bound.)
    _, g64o, abo, z64o, p64o, a64o = reference(np.float64, None)
    _, g32o, _, _, _, _ = reference(np.float32, None)
    f1, f2 = m1 != (z64o[0] > 0), m2 != (z64o[1] > 0)
    W2abs = float(np.abs(lins[1].weight.data()).max())
    d2 = float(np.abs(p64o[1][f2]).sum()) if f2.any() else 0.0      # sum over flips of |p2[i, u]|
    d1 = float(np.abs(p64o[0][f1]).sum()) if f1.any() else 0.0
    allow = [d1 * 1.0 + d2 * W2abs * 1.0,                            # dW1: own flips (|x| <= 1) + layer-2 flips through one entry of W2
             d2 * float(np.abs(a64o[0]).max()),                     # dW2: rows of the flipped units
             0.0]                                                    # dW3: no mask behind it
    for k, (lin, (dw64, db64), (dw32, _), gab) in enumerate(zip(lins, g64o, g32o, abo)):
        err_gpu, err_cpu = np.abs(lin.weight.grad() - dw64).max(), np.abs(dw32 - dw64).max()
        record_margin(tag + ":dW with every evaluation's own masks (bound + flip allowance)", err_gpu, err_cpu, abs_term(n, gab, 1.0) + allow[k])
        assert err_gpu <= max(2 * err_cpu, abs_term(n, gab, 1.0)) + allow[k], (k, err_gpu, err_cpu, allow[k])


def test_C5_attention_full_size(nk):
    """C5 (d=1024, h=16, S=1024, B=32) with dropout off: the output rows and input gradients of
    ONE sample against an f64 restatement of that sample (attention does not couple samples),
    and every row of the attention probabilities sums to 1."""
    dev = nk.Device(0)
    B, S, d, H = 32, 1024, 1024, 16
    mha = nk.nn.MultiheadAttention(dev, d, H, 0.0, 1)
    x = rnd(0, (B * S, d)); g = rnd(5, (B * S, d))
    X = nk.from_ndarray(dev, x).requires_grad()
    out = mha.forward(X, B)
    loss = (out * nk.from_ndarray(dev, g)).sum()
    loss.forward(); loss.backward(1.0)
    b = 17
    rows = slice(b * S, (b + 1) * S)
    Wp = {n: (getattr(mha, n).weight.data().astype(np.float64), getattr(mha, n).bias.data().astype(np.float64)) for n in "qkvo"}
    ref, grads = O.mha_forward_backward(x[rows].astype(np.float64), *Wp["q"], *Wp["k"], *Wp["v"], *Wp["o"], H, 1, 0.0,
                                        np.ones((H, S, S)), g[rows].astype(np.float64))
    got = out.data()[rows]
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    gx = X.grad()[rows]
    assert np.abs(gx - grads["x"]).max() <= 2e-5 * np.abs(grads["x"]).max()


def _mha_oracle64(mha, x, g, H, B, p, noise):
    dt = np.float64
    W = {n: (getattr(mha, n).weight.data().astype(dt), getattr(mha, n).bias.data().astype(dt)) for n in "qkvo"}
    return O.mha_forward_backward(x.astype(dt), *W["q"], *W["k"], *W["v"], *W["o"], H, B, p, noise, g.astype(dt))


def test_C5_attention_full_size_with_dropout(nk):
    """C5 exactly as benchmarked (d=1024, h=16, S=1024, B=32, dropout 0.1, fused probabilities recomputed in the
    backward pass): output rows and input gradients of ONE sample against the f64 oracle of that sample, fed the
    Philox mask of that sample's slice of the (B*H, S, S) probabilities (key from manual_seed, counter offset
    b*H*S*S/4).  Attention does not couple samples, so one sample pins the activations path at full size; the
    parameter gradients (sums over the batch) are pinned at this geometry by the 4-sample test below."""
    dev = nk.Device(0)
    B, S, d, H, p, seed = 32, 1024, 1024, 16, 0.1, 7
    nk.manual_seed(seed)
    mha = nk.nn.MultiheadAttention(dev, d, H, p, 1)
    x = rnd(0, (B * S, d)); g = rnd(5, (B * S, d))
    X = nk.from_ndarray(dev, x).requires_grad()
    out = mha.forward(X, B)
    out.forward(); out.backward_from(nk.from_ndarray(dev, g))
    b = 29
    rows = slice(b * S, (b + 1) * S)
    noise = O.dropout_noise(H * S * S, p, seed, b * H * S * S // 8).reshape(H, S, S).astype(np.float64)
    ref, grads = _mha_oracle64(mha, x[rows], g[rows], H, 1, p, noise)
    got = out.data()[rows]
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    gx = X.grad()[rows]
    assert np.abs(gx - grads["x"]).max() <= 2e-5 * np.abs(grads["x"]).max()
    # a different mask must NOT match (the check has teeth): the neighbouring sample's mask
    wrong = O.dropout_noise(H * S * S, p, seed, (b - 1) * H * S * S // 8).reshape(H, S, S).astype(np.float64)
    ref_w, _ = _mha_oracle64(mha, x[rows], g[rows], H, 1, p, wrong)
    assert np.abs(got - ref_w).max() > 1e-3 * np.abs(ref).max()


def test_C5_geometry_parameter_gradients_with_dropout(nk):
    """The C5 layer shape (d=1024, h=16, S=1024, dropout 0.1: the L = 1024 instances of the fused probability
    kernels, the K = 64 attention GEMMs, the 1024 x 1024 x (B*S) weight-gradient GEMMs) on a 4-sample batch the f64
    oracle can replay whole: output, input gradient and ALL EIGHT parameter gradients (which sum over samples)."""
    dev = nk.Device(0)
    B, S, d, H, p, seed = 4, 1024, 1024, 16, 0.1, 99
    nk.manual_seed(seed)
    mha = nk.nn.MultiheadAttention(dev, d, H, p, 1)
    x = rnd(0, (B * S, d), -1, 1); g = rnd(5, (B * S, d), -1, 1)
    X = nk.from_ndarray(dev, x).requires_grad()
    out = mha.forward(X, B)
    out.forward(); out.backward_from(nk.from_ndarray(dev, g))
    noise = O.dropout_noise(B * H * S * S, p, seed, 0).reshape(B * H, S, S).astype(np.float64)
    ref, grads = _mha_oracle64(mha, x, g, H, B, p, noise)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(out.data(), ref) <= 1e-5
    assert rel(X.grad(), grads["x"]) <= 2e-5
    for n in "qkvo":
        assert rel(getattr(mha, n).weight.grad(), grads["w" + n]) <= 2e-5, n      # K = B*S = 4096 f32 fma chains
        # bias gradient = column sums of the same dZ the weight gradient contracts; the key bias gradient is exactly
        # zero in exact arithmetic (softmax is shift-invariant along the key axis), so the yardstick is the size of the
        # terms summed, taken from the weight gradient, not the (vanishing) result
        scale = max(np.abs(grads["b" + n]).max(), np.abs(grads["w" + n]).max())
        assert np.abs(getattr(mha, n).bias.grad() - grads["b" + n]).max() <= 2e-5 * scale, n
