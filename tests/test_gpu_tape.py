"""GPU tests of the C++ tape mirror (host/neuronika.hpp via neuronika_amd._tape): graph-level
behaviour of Var / VarDiff as in the reference's neuronika-variable/src/test.rs, and the
BASELINE configurations composed from it, against the CPU oracle."""
import numpy as np
import pytest

from oracle import neuronika_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nk():
    import neuronika_amd
    return neuronika_amd.tape


@pytest.fixture(scope="module")
def tdev(nk):
    return nk.Device(0)


def rnd(seed, shape, lo=0.0, hi=1.0):
    a = np.random.default_rng(seed).random(shape, dtype=np.float32)
    return np.asarray(a * np.float32(hi - lo) + np.float32(lo), dtype=np.float32).reshape(shape)


def close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def test_differentiate_loop(nk, tdev):
    """neuronika-variable/src/test.rs:127-141: x*4 five times -> 1024, grad 1024."""
    x = nk.ones(tdev, []).requires_grad()
    y = x
    for _ in range(5):
        x = x * 4.0
    x.forward()
    x.backward(1.0)
    assert x.data()[()] == 1024.0
    assert y.grad()[()] == 1024.0


def test_scalar_ops(nk, tdev):
    """test.rs:19-125 style: scalar arithmetic on full((2,2), 3)."""
    x = nk.full(tdev, [2, 2], 3.0)
    for op, want in ((lambda v: v + 9.0, 12.0), (lambda v: v - 1.0, 2.0), (lambda v: v * 3.0, 9.0), (lambda v: v / 3.0, 1.0)):
        y = op(x)
        y.forward()
        assert np.array_equal(y.data(), np.full((2, 2), want, np.float32))


def test_tape_lengths_and_laziness(nk, tdev):
    """test.rs:538-806: each op adds exactly one tape entry; nothing is computed before forward()."""
    a, b = nk.rand(tdev, [3, 3], 1).requires_grad(), nk.rand(tdev, [3, 3], 2).requires_grad()
    for build in (lambda: a.mm(b), lambda: a.mm_t(b), lambda: a + b, lambda: a.relu(), lambda: a.softmax(1),
                  lambda: a.log_softmax(0), lambda: a.sum(), lambda: a.mean(), lambda: a.t(),
                  lambda: a.dropout(0.5, nk.Status(True))):
        y = build()
        assert y.history_len() == 1 and y.forward_history_len() == 1
        assert not y.data().any()                 # zero-filled until forward()
    y = (a.mm(b) + a).relu()
    assert y.history_len() == 3
    with pytest.raises(RuntimeError, match="forgot to call .forward"):
        y.backward(1.0)
    shared = a.mm(b)
    z = shared + shared                            # one node reached twice is recorded once
    assert z.history_len() == 2


def test_backward_accumulates_and_no_grad(nk, tdev):
    """matrix_matrix_mul/test.rs:85-91 at graph level + gradient.rs:64-79."""
    a = nk.from_ndarray(tdev, np.linspace(1, 9, 9, dtype=np.float32).reshape(3, 3)).requires_grad()
    b = nk.from_ndarray(tdev, np.linspace(10, 18, 9, dtype=np.float32).reshape(3, 3)).requires_grad()
    y = a.mm(b)
    y.forward(); y.backward(1.0)
    want = np.array([[33, 42, 51]] * 3, np.float32)
    close(a.grad(), want)
    y.backward(1.0)
    close(a.grad(), 2 * want)                      # leaves accumulate
    a.zero_grad(); close(a.grad(), 0 * want)
    h = a.mm(b).relu()                              # fresh intermediate node
    h.forward(); h.backward(1.0); h.backward(1.0)   # intermediate grads are NOT re-zeroed
    close(a.grad(), 3 * want)                       # 1 + 2 (the mm node's grad is 1 then 2)
    h.no_grad()
    with pytest.raises(RuntimeError, match="de-allocated gradient"):
        h.grad()
    h.with_grad()
    a.zero_grad()
    h.backward(1.0)
    close(a.grad(), want)


def test_quickstart_mlp_C1(nk, tdev, golden):
    """Config C1: the 3->5->5->1 MLP of examples/quickstart.rs with its embedded weights,
    batch 64, MSE mean, backward(1.0) — checked against the oracle's mlp_step."""
    q = golden["quickstart_mlp"]
    params = []
    for name in ("lin1", "lin2", "lin3"):
        w = np.asarray(q[f"{name}.weight"]["data"], np.float32).reshape(q[f"{name}.weight"]["dim"])
        b = np.asarray(q[f"{name}.bias"]["data"], np.float32).reshape(q[f"{name}.bias"]["dim"])
        params.append((w, b))
    x, t = rnd(0, (64, 3)), rnd(1, (64, 1))
    loss_ref, grads_ref = O.mlp_step(x, t, params)
    lins = [nk.nn.Linear(nk.from_ndarray(tdev, w).requires_grad(), nk.from_ndarray(tdev, b).requires_grad()) for w, b in params]
    X, T = nk.from_ndarray(tdev, x), nk.from_ndarray(tdev, t)
    out = lins[2].forward(lins[1].forward(lins[0].forward(X).relu()).relu())
    loss = out.mse(T, nk.Reduction.Mean)
    # the reference's spelling `forward(x).relu()` builds the Linear+ReLU node (graph-build peephole, VarDiff::linear_origin)
    assert loss.history_len() == 4 and loss.forward_history_len() == 4   # [Linear+ReLU]x2, Linear, MSE
    was = nk.nn.set_relu_peephole(False)
    try:
        by_node = lins[2].forward(lins[1].forward(lins[0].forward(X).relu()).relu()).mse(T, nk.Reduction.Mean)
    finally:
        nk.nn.set_relu_peephole(was)
    assert was is True and by_node.history_len() == 6 and by_node.forward_history_len() == 6   # [Linear,ReLU]x2, Linear, MSE
    unfused = []
    for w, b in params:
        unfused.append(nk.nn.Linear(nk.from_ndarray(tdev, w).requires_grad(), nk.from_ndarray(tdev, b).requires_grad()))
        unfused[-1].fused = False
    ref_graph = unfused[2].forward(unfused[1].forward(unfused[0].forward(X).relu()).relu()).mse(T, nk.Reduction.Mean)
    assert ref_graph.history_len() == 9 and ref_graph.forward_history_len() == 9   # the reference's [MMT,Add,ReLU]x2, MMT, Add, MSE
    loss.forward(); loss.backward(1.0)
    close(loss.item(), loss_ref, 1e-5, 1e-6)
    for lin, (dw, db) in zip(lins, grads_ref):
        close(lin.weight.grad(), dw, 1e-4, 1e-6)
        close(lin.bias.grad(), db, 1e-4, 1e-6)
    # one SGD step (neuronika-optim/src/sgd/mod.rs:186-236), then zero_grad
    opt = nk.optim.SGD(0.01)
    for lin in lins:
        opt.register(lin.weight); opt.register(lin.bias)
    opt.step()
    close(lins[0].weight.data(), params[0][0] - np.float32(0.01) * grads_ref[0][0], 1e-5, 1e-7)
    opt.zero_grad()
    assert not lins[0].weight.grad().any()


def test_mlp_C4_shape_small_and_linearity(nk, tdev):
    """C4-shaped MLP (three Linear(n,n) + ReLU, MSE mean) at n=256: parity with the oracle, and
    the seed scales every leaf gradient linearly (backward(1/p) = mean over p shards)."""
    n = 256
    x, t = rnd(100, (n, n)), rnd(200, (n, n))
    k = 1 / np.sqrt(n)
    params = [(rnd(s, (n, n), -k, k), rnd(s + 10, (n,), -k, k)) for s in (1, 2, 3)]
    loss_ref, grads_ref = O.mlp_step(x.astype(np.float64), t.astype(np.float64),
                                     [(w.astype(np.float64), b.astype(np.float64)) for w, b in params], seed=0.25)
    lins = [nk.nn.Linear(nk.from_ndarray(tdev, w).requires_grad(), nk.from_ndarray(tdev, b).requires_grad()) for w, b in params]
    X, T = nk.from_ndarray(tdev, x), nk.from_ndarray(tdev, t)
    loss = lins[2].forward(lins[1].forward(lins[0].forward(X).relu()).relu()).mse(T, nk.Reduction.Mean)
    loss.forward(); loss.backward(0.25)
    close(loss.item(), loss_ref, 1e-5)
    for lin, (dw, db) in zip(lins, grads_ref):
        close(lin.weight.grad(), dw, 1e-3, 1e-7)
        close(lin.bias.grad(), db, 1e-3, 1e-7)


def test_conv2d_module(nk, tdev):
    """nn::Conv2d forward = pad -> convolution -> + bias (defined here; `todo!()` in the reference)."""
    conv = nk.nn.GroupedConv2d(tdev, 4, 6, [3, 3], [1, 1], nk.PaddingMode.zero(), [1, 1], [1, 1], 2, 5)
    x = rnd(0, (2, 4, 9, 8))
    X = nk.from_ndarray(tdev, x).requires_grad()
    y = conv.forward(X)
    y.forward()
    w, b = conv.weight.data(), conv.bias.data()
    xp = np.zeros((2, 4, 11, 10), np.float32); O.pad_constant_forward(x, xp, (1, 1), 0.0)
    yr = np.zeros((2, 6, 9, 8), np.float32); O.convolution_forward(xp, w, yr, (1, 1), (1, 1), 2)
    close(y.data(), yr + b, 1e-5, 1e-5)
    s = y.sum(); s.forward(); s.backward(1.0)
    g = np.ones_like(yr)
    dw = np.zeros_like(w); O.convolution_backward_kernel(dw, g, xp, (1, 1), (1, 1), 2)
    dxp = np.zeros_like(xp); O.convolution_backward_input(dxp, g, w, (1, 1), (1, 1), 2)
    close(conv.weight.grad(), dw, 1e-4, 1e-4)
    close(conv.bias.grad().reshape(-1), g.sum((0, 2, 3)), 1e-5, 1e-4)
    close(X.grad(), dxp[:, :, 1:-1, 1:-1], 1e-4, 1e-5)


@pytest.mark.parametrize("groups", [2, 1])
@pytest.mark.parametrize("nd,mode", [(1, "reflective"), (1, "zero"), (3, "replicative"), (3, "constant"), (2, "reflective")])
def test_conv_nd_modules(nk, tdev, nd, mode, groups):
    """nn::Conv1d / Conv2d / Conv3d (neuronika-nn/src/lib.rs:630-916; constructor argument order of the reference's `new`)
    and nn::GroupedConv1d / 2d / 3d (named in the reference's module index, src/lib.rs:783-797; `groups` after `dilation`)
    with every PaddingMode: forward and the three gradients vs the oracle composition pad -> convolution(groups) -> + bias
    (backward of pad = centre slice, all modes)."""
    pm = {"zero": nk.PaddingMode.zero(), "constant": nk.PaddingMode.constant(0.5),
          "reflective": nk.PaddingMode.reflective(), "replicative": nk.PaddingMode.replicative()}[mode]
    spatial = {1: (19,), 2: (9, 8), 3: (5, 6, 7)}[nd]
    kernel, pad, stride, dil = {1: ((3,), (2,), (2,), (1,)), 2: ((3, 2), (1, 2), (1, 2), (2, 1)),
                                3: ((2, 3, 2), (1, 1, 2), (1, 2, 1), (1, 1, 2))}[nd]
    cin, cout = 4, 6
    k_, p_, s_, d_ = ((kernel[0], pad[0], stride[0], dil[0]) if nd == 1 else (list(kernel), list(pad), list(stride), list(dil)))
    if groups == 1:
        conv = getattr(nk.nn, f"Conv{nd}d")(tdev, cin, cout, k_, p_, pm, s_, d_, 3)
    else:
        conv = getattr(nk.nn, f"GroupedConv{nd}d")(tdev, cin, cout, k_, p_, pm, s_, d_, groups, seed=3)
    assert conv.groups == groups and list(conv.padding) == list(pad) and list(conv.stride) == list(stride) and list(conv.dilation) == list(dil)
    w, b = conv.weight.data(), conv.bias.data()
    assert list(w.shape) == [cout, cin // groups] + list(kernel) and list(b.shape) == [cout] + [1] * nd
    bound = np.sqrt(1.0 / (cin // groups * np.prod(kernel)))
    assert np.abs(w).max() <= bound and np.abs(b).max() <= bound
    x = rnd(0, (2, cin) + spatial, -1, 1)
    X = nk.from_ndarray(tdev, x).requires_grad()
    y = conv.forward(X)
    y.forward()
    xp = np.zeros((2, cin) + tuple(n + 2 * p for n, p in zip(spatial, pad)), np.float32)
    if mode in ("zero", "constant"):
        O.pad_constant_forward(x, xp, pad, 0.5 if mode == "constant" else 0.0)
    else:
        O.pad_mode_forward(x, xp, pad, mode)
    out_sp = tuple((n - d * (k - 1) - 1) // s + 1 for n, k, s, d in zip(xp.shape[2:], kernel, stride, dil))
    yr = np.zeros((2, cout) + out_sp, np.float32); O.convolution_forward(xp, w, yr, stride, dil, groups)
    assert list(y.shape) == list(yr.shape)
    close(y.data(), yr + b, 1e-5, 1e-5)
    gw = rnd(7, yr.shape, -1, 1)
    s = (y * nk.from_ndarray(tdev, gw)).sum(); s.forward(); s.backward(1.0)
    dw = np.zeros_like(w); O.convolution_backward_kernel(dw, gw, xp, stride, dil, groups)
    dxp = np.zeros_like(xp); O.convolution_backward_input(dxp, gw, w, stride, dil, groups)
    dx = np.zeros_like(x); O.pad_backward(dx, dxp, pad)
    close(conv.weight.grad(), dw, 1e-4, 1e-4)
    close(conv.bias.grad().reshape(-1), gw.sum(tuple(i for i in range(gw.ndim) if i != 1)), 1e-5, 1e-4)
    close(X.grad(), dx, 1e-4, 1e-5)


def test_conv_module_fused_equals_two_nodes(nk, tdev):
    """ConvNd as one node (bias in the conv epilogue, db from the node's own G) == convolution node + Addition node."""
    x = rnd(1, (3, 4, 10, 9), -1, 1)
    res = {}
    for fused in (True, False):
        conv = nk.nn.GroupedConv2d(tdev, 4, 6, [3, 3], [1, 1], nk.PaddingMode.zero(), [1, 1], [1, 1], 2, 5)
        conv.fused = fused
        X = nk.from_ndarray(tdev, x).requires_grad()
        y = conv.forward(X)
        s = (y * y).sum(); s.forward(); s.backward(1.0)
        res[fused] = [y.data(), X.grad(), conv.weight.grad(), conv.bias.grad(), s.history_len(), s.forward_history_len()]
    # forward: pad, conv(+bias) instead of pad, conv, add; backward: the Pad node's backward is folded into the
    # convolution's as well (zero padding), so two nodes fewer
    assert res[True][5] == res[False][5] - 1
    assert res[True][4] == res[False][4] - 2
    for a, b in zip(res[True][:4], res[False][:4]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("nd,cin,cout,k,pad,stride,dil,groups", [
    (2, 32, 64, [3, 3], [1, 1], [1, 1], [1, 1], 1),      # fast backward-input kernel, crop fused
    (2, 32, 32, [3, 5], [2, 3], [1, 1], [1, 1], 1),      # wider padding than the kernel needs
    (2, 6, 4, [3, 3], [1, 2], [2, 1], [1, 1], 2),        # generic kernel, stride, groups
    (1, 32, 32, [5], [4], [1], [2], 1),
    (3, 4, 4, [2, 3, 2], [1, 0, 2], [1, 1, 1], [1, 1, 2], 1),
])
def test_conv_module_padded_backward_matches_two_nodes(nk, tdev, nd, cin, cout, k, pad, stride, dil, groups):
    """Zero padding: the fused module node writes the unpadded input gradient directly (nk_conv_bwd_input_padded) and
    accumulates into a gradient that another node wrote first; values equal the pad -> conv -> add graph's."""
    spatial = {1: (23,), 2: (9, 13), 3: (5, 6, 7)}[nd]
    x = rnd(7, (2, cin) + spatial, -1, 1)
    res = {}
    for fused in (True, False):
        k_, p_, s_, d_ = (k[0], pad[0], stride[0], dil[0]) if nd == 1 else (k, pad, stride, dil)
        if groups == 1:
            conv = getattr(nk.nn, f"Conv{nd}d")(tdev, cin, cout, k_, p_, nk.PaddingMode.zero(), s_, d_, 9)
        else:
            conv = getattr(nk.nn, f"GroupedConv{nd}d")(tdev, cin, cout, k_, p_, nk.PaddingMode.zero(), s_, d_, groups, 9)
        conv.fused = fused
        X = nk.from_ndarray(tdev, x).requires_grad()
        y = conv.forward(X)
        s = (y * y).sum() + (X * 3.0).sum()          # a second consumer of X: one of the two writes accumulates
        s.forward(); s.backward(1.0)
        res[fused] = [y.data(), X.grad(), conv.weight.grad(), conv.bias.grad()]
    for a, b in zip(res[True][:3], res[False][:3]):
        assert np.array_equal(a, b)
    # the bias gradient of the fused node is summed on the way by the kernel-gradient pass (another order of the same sum)
    np.testing.assert_allclose(res[True][3], res[False][3], rtol=2e-5, atol=2e-5 * np.abs(res[False][3]).max())


def test_chunks_cat_dropout_graph(nk, tdev):
    x = rnd(3, (6, 8))
    X = nk.from_ndarray(tdev, x).requires_grad()
    parts = X.chunks([3, 4])
    assert len(parts) == 4
    y = parts[0].cat([parts[1]], 1).cat([parts[2].cat([parts[3]], 1)], 0)   # reassembles x
    st = nk.Status(True)
    z = y.dropout(0.0, st)
    s = (z * 2.0).sum()
    s.forward(); s.backward(1.0)
    close(y.data(), x)
    close(s.item(), 2 * x.astype(np.float64).sum(), 1e-5)
    close(X.grad(), np.full_like(x, 2.0))
    with pytest.raises(RuntimeError, match="Wrong probability"):
        X.dropout(1.5, st)


def test_mha_C5_small(nk, tdev):
    """C5-shaped composed attention at a small size against the oracle composition, with the
    dropout mask read back from the device (same-mask parity) — p = 0 here, p > 0 in bench."""
    B, S, d, H = 2, 64, 128, 4
    mha = nk.nn.MultiheadAttention(tdev, d, H, 0.0, 11)
    x = rnd(0, (B * S, d))
    X = nk.from_ndarray(tdev, x).requires_grad()
    out = mha.forward(X, B)
    g = rnd(5, (B * S, d))
    loss = (out * nk.from_ndarray(tdev, g)).sum()
    loss.forward(); loss.backward(1.0)
    W = {n: (getattr(mha, n).weight.data().astype(np.float64), getattr(mha, n).bias.data().astype(np.float64)) for n in "qkvo"}
    noise = np.ones((B * H, S, S))
    ref, grads = O.mha_forward_backward(x.astype(np.float64), *W["q"], *W["k"], *W["v"], *W["o"], H, B, 0.0, noise, g.astype(np.float64))
    close(out.data(), ref, 1e-4, 1e-5)
    close(X.grad(), grads["x"], 1e-3, 1e-5)
    for n in "qkvo":
        close(getattr(mha, n).weight.grad(), grads["w" + n], 1e-3, 1e-4)
        close(getattr(mha, n).bias.grad(), grads["b" + n], 1e-3, 1e-4)


def test_pointwise_nodes_graph(nk, tdev):
    """Row f-2 through the tape: a chain of the pointwise nodes + unsqueeze against f64 autograd-free math."""
    x = rnd(1, (6, 5), 0.5, 2.0)
    X = nk.from_ndarray(tdev, x).requires_grad()
    y = ((-(X.ln())).exp().sqrt() + X.sigmoid() * X.tanh() + X.softplus() + X.pow(3).leaky_relu()).unsqueeze(1)
    assert y.shape == [6, 1, 5]
    s = y.sum(); s.forward(); s.backward(1.0)
    x64 = x.astype(np.float64)
    sig = 1 / (1 + np.exp(-x64))
    f = np.sqrt(np.exp(-np.log(x64))) + sig * np.tanh(x64) + np.log1p(np.exp(x64)) + x64 ** 3
    close(s.item(), f.sum(), 1e-5)
    df = -0.5 * x64 ** -1.5 + sig * (1 - sig) * np.tanh(x64) + sig * (1 - np.tanh(x64) ** 2) + sig + 3 * x64 ** 2
    close(X.grad(), df, 2e-5, 1e-6)


def test_nn_init(nk, tdev):
    """neuronika-nn/src/init.rs: gains, fans (trailing extents summed, as there), constant/eye/dirac patterns and
    the range / moments of the random initialisers."""
    I = nk.nn.init
    assert I.calculate_gain("linear") == 1.0 and I.calculate_gain("sigmoid") == 1.0
    close(I.calculate_gain("tanh"), 5 / 3, 1e-7); close(I.calculate_gain("relu"), np.sqrt(2), 1e-7)
    close(I.calculate_gain("leaky_relu"), np.sqrt(2 / (1 + 0.01 ** 2)), 1e-7)
    with pytest.raises(RuntimeError, match="unsupported nonlinearity"):
        I.calculate_gain("gelu")
    P = nk.zeros(tdev, [6, 4, 3, 5]).requires_grad()
    assert I.calculate_fan_in_fan_out(P) == (4.0 * 8, 6.0 * 8)              # (3 + 5), not 3 * 5: init.rs:55
    assert I.calculate_fan_in_fan_out(nk.zeros(tdev, [7, 2]).requires_grad()) == (2.0, 7.0)
    I.constant(P, 2.5); assert np.array_equal(P.data(), np.full((6, 4, 3, 5), 2.5, np.float32))
    I.ones(P); assert P.data().min() == 1.0
    I.zeros(P); assert P.data().max() == 0.0
    I.dirac(P, 2)                                                            # 3 outputs per group, min(3, 4) diagonals
    want = np.zeros((6, 4, 3, 5), np.float32)
    for g in range(2):
        for d in range(3):
            want[g * 3 + d, d, 1, 2] = 1.0
    assert np.array_equal(P.data(), want)
    with pytest.raises(RuntimeError, match="divisible by groups"):
        I.dirac(P, 4)
    E = nk.zeros(tdev, [3, 5]).requires_grad()
    I.eye(E); assert np.array_equal(E.data(), np.eye(3, 5, dtype=np.float32))
    W = nk.zeros(tdev, [256, 128]).requires_grad()
    I.uniform(W, -0.5, 0.25, 3); w = W.data()
    assert w.min() >= -0.5 and w.max() < 0.25 and abs(w.mean() + 0.125) < 0.01
    I.normal(W, 1.0, 2.0, 3); w = W.data()
    assert abs(w.mean() - 1.0) < 0.05 and abs(w.std() - 2.0) < 0.05
    I.xavier_uniform(W, 2.0, 5); w = W.data()
    a = np.sqrt(3.0) * 2.0 * np.sqrt(2.0 / (128 + 256))
    assert np.abs(w).max() <= a and np.abs(w).max() > 0.95 * a
    I.xavier_normal(W, 1.0, 5); w = W.data()
    assert abs(w.std() - np.sqrt(2.0 / 384)) < 0.003
    I.uniform(W, 0.0, 1.0, 9); w1 = W.data().copy(); I.uniform(W, 0.0, 1.0, 9)
    assert np.array_equal(w1, W.data())                                      # reproducible for a seed


def test_lazy_zero_gradients(nk, tdev):
    """Gradients are born with a PENDING zero fill (gradient.rs:47-54 semantics, no memset): an untouched gradient
    reads as zeros, the first writer assigns, later writers accumulate, zero_grad() makes the fill pending again,
    and a gradient fed by two branches still sums both."""
    x = rnd(1, (5, 4), -1, 1)
    X = nk.from_ndarray(tdev, x).requires_grad()
    U = nk.from_ndarray(tdev, x).requires_grad()                 # never used in a graph
    assert np.array_equal(U.grad(), np.zeros_like(x))
    y = (X.relu() + X.relu() * 2.0).sum()                        # two ReLU nodes write into X.grad
    y.forward(); y.backward(1.0)
    m = (x > 0).astype(np.float32)
    close(X.grad(), 3 * m)
    y.backward(1.0)                                              # no zero_grad: leaves keep accumulating
    g2 = X.grad()
    assert g2[m > 0].min() >= 6.0                                # (intermediates accumulate too: reference behaviour)
    X.zero_grad()
    y.no_grad(); y.with_grad()
    y.backward(1.0)
    close(X.grad(), 3 * m)
    w = rnd(2, (4, 4), -1, 1)
    W = nk.from_ndarray(tdev, w).requires_grad()
    z = X.mm(W).mm(W).sum()                                      # W.grad written by two GEMM nodes: beta 0 then beta 1
    X.zero_grad(); z.forward(); z.backward(1.0)
    ones = np.ones((5, 4))
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    close(W.grad(), (x64 @ w64).T @ ones + x64.T @ (ones @ w64.T), 1e-5, 1e-5)
    close(X.grad(), ones @ w64.T @ w64.T, 1e-5, 1e-5)


def test_device_loader_pipeline(nk, tdev):
    """data::DeviceLoader: page-locked records, double-buffered uploads on the copy stream, batches arrive in
    order and intact over several epochs (odd and even batch counts, ragged tail); the refilled leaves drive a
    graph that was built once."""
    rng = np.random.default_rng(0)
    rec, lab = rng.random((70, 6, 5), dtype=np.float32), rng.random((70, 3), dtype=np.float32)
    ds = nk.data.LabeledDataset(rec, lab)
    for bs, drop in ((16, True), (16, False), (10, True), (70, True), (64, False)):
        ranges = nk.data.batch_ranges(70, bs, drop)
        ld = nk.data.DeviceLoader(tdev, ds, bs, drop)
        assert ld.batches() == len(ranges)
        X, Y = nk.zeros(tdev, [ranges[0][1], 6, 5]), nk.zeros(tdev, [ranges[0][1], 3])
        for epoch in range(3):
            for start, rows in ranges:
                assert ld.next_into(X, Y) == rows
                assert np.array_equal(X.data()[:rows], rec[start:start + rows])
                assert np.array_equal(Y.data()[:rows], lab[start:start + rows])
                assert not X.data()[rows:].any() and not Y.data()[rows:].any()   # a ragged tail never keeps the previous batch
            assert ld.next_into(X, Y) == 0                                   # end of the epoch
    ld = nk.data.DeviceLoader(tdev, nk.data.Dataset(rec), 32, False)
    seen = []
    while True:
        rows, x, y = ld.next()
        if rows == 0:
            break
        assert y is None and x.shape == [rows, 6, 5]
        seen.append(x.data())
    assert np.array_equal(np.concatenate(seen), rec)
    # a graph built once, fed by the loader: per-batch loss equals the host computation
    w = rng.random((3, 30), dtype=np.float32)
    W = nk.from_ndarray(tdev, w).requires_grad()
    ld = nk.data.DeviceLoader(tdev, ds, 35, True)
    X, Y = nk.zeros(tdev, [35, 30]), nk.zeros(tdev, [35, 3])
    loss = X.mm_t(W).mse(Y, nk.Reduction.Mean)
    for start in (0, 35):
        assert ld.next_into(X, Y) == 35
        loss.forward()
        ref = ((rec[start:start + 35].reshape(35, 30).astype(np.float64) @ w.T.astype(np.float64) - lab[start:start + 35]) ** 2).mean()
        close(loss.item(), ref, 1e-5)


def test_serde_wire_format(nk, tdev, golden):
    """serde.rs:10-58: leaves travel as ndarray's {"v":1,"dim":[..],"data":[..]}; the quickstart model JSON
    (examples/quickstart.rs:53-169) loads into nn::Linear, and to_json -> from_json is the identity on the bits."""
    import json
    q = golden["quickstart_mlp"]
    for name in ("lin1", "lin2", "lin3"):
        text = json.dumps({p: {"v": 1, "dim": q[f"{name}.{p}"]["dim"], "data": q[f"{name}.{p}"]["data"]} for p in ("weight", "bias")},
                          indent=3)
        lin = nk.serde.linear_from_json(tdev, text)
        for p in ("weight", "bias"):
            want = np.asarray(q[f"{name}.{p}"]["data"], np.float32).reshape(q[f"{name}.{p}"]["dim"])
            assert np.array_equal(getattr(lin, p).data(), want)
            assert getattr(lin, p).grad().shape == want.shape          # VarDiff leaves: requires_grad()
        back = json.loads(nk.serde.to_json(lin))
        assert list(back) == ["weight", "bias"] and back["weight"]["v"] == 1 and back["weight"]["dim"] == q[f"{name}.weight"]["dim"]
        assert np.array_equal(np.asarray(back["weight"]["data"], np.float32).reshape(back["weight"]["dim"]), lin.weight.data())
    x = np.array([[1.0, -0.0, 3.4028235e38], [1e-45, 0.1, -2.5]], np.float32)       # integral, signed zero, max, denormal
    text = nk.serde.to_json(nk.from_ndarray(tdev, x))
    assert text.startswith('{"v":1,"dim":[2,3],"data":[1.0,-0.0,')
    y = nk.serde.var_from_json(tdev, text)
    assert np.array_equal(y.data().view(np.uint32), x.view(np.uint32))
    s0 = nk.serde.var_from_json(tdev, '{"v":1,"dim":[],"data":[2.5]}')
    assert s0.shape == [] and s0.item() == 2.5
    for bad in ('{"v":2,"dim":[1],"data":[0.0]}', '{"v":1,"dim":[2],"data":[0.0]}', '{"v":1,"dim":[1]}', '[1,2'):
        with pytest.raises(RuntimeError, match="json"):
            nk.serde.var_from_json(tdev, bad)


def test_linear_fused_equals_two_nodes(nk, tdev):
    """nn::Linear as one node (bias in the GEMM epilogue, backward = the three reference accumulations) gives
    bit-identical values and gradients to the reference's mm_t + Addition nodes, for Var and VarDiff inputs."""
    x = rnd(1, (96, 40), -1, 1)
    res = {}
    for fused in (True, False):
        l1, l2 = nk.nn.Linear(tdev, 40, 64, 7), nk.nn.Linear(tdev, 64, 24, 9)
        l1.fused = l2.fused = fused
        y = l2.forward(l1.forward(nk.from_ndarray(tdev, x)).relu())
        s = (y * y).sum(); s.forward(); s.backward(1.0)
        assert y.history_len() == (2 if fused else 5)                 # backward nodes: Linear+ReLU, Linear | 2x(mm_t, +) + ReLU
        res[fused] = [y.data()] + [p.grad() for p in (l1.weight, l1.bias, l2.weight, l2.bias)]
    for a, b in zip(res[True], res[False]):
        assert np.array_equal(a, b)


def test_linear_forward_relu_equals_the_two_nodes(nk, tdev):
    """`nn::Linear::forward_relu` (ReLU in the forward GEMM's epilogue, its backward mask applied by the GEMM that produces
    the node's output gradient) gives bit-identical values and gradients to `forward(x).relu()`:
      (a) a chain of Linear layers - every writer of the fused nodes' gradients is mask-capable, no ReLU kernel runs at all;
      (b) the fused output feeds a NON-Linear consumer - the writers store plain values and the owner masks in place;
      (c) a diamond: one Linear consumer and one other consumer of the same fused output (mixed writers -> fallback);
      (d) the fused node is the root of `backward` / `backward_from` (seed plain; the caller's seed tensor stays untouched);
      (e) a second pass after `no_grad(); with_grad()` (fresh intermediate gradients, as a training loop does) keeps
          accumulating on the leaves exactly as the node-by-node graph does.  (Without that reset the reference ALSO keeps
          stale sums in every intermediate gradient, vardiff.rs:125-141; a fused node has one such buffer fewer, so only
          the fresh-gradient case is comparable - SURVEY.md 8c, "parity is defined for ... a fresh graph".)"""
    x, t = rnd(1, (96, 40), -1, 1), rnd(2, (96, 24), -1, 1)
    side = rnd(3, (96, 64), -1, 1)

    def by_nodes(l, v):                                               # the ReLU node over the Linear's output: peephole off
        was = nk.nn.set_relu_peephole(False)
        try:
            return l.forward(v).relu()
        finally:
            nk.nn.set_relu_peephole(was)

    def build(fused, case):
        l1, l2, l3 = nk.nn.Linear(tdev, 40, 64, 7), nk.nn.Linear(tdev, 64, 64, 9), nk.nn.Linear(tdev, 64, 24, 11)
        X = nk.from_ndarray(tdev, x).requires_grad()
        # True: the explicit node; "spelling": the reference's words `forward(x).relu()` (the peephole builds the same node);
        # False: node by node
        act = {True: lambda l, v: l.forward_relu(v), "spelling": lambda l, v: l.forward(v).relu(), False: by_nodes}[fused]
        a1 = act(l1, X)
        if case == "chain":
            root = l3.forward(act(l2, a1)).mse(nk.from_ndarray(tdev, t), nk.Reduction.Mean)
        elif case == "other_consumer":
            root = (a1 * nk.from_ndarray(tdev, side)).sum()
        elif case == "diamond":
            a2 = act(l2, a1)
            root = (l3.forward(a2).mse(nk.from_ndarray(tdev, t), nk.Reduction.Sum) + (a1 * nk.from_ndarray(tdev, side)).sum()) + (a2 * a2).sum()
        else:
            root = a1
        return root, X, [l1, l2, l3]

    def grads(X, lins):
        out = [X.grad()]
        for l in lins:
            for p in (l.weight, l.bias):
                out.append(p.grad())
        return out

    for case in ("chain", "other_consumer", "diamond", "root"):
        res = {}
        for fused in (True, "spelling", False):
            root, X, lins = build(fused, case)
            root.forward()
            if case == "root":
                seed = rnd(5, (96, 64), -1, 1)
                S = nk.from_ndarray(tdev, seed)
                root.backward_from(S)
                assert np.array_equal(S.data(), seed)                  # the caller's tensor is not masked in place
                used = [lins[0]]
            else:
                root.backward(0.5)
                root.no_grad(); root.with_grad()                       # (e): fresh intermediates, the leaves keep accumulating
                root.backward(0.25)
                used = lins if case != "other_consumer" else [lins[0]]
            res[fused] = [root.data()] + grads(X, used)
        for i, (a, b, c) in enumerate(zip(res[True], res[False], res["spelling"])):
            assert np.array_equal(a, b) and np.array_equal(c, b), (case, i)
    # (a) really runs without ReLU launches and without the pre-activation: 3 backward nodes, not 5
    for how, nodes in ((True, 4), ("spelling", 4), (False, 6)):
        root, X, lins = build(how, "chain")
        assert root.history_len() == nodes, how                        # 3 Linear(+ReLU) nodes + the loss | + 2 ReLU nodes
    # a root's gradient stays the seed (vardiff.rs:133) in every spelling: the fused node masks a scratch copy, not the buffer
    for how in (True, "spelling", False):
        root, X, lins = build(how, "root")
        root.forward(); root.backward(0.5)
        assert np.all(root.grad() == 0.5), how
    # a Linear output that is KEPT (`z = l1.forward(X)` - in Rust `z.clone().relu()`, since `relu(self)` consumes its operand) is not
    # folded: `z.relu()` is the ReLU node over z whatever the peephole setting - z is in the root's history, its data is computed and its
    # gradient receives the ReLU path's contribution, as in the reference's graph (only a TEMPORARY may fold: nobody can look at it)
    res = {}
    for peephole in (True, False):
        was = nk.nn.set_relu_peephole(peephole)
        try:
            l1 = nk.nn.Linear(tdev, 40, 64, 7)
            X = nk.from_ndarray(tdev, x).requires_grad()
            z = l1.forward(X)
            root = (z * nk.from_ndarray(tdev, side) + z.relu()).sum()
        finally:
            nk.nn.set_relu_peephole(was)
        root.forward(); root.backward(1.0)
        res[peephole] = [root.data(), z.data(), z.grad(), X.grad(), l1.weight.grad(), l1.bias.grad(), root.history_len()]
    for a, b in zip(res[True], res[False]):
        assert np.array_equal(a, b)
    zd, zg = res[True][1], res[True][2]
    assert np.abs(zd).max() > 0 and np.array_equal(zg, side + (zd > 0).astype(np.float32))   # d root / dz = side + (z > 0)


@pytest.mark.parametrize("B,S,H,d,p", [(2, 128, 2, 128, 0.0), (3, 100, 4, 128, 0.2), (2, 64, 2, 256, 0.1)])
def test_mha_packed_projections_equal_three_linear_nodes(nk, tdev, B, S, H, d, p):
    """`nn::MultiheadAttention` with packed Q / K / V projections (ONE GEMM with N = 3d, one input-gradient GEMM with K = 3d,
    one weight-gradient GEMM with M = 3d, attention kernels reading the packed output) against the same module run as three
    Linear nodes + the attention node (`packed_qkv = False`) on the same weights and the same dropout mask:
      * q / k / v are ordinary parameters that happen to share one allocation (views): data(), grad(), optimizers work;
      * output, dQ-side quantities and every PARAMETER gradient are bit-identical (same chains: the packing only moves tiles);
      * the input gradient is one K = 3d chain instead of three K = d chains added up - equal to contraction tolerance;
      * zero_grad + a second pass accumulates on the packed views exactly like the unpacked module."""
    x, g = rnd(1, (B * S, d), -1, 1), rnd(2, (B * S, d), -1, 1)
    res = {}
    for packed in (True, False):
        nk.manual_seed(123)
        mha = nk.nn.MultiheadAttention(tdev, d, H, p, 5)
        mha.packed_qkv = packed
        X = nk.from_ndarray(tdev, x).requires_grad()
        out = mha.forward(X, B)
        assert out.history_len() == (2 if packed else 5)              # [projections + core] + out-projection | q, k, v, core, out
        out.forward(); out.backward_from(nk.from_ndarray(tdev, g))
        first = [getattr(mha, n).weight.grad().copy() for n in "qkvo"]
        out.no_grad(); out.with_grad()
        out.forward(); out.backward_from(nk.from_ndarray(tdev, g))    # leaves accumulate: 2x (fresh dropout mask in the second pass)
        res[packed] = dict(out=out.data(), dx=X.grad(), first=first,
                           w=[getattr(mha, n).weight.grad() for n in "qkvo"], b=[getattr(mha, n).bias.grad() for n in "qkvo"],
                           data=[getattr(mha, n).weight.data() for n in "qkvo"])
    a, b = res[True], res[False]
    for u, v in zip(a["data"], b["data"]):
        assert np.array_equal(u, v)                                    # same initialisation law, packed or not
    assert np.array_equal(a["out"], b["out"])
    for key in ("first", "w"):
        for i, (u, v) in enumerate(zip(a[key], b[key])):
            assert np.array_equal(u, v), (key, "qkvo"[i])
    for i, (u, v) in enumerate(zip(a["b"], b["b"])):                   # column sums of a (n, 3d) / (n, d) matrix: the row partition of the
        np.testing.assert_allclose(u, v, rtol=0, atol=1e-5 * max(1.0, float(np.abs(v).max())), err_msg="qkvo"[i])   # two-pass sum may differ
    scale = np.abs(b["dx"]).max()
    assert np.abs(a["dx"] - b["dx"]).max() <= 2e-6 * 3 * d * scale / np.sqrt(3 * d) + 1e-6 * scale
    # an optimizer step on the views moves the packed storage
    mha = nk.nn.MultiheadAttention(tdev, d, H, 0.0, 5)
    X = nk.from_ndarray(tdev, x).requires_grad()
    out = mha.forward(X, B)
    loss = (out * out).sum(); loss.forward(); loss.backward(1.0)
    opt = nk.optim.SGD(0.01)
    for n in "qkvo":
        opt.register(getattr(mha, n).weight); opt.register(getattr(mha, n).bias)
    w0, g0 = mha.k.weight.data().copy(), mha.k.weight.grad().copy()
    opt.step(); opt.zero_grad()
    np.testing.assert_allclose(mha.k.weight.data(), w0 - np.float32(0.01) * g0, rtol=1e-6, atol=1e-7)   # (the kernel may contract w - g * lr into one fma)
    assert not mha.k.weight.grad().any() and np.abs(mha.k.weight.data() - w0).max() > 0
    out2_before = out.data().copy()
    loss.forward()                                                     # (a second forward() of the same root recomputes every node, var.rs:110-128)
    assert not np.array_equal(out.data(), out2_before)                # the forward GEMM reads the updated packed weights


def test_losses_gemv_stack_graph(nk, tdev):
    """Row f-4 through the tape: classifier head x.mm_t(W) -> log_softmax -> nll; bce / bce_with_logits / kldiv /
    mae heads; mv / vm / vv; stack — values and gradients against the oracle nodes."""
    R = nk.Reduction
    x, w = rnd(1, (8, 6), -1, 1), rnd(2, (5, 6), -1, 1)
    t = np.array([0, 4, 2, 2, 1, 3, 0, 4], np.float32)
    W = nk.from_ndarray(tdev, w).requires_grad()
    logits = nk.from_ndarray(tdev, x).mm_t(W)
    loss = logits.log_softmax(1).nll(nk.from_ndarray(tdev, t), R.Mean)
    loss.forward(); loss.backward(1.0)
    z = x @ w.T; lz = np.zeros_like(z); O.log_softmax_forward(z, lz, 1)
    close(loss.item(), O.nll_forward(lz, t, "mean"), 1e-5)
    dlz = np.zeros_like(z); O.nll_backward(dlz, 1.0, t, "mean")
    dz = np.zeros_like(z); O.log_softmax_backward(dz, dlz, lz, 1)
    close(W.grad(), dz.T @ x, 1e-4, 1e-6)

    p, q = rnd(3, (4, 7), 0.05, 0.95), rnd(4, (4, 7), 0.0, 1.0)
    Q = nk.from_ndarray(tdev, q)
    for name, fwd, bwd, inp in (("bce", O.bce_forward, O.bce_backward, p), ("mae", O.mae_forward, O.mae_backward, p),
                                ("bce_with_logits", O.bce_with_logits_forward, O.bce_with_logits_backward, 4 * p - 2)):
        for red, rname in ((R.Mean, "mean"), (R.Sum, "sum")):
            P = nk.from_ndarray(tdev, inp).requires_grad()
            l = getattr(P, name)(Q, red); l.forward(); l.backward(1.0)
            close(l.item(), fwd(inp, q, rname), 2e-5)
            d = np.zeros_like(inp); bwd(d, 1.0, inp, q, rname); close(P.grad(), d, 1e-5, 1e-6)
    P = nk.from_ndarray(tdev, np.log(p)).requires_grad()
    l = P.kldiv(Q, R.Mean); l.forward(); l.backward(1.0)
    close(l.item(), O.kldiv_forward(np.log(p), q, "mean"), 2e-5)
    d = np.zeros_like(p); O.kldiv_backward(d, 1.0, q, "mean"); close(P.grad(), d, 1e-5, 1e-6)

    a, v, u = rnd(5, (6, 9), -1, 1), rnd(6, (9,), -1, 1), rnd(7, (6,), -1, 1)
    A, V, U = (nk.from_ndarray(tdev, z).requires_grad() for z in (a, v, u))
    s = A.mv(V).vv(U) + U.vm(A).vv(V)            # u.(A v) twice
    s.forward(); s.backward(1.0)
    close(s.item(), 2 * (u @ a @ v), 1e-5)
    close(A.grad(), 2 * np.outer(u, v), 1e-5, 1e-6); close(V.grad(), 2 * (a.T @ u), 1e-5, 1e-6); close(U.grad(), 2 * (a @ v), 1e-5, 1e-6)

    b, c = rnd(8, (3, 4)), rnd(9, (3, 4))
    B, C = nk.from_ndarray(tdev, b).requires_grad(), nk.from_ndarray(tdev, c).requires_grad()
    for axis in (0, 1, 2):
        st = B.stack([C, B], axis)
        ref = np.stack([b, c, b], axis)
        assert list(st.shape) == list(ref.shape)
        wgt = rnd(10 + axis, ref.shape)
        B.zero_grad(); C.zero_grad()
        l = (st * nk.from_ndarray(tdev, wgt)).sum(); l.forward(); l.backward(1.0)
        close(st.data(), ref)
        close(C.grad(), np.take(wgt, 1, axis), 1e-6); close(B.grad(), np.take(wgt, 0, axis) + np.take(wgt, 2, axis), 1e-6)
    with pytest.raises(RuntimeError, match="nll"):
        nk.from_ndarray(tdev, x).nll(nk.from_ndarray(tdev, x), R.Mean)


def test_lstm_gru_cells(nk, tdev):
    """nn::LSTMCell / nn::GRUCell (neuronika-nn/src/lib.rs:512-541, 602-624) vs the oracle composition in
    f64; gradients vs central differences of the f64 oracle."""
    B, I, H = 5, 7, 6
    x, h0, c0 = rnd(1, (B, I), -1, 1), rnd(2, (B, H), -1, 1), rnd(3, (B, H), -1, 1)
    wt = rnd(4, (B, H), -1, 1)                               # loss = sum(out * wt)
    X = nk.from_ndarray(tdev, x)
    Wt = nk.from_ndarray(tdev, wt)
    names = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")

    lstm = nk.nn.LSTMCell(tdev, I, H, seed=11)
    Hh, Cc = nk.from_ndarray(tdev, h0).requires_grad(), nk.from_ndarray(tdev, c0).requires_grad()
    new_c, new_h = lstm.forward((Cc, Hh), X)
    loss = (new_h * Wt).sum() + (new_c * Wt).sum()
    loss.forward(); loss.backward(1.0)
    p = {n: getattr(lstm, n).data().astype(np.float64) for n in names}
    st = {"c": c0.astype(np.float64), "h": h0.astype(np.float64)}
    x64, w64 = x.astype(np.float64), wt.astype(np.float64)

    def f_lstm():
        c, h = O.lstm_cell_forward(st["c"], st["h"], x64, p["weight_ih"], p["weight_hh"], p["bias_ih"], p["bias_hh"])
        return float((h * w64).sum() + (c * w64).sum())
    c_ref, h_ref = O.lstm_cell_forward(st["c"], st["h"], x64, *(p[n] for n in names))
    close(new_c.data(), c_ref, 2e-5, 1e-6); close(new_h.data(), h_ref, 2e-5, 1e-6)
    for n in names:
        close(getattr(lstm, n).grad(), O.numeric_grad(f_lstm, p[n]), 2e-4, 2e-5)
    close(Hh.grad(), O.numeric_grad(f_lstm, st["h"]), 2e-4, 2e-5)
    close(Cc.grad(), O.numeric_grad(f_lstm, st["c"]), 2e-4, 2e-5)

    gru = nk.nn.GRUCell(tdev, I, H, seed=21)
    Hg = nk.from_ndarray(tdev, h0).requires_grad()
    out = gru.forward(Hg, X)
    loss = (out * Wt).sum(); loss.forward(); loss.backward(1.0)
    q = {n: getattr(gru, n).data().astype(np.float64) for n in names}
    hg = h0.astype(np.float64)

    def f_gru():
        return float((O.gru_cell_forward(hg, x64, *(q[n] for n in names)) * w64).sum())
    close(out.data(), O.gru_cell_forward(hg, x64, *(q[n] for n in names)), 2e-5, 1e-6)
    for n in names:
        close(getattr(gru, n).grad(), O.numeric_grad(f_gru, q[n]), 2e-4, 2e-5)
    close(Hg.grad(), O.numeric_grad(f_gru, hg), 2e-4, 2e-5)


@pytest.mark.parametrize("make", ["sgd", "sgd_momentum", "adam", "amsgrad", "adagrad", "rmsprop"])
def test_optimizers_reduce_loss(nk, tdev, make):
    """neuronika-optim/src/*/test.rs: "loss after 10 steps < initial loss" on a random problem."""
    opt = {"sgd": lambda: nk.optim.SGD(0.05), "sgd_momentum": lambda: nk.optim.SGD(0.02, momentum=0.9, nesterov=True),
           "adam": lambda: nk.optim.Adam(0.05), "amsgrad": lambda: nk.optim.Adam(0.05, amsgrad=True),
           "adagrad": lambda: nk.optim.Adagrad(0.2), "rmsprop": lambda: nk.optim.RMSProp(0.02, momentum=0.5, centered=True)}[make]()
    lin = nk.nn.Linear(tdev, 3, 3, 7)
    opt.register(lin.weight); opt.register(lin.bias)
    X, T = nk.rand(tdev, [16, 3], 1), nk.rand(tdev, [16, 3], 2)
    loss = lin.forward(X).mse(T, nk.Reduction.Mean)
    loss.forward()
    first = loss.item()
    for _ in range(10):
        loss.forward(); loss.no_grad(); loss.with_grad(); loss.backward(1.0)
        opt.step(); opt.zero_grad()
    loss.forward()
    assert loss.item() < first


def test_mha_fused_equals_unfused_with_dropout(nk, tdev):
    """The one-node attention probabilities give the same result as the three reference nodes."""
    B, S, d, H = 2, 64, 128, 4
    x = rnd(0, (B * S, d)); g = rnd(5, (B * S, d))
    outs = []
    for fused in (True, False):
        mha = nk.nn.MultiheadAttention(tdev, d, H, 0.0, 11)
        mha.fused, mha.fused_core = fused, False               # (the node-by-node paths: dh = 32 would take the fused core)
        X = nk.from_ndarray(tdev, x).requires_grad()
        loss = (mha.forward(X, B) * nk.from_ndarray(tdev, g)).sum()
        n_nodes = loss.history_len()
        loss.forward(); loss.backward(1.0)
        outs.append((loss.item(), X.grad().copy(), mha.q.weight.grad().copy(), n_nodes))
    assert outs[0][3] == outs[1][3] - 2                       # three nodes became one
    close(outs[0][0], outs[1][0], 1e-6)
    close(outs[0][1], outs[1][1], 1e-5, 1e-7)
    close(outs[0][2], outs[1][2], 1e-5, 1e-6)
    # with dropout: outputs are scaled copies of the no-dropout probabilities where kept
    mha = nk.nn.MultiheadAttention(tdev, d, H, 0.25, 11)
    X = nk.from_ndarray(tdev, x).requires_grad()
    out = mha.forward(X, B)
    out.forward()
    assert np.isfinite(out.data()).all()
    mha.drop.eval(); out.forward(); ev = out.data().copy()
    mha0 = nk.nn.MultiheadAttention(tdev, d, H, 0.0, 11)
    o0 = mha0.forward(nk.from_ndarray(tdev, x).requires_grad(), B); o0.forward()
    close(ev, o0.data(), 1e-6, 1e-7)                         # eval mode == no dropout


def _mha_oracle(mha, x, g, H, B, p, noise, dt=np.float64):
    W = [getattr(mha, n).weight.data().astype(dt) for n in "qkvo"]
    Bs = [getattr(mha, n).bias.data().astype(dt) for n in "qkvo"]
    return O.mha_forward_backward(x.astype(dt), W[0], Bs[0], W[1], Bs[1], W[2], Bs[2], W[3], Bs[3], H, B, p,
                                  noise.astype(dt), g.astype(dt))


@pytest.mark.parametrize("fused,strided", [(True, True), (False, True), (True, False), (False, False)])
@pytest.mark.parametrize("p", [0.1, 0.5])
def test_mha_with_dropout_equals_oracle(nk, tdev, fused, strided, p):
    """The module AS BENCHMARKED (dropout in training mode) end to end against the oracle composition of SURVEY.md
    section 8a: the oracle is fed the Philox mask the device draws (`O.dropout_noise` with the node's key and offset,
    key fixed by `manual_seed`), and the output, the input gradient and all eight parameter gradients must agree -
    for the fused node (stored and recomputed probabilities), the three reference nodes, and both head layouts.
    A second forward resamples the mask (offset advances) and is checked the same way."""
    B, S, d, H = 2, 64, 128, 4
    x, g = rnd(0, (B * S, d), -1, 1), rnd(5, (B * S, d), -1, 1)
    seed = 1234567
    nk.manual_seed(seed)
    mha = nk.nn.MultiheadAttention(tdev, d, H, p, 3)
    mha.fused, mha.strided_heads, mha.fused_core = fused, strided, False   # the paths BESIDE the fused core (tested below)
    X = nk.from_ndarray(tdev, x).requires_grad()
    y = mha.forward(X, B)
    G = nk.from_ndarray(tdev, g)
    leaves = [X] + [getattr(getattr(mha, n), w) for n in "qkvo" for w in ("weight", "bias")]
    n = B * H * S * S
    for call in range(2):
        noise = O.dropout_noise(n, p, seed, call * O.dropout_draws_calls(n)).reshape(B * H, S, S)
        assert 0.5 * (1 - p) < noise.mean() < min(1.0, 1.5 * (1 - p))
        for v in leaves:
            v.zero_grad()
        y.forward(); y.no_grad(); y.with_grad()
        y.backward_from(G)
        ref, grads = _mha_oracle(mha, x, g, H, B, p, noise)
        ref32, grads32 = _mha_oracle(mha, x, g, H, B, p, noise, np.float32)
        def check(got, want, want32, what, floor=0.0):
            scale = max(np.abs(want).max(), floor)
            err_gpu, err_cpu = np.abs(got - want).max(), np.abs(want32 - want).max()
            from conftest import record_margin
            record_margin("mha_module:" + what, err_gpu, err_cpu, 1e-6 * scale)
            assert err_gpu <= max(2 * err_cpu, 1e-6 * scale), (what, call, err_gpu, err_cpu, scale)   # SURVEY 8c (ii) as stated
        check(y.data(), ref, ref32, "out")
        check(X.grad(), grads["x"], grads32["x"], "dx")
        for nme in "qkvo":
            check(getattr(mha, nme).weight.grad(), grads["w" + nme], grads32["w" + nme], "dw" + nme)
            # (the key bias gradient is exactly zero in exact arithmetic: yardstick = the weight gradient's size)
            check(getattr(mha, nme).bias.grad(), grads["b" + nme], grads32["b" + nme], "db" + nme, np.abs(grads["w" + nme]).max())
    if call == 1:                                    # the two forwards drew different masks
        assert not np.array_equal(O.dropout_noise(n, p, seed, 0), noise.reshape(-1))


@pytest.mark.parametrize("core", [True, False])
@pytest.mark.parametrize("p,S,d,H", [(0.1, 96, 128, 2), (0.0, 64, 128, 2), (0.3, 160, 128, 2), (0.1, 96, 128, 4), (0.2, 64, 256, 2),   # dh = 64, 64, 64, 32, 128
                                     (0.1, 100, 128, 2), (0.25, 36, 128, 4), (0.0, 44, 256, 2)])   # ragged lengths: dh = 64, 32, 128
def test_mha_fused_attention_core_equals_oracle(nk, tdev, core, p, S, d, H):
    """Head dimension 64 (the C5 geometry): the module routes scores -> probabilities -> context through the fused
    attention kernels (`fused_core`, one node instead of three).  Same check as above - the oracle composition fed the
    device's Philox mask, output, input gradient and all eight parameter gradients, two forwards (the mask is
    resampled) - for the fused core and for the node-by-node path it replaces, which must also agree with each other.
    Head dimensions 64 (the C5 geometry), 32 and 128: the three instantiations of the fused kernels."""
    B = 2
    x, g = rnd(0, (B * S, d), -1, 1), rnd(5, (B * S, d), -1, 1)
    seed = 7654321
    nk.manual_seed(seed)
    mha = nk.nn.MultiheadAttention(tdev, d, H, p, 3)
    mha.fused_core = core
    X = nk.from_ndarray(tdev, x).requires_grad()
    y = mha.forward(X, B)
    other = nk.nn.MultiheadAttention(tdev, d, H, p, 3)
    other.fused_core = not core
    # backward nodes: [packed projections + fused core] + out-projection = 2 | q, k, v, scores, probabilities, context, out = 7
    assert y.history_len() == other.forward(nk.from_ndarray(tdev, x).requires_grad(), B).history_len() + (-5 if core else 5)
    G = nk.from_ndarray(tdev, g)
    leaves = [X] + [getattr(getattr(mha, n), w) for n in "qkvo" for w in ("weight", "bias")]
    # the fused core indexes its draws in the score tensor padded to whole 32 x 32 tiles (include/neuronika_hip.h); the node path in (B*H, S, S)
    SP = (S + 31) // 32 * 32 if core else S
    n = B * H * SP * SP
    for call in range(2):
        noise = (np.ascontiguousarray(O.dropout_noise(n, p, seed, call * O.dropout_draws_calls(n)).reshape(B * H, SP, SP)[:, :S, :S]) if p
                 else np.ones((B * H, S, S), np.float32))
        for v in leaves:
            v.zero_grad()
        y.forward(); y.no_grad(); y.with_grad()
        y.backward_from(G)
        ref, grads = _mha_oracle(mha, x, g, H, B, p, noise)
        ref32, grads32 = _mha_oracle(mha, x, g, H, B, p, noise, np.float32)
        def check(got, want, want32, what, floor=0.0):
            scale = max(np.abs(want).max(), floor)
            err_gpu, err_cpu = np.abs(got - want).max(), np.abs(want32 - want).max()
            from conftest import record_margin
            record_margin("mha_module:" + what, err_gpu, err_cpu, 1e-6 * scale)
            assert err_gpu <= max(2 * err_cpu, 1e-6 * scale), (what, call, err_gpu, err_cpu, scale)   # SURVEY 8c (ii) as stated
        check(y.data(), ref, ref32, "out")
        check(X.grad(), grads["x"], grads32["x"], "dx")
        for nme in "qkvo":
            check(getattr(mha, nme).weight.grad(), grads["w" + nme], grads32["w" + nme], "dw" + nme)
            check(getattr(mha, nme).bias.grad(), grads["b" + nme], grads32["b" + nme], "db" + nme, np.abs(grads["w" + nme]).max())
    # eval mode: no mask
    mha.drop.eval()
    for v in leaves:
        v.zero_grad()
    y.forward(); y.no_grad(); y.with_grad(); y.backward_from(G)
    ref, grads = _mha_oracle(mha, x, g, H, B, 0.0, np.ones((B * H, S, S), np.float32))
    np.testing.assert_allclose(y.data(), ref, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(X.grad(), grads["x"], rtol=1e-3, atol=2e-6 * np.abs(grads["x"]).max() + 1e-7)


def test_heads_attention_node_without_gradients_keeps_nothing(nk, tdev):
    """`Var::heads_attention` (a graph without gradients: inference) gives the output of the `VarDiff` node bit for bit
    and adds ONE forward node; the shapes the fused kernels do not take are refused with the reason."""
    B, S, H, dh = 2, 64, 2, 64
    q, k, v = (rnd(s_, (B * S, H * dh), -1, 1) for s_ in (1, 2, 3))
    st = nk.Status()
    nk.manual_seed(99)
    a = nk.from_ndarray(tdev, q).heads_attention(nk.from_ndarray(tdev, k), nk.from_ndarray(tdev, v), B, S, H, dh, 0.125, 0.1, st)
    nk.manual_seed(99)
    b = nk.from_ndarray(tdev, q).requires_grad().heads_attention(nk.from_ndarray(tdev, k).requires_grad(),
                                                                 nk.from_ndarray(tdev, v).requires_grad(), B, S, H, dh, 0.125, 0.1, st)
    assert a.history_len() == 1
    a.forward(); b.forward()
    assert np.array_equal(a.data(), b.data())
    assert nk.Var.attention_core_supported(1024, 64, 0.1) and nk.Var.attention_core_supported(1024, 32, 0.1)
    assert nk.Var.attention_core_supported(1024, 128, 0.1) and not nk.Var.attention_core_supported(1024, 16, 0.1)
    with pytest.raises(RuntimeError, match="dh in"):
        nk.from_ndarray(tdev, rnd(4, (64, 32), -1, 1)).heads_attention(nk.from_ndarray(tdev, rnd(5, (64, 32), -1, 1)),
                                                                        nk.from_ndarray(tdev, rnd(6, (64, 32), -1, 1)), 1, 64, 2, 16, 0.1, 0.0, st)


def test_mha_packed_views_guards(nk, tdev):
    """(a) d_model % 4 != 0 - the packed bias views would start 16-byte-unaligned, and `nn::init::constant` /
    the pointwise kernels refuse such buffers: the module then holds three ordinary layers; (b) q / k / v are assignable
    members - after `mha.q = Linear(...)` forward must run on the NEW layer (three-node path), not on the stale packed
    storage; (c) a parameter registered twice with SGD is updated twice, one after the other."""
    mha = nk.nn.MultiheadAttention(tdev, 6, 3, 0.0, 5)               # d_model = 6: not a multiple of 4
    for lin in (mha.q, mha.k, mha.v):
        nk.nn.init.constant(lin.bias, 0.25)                          # nk_fill with a value on every bias
        assert np.all(lin.bias.data() == 0.25)
    X = nk.from_ndarray(tdev, rnd(1, (2 * 5, 6), -1, 1)).requires_grad()
    y = mha.forward(X, 2); s = (y * y).sum(); s.forward(); s.backward(1.0)
    assert np.isfinite(X.grad()).all() and np.abs(mha.k.bias.grad()).sum() >= 0

    d, H, B, S = 128, 2, 2, 64
    x = rnd(2, (B * S, d), -1, 1)
    ref = nk.nn.MultiheadAttention(tdev, d, H, 0.0, 7)
    new_q = nk.nn.Linear(tdev, d, d, 99)
    outs = []
    for packed in (True, False):
        mha = nk.nn.MultiheadAttention(tdev, d, H, 0.0, 7)
        mha.packed_qkv = packed
        mha.q = nk.nn.Linear(nk.from_ndarray(tdev, new_q.weight.data()).requires_grad(), nk.from_ndarray(tdev, new_q.bias.data()).requires_grad())
        X = nk.from_ndarray(tdev, x).requires_grad()
        y = mha.forward(X, B)
        assert y.history_len() == 5                                  # q, k, v, core, out-projection: the packed node is not taken
        s = (y * y).sum(); s.forward(); s.backward(1.0)
        outs.append([y.data(), X.grad(), mha.q.weight.grad(), mha.k.weight.grad()])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert np.abs(outs[0][2]).sum() > 0                              # the gradient landed in the NEW layer
    yref = ref.forward(nk.from_ndarray(tdev, x).requires_grad(), B); yref.forward()
    assert not np.array_equal(yref.data(), outs[0][0])

    w = nk.from_ndarray(tdev, np.ones((4, 4), np.float32)).requires_grad()
    opt = nk.optim.SGD(0.5)
    opt.register(w); opt.register(w)
    (w * 1.0).sum().forward()
    loss = (w * 2.0).sum(); loss.forward(); loss.backward(1.0)      # grad = 2
    opt.step()
    assert np.all(w.data() == 1.0 - 2 * 0.5 * 2.0)                  # two sequential updates


def test_backward_from_equals_weighted_sum_scaffolding(nk, tdev):
    """`y.backward_from(G)` (the upstream gradient tensor stands in for the root gradient while the tape runs - no copy,
    no extra nodes) gives bit-identical leaf gradients to the scaffolding `(y * G).sum().backward(1.0)`; the seed is left
    untouched, the root gradient is its own buffer again afterwards, and a second call accumulates like backward()."""
    x, g = rnd(1, (24, 40), -1, 1), rnd(2, (24, 16), -1, 1)
    def leaves():
        lin = nk.nn.Linear(tdev, 40, 16, 5)
        X = nk.from_ndarray(tdev, x).requires_grad()
        return lin, X, lin.forward(X).relu()
    lin1, X1, y1 = leaves()
    loss = (y1 * nk.from_ndarray(tdev, g)).sum()
    loss.forward(); loss.backward(1.0)
    lin2, X2, y2 = leaves()
    G = nk.from_ndarray(tdev, g)
    y2.forward(); y2.backward_from(G)
    assert np.array_equal(X2.grad(), X1.grad()) and np.array_equal(lin2.weight.grad(), lin1.weight.grad())
    assert np.array_equal(lin2.bias.grad(), lin1.bias.grad())
    assert np.array_equal(G.data(), g)                         # the seed tensor is only read
    assert not y2.grad().any()                                 # the root gradient's own buffer is back (still zero)
    y2.no_grad(); y2.with_grad()                               # re-arm the intermediate gradients (never re-zeroed by backward)
    y2.backward_from(G)                                        # leaves accumulate (vardiff.rs:125-141 semantics)
    assert np.array_equal(X2.grad(), 2 * X1.grad())
    with pytest.raises(RuntimeError, match="shape"):
        y2.backward_from(nk.zeros(tdev, [24, 15]))


def test_rccl_single_rank_and_gradient_sync(nk, tdev):
    """Exercise the RCCL entry points on the GPU box (1 GPU => world of one): unique id,
    communicator, side-stream all-reduce ordered after a compute-stream event, join; and the
    overlapped GradientSync hook (grad_ready fires once per registered parameter, in reverse
    layer order, gradients unchanged by a sum over one rank)."""
    from neuronika_amd import capi
    cdev = capi.Device(handle=tdev.raw())
    uid = capi.Comm.unique_id()
    assert len(uid) == 128
    comm = capi.Comm(cdev, 1, 0, uid)
    x = np.arange(1 << 20, dtype=np.float32)
    X = cdev.array(x)
    capi.relu_fwd(cdev, X, X)                       # work on the compute stream first
    ev = cdev.event().record()
    comm.allreduce_sum_async(X, ev)
    comm.allreduce_sum_async(X, None)
    comm.join()
    assert np.array_equal(X.numpy(), x)
    comm.close()

    tcomm = nk.dp.Communicator(tdev, 1, 0, nk.dp.Communicator.unique_id())
    lins = [nk.nn.Linear(tdev, 32, 32, s) for s in (1, 2)]
    params = [p for l in lins for p in (l.weight, l.bias)]
    sync = nk.dp.GradientSync(tcomm, params)
    assert sync.bytes_per_step() == 2 * (32 * 32 + 32) * 4
    X = nk.rand(tdev, [16, 32], 3)
    loss = lins[1].forward(lins[0].forward(X).relu()).mse(nk.rand(tdev, [16, 32], 4), nk.Reduction.Mean)
    loss.forward(); loss.backward(1.0)
    want = [p.grad().copy() for p in params]
    for p in params:
        p.zero_grad()
    loss.no_grad(); loss.with_grad()
    loss.backward_sync(1.0, sync); sync.join()
    for p, w in zip(params, want):
        close(p.grad(), w, 1e-6, 1e-7)


@pytest.mark.gpu
def test_gradient_sync_piecewise_exchange(nk, tdev):
    """The C4-sized weight gradient is produced and handed to the exchange in two row blocks (only half of the last
    gradient stays exposed behind the backward pass); the bias gradients of the whole model travel as ONE group, sent
    as soon as the last of them is final.  With one rank the sum is the identity, so the gradients must equal those of
    a plain backward bit for bit, and the hook must have issued 2 pieces per layer + 1 group."""
    tcomm = nk.dp.Communicator(tdev, 1, 0, nk.dp.Communicator.unique_id())
    lins = [nk.nn.Linear(tdev, 4096, 4096, s) for s in (1, 3)]
    params = [p for l in lins for p in (l.weight, l.bias)]
    X, T = nk.rand(tdev, [256, 4096], 5), nk.rand(tdev, [256, 4096], 6)
    loss = lins[1].forward(lins[0].forward(X).relu()).mse(T, nk.Reduction.Mean)
    loss.forward(); loss.backward(0.5)
    want = [p.grad().copy() for p in params]
    sync = nk.dp.GradientSync(tcomm, params)
    sync.set_force_exchange(True)
    sync.set_parts("all")                                      # (the default splits only the gradient that was final last)
    total = sum(int(np.prod(p.shape)) for p in params)
    for rep in range(2):
        for p in params:
            p.zero_grad()
        loss.no_grad(); loss.with_grad()
        loss.backward_sync(0.5, sync); sync.join()
        assert sync.exchanges_issued() == 5 * (rep + 1)        # per layer: 2 halves of dW; + one group of both biases
        assert sync.elements_exchanged() == total * (rep + 1)
        for p, w in zip(params, want):
            assert np.array_equal(p.grad(), w)
    small = nk.nn.Linear(tdev, 64, 64, 9)                      # everything below the threshold: one group
    s2 = nk.dp.GradientSync(tcomm, [small.weight, small.bias]); s2.set_force_exchange(True)
    l2 = small.forward(nk.rand(tdev, [8, 64], 1)).sum()
    l2.forward(); l2.backward_sync(1.0, s2); s2.join()
    assert s2.exchanges_issued() == 1
    s3 = nk.dp.GradientSync(tcomm, [small.weight, small.bias], small_elems=0); s3.set_force_exchange(True)   # no grouping
    small.weight.zero_grad(); small.bias.zero_grad(); l2.no_grad(); l2.with_grad()
    l2.backward_sync(1.0, s3); s3.join()
    assert s3.exchanges_issued() == 2


def _replica_check(nk, tdev, ranks, build, seed=1.0, parts=None, channels=0, gbps=0.0):
    """Run `build()`'s loss once plainly and once through GradientSync over a replica communicator of `ranks` virtual
    ranks that all hold this rank's values: every element of every registered gradient must come back multiplied by
    `ranks` exactly once - an element skipped, sent twice or sent before its last writer ran shows up as a mismatch
    (a sum over ONE real rank is the identity and would hide all three)."""
    loss, params = build()
    loss.forward(); loss.backward(seed)
    want = [p.grad().copy() for p in params]
    comm = nk.dp.Communicator.replicas(tdev, ranks, channels, gbps)
    assert comm.size == ranks
    sync = nk.dp.GradientSync(comm, params)
    if parts is not None:
        sync.set_parts(parts)
    for rep in range(2):
        for p in params:
            p.zero_grad()
        loss.no_grad(); loss.with_grad()
        loss.backward_sync(seed, sync); sync.join()
        for p, w in zip(params, want):
            assert np.array_equal(p.grad(), w * np.float32(ranks)), p.shape
    assert sync.elements_exchanged() == 2 * sum(int(np.prod(p.shape)) for p in params)
    return sync


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,parts,launches", [(2, "all", 2 * (3 * 2 + 1)), (8, "all", 14), (8, "none", 2 * (3 + 1)), (8, "last", (3 + 1) + (4 + 1)), (2, None, 9)])
def test_gradient_sync_covers_every_element_once(nk, tdev, ranks, parts, launches):
    """`parts`: which weight gradients are handed over in two row blocks - all of them, only the one that was final last
    in the previous pass (the default: its exchange is the exposed one; the first pass splits nothing), or none."""
    def mlp():                                                 # C4-shaped: piecewise weight gradients + grouped biases
        lins = [nk.nn.Linear(tdev, 4096, 4096, s) for s in (1, 3, 5)]
        X, T = nk.rand(tdev, [128, 4096], 5), nk.rand(tdev, [128, 4096], 6)
        loss = lins[2].forward(lins[1].forward(lins[0].forward(X).relu()).relu()).mse(T, nk.Reduction.Mean)
        return loss, [p for l in lins for p in (l.weight, l.bias)]
    sync = _replica_check(nk, tdev, ranks, mlp, 1.0 / ranks, parts)
    assert sync.exchanges_issued() == launches

    def ragged():                                              # below the split threshold, odd sizes, a mid-sized weight
        l1, l2 = nk.nn.Linear(tdev, 300, 700, 1), nk.nn.Linear(tdev, 700, 129, 2)
        loss = l2.forward(l1.forward(nk.rand(tdev, [37, 300], 3)).relu()).sum()
        return loss, [l1.weight, l1.bias, l2.weight, l2.bias]
    _replica_check(nk, tdev, ranks, ragged, 1.0, parts)


@pytest.mark.gpu
def test_paced_replica_exchange_covers_every_element_once(nk, tdev):
    """The overlap projection's stand-in (`channels` workgroups pacing their pass to `gbps`, benchmarks/overlap_projection.py)
    is the same coverage-checked exchange: odd sizes, tails that are not whole float4s, buffers that are only 4-byte aligned
    (row-block pieces of a weight gradient with an odd row length fall back to the unpaced kernel)."""

    def ragged():
        l1, l2 = nk.nn.Linear(tdev, 301, 703, 1), nk.nn.Linear(tdev, 703, 129, 2)
        loss = l2.forward(l1.forward(nk.rand(tdev, [37, 301], 3)).relu()).sum()
        return loss, [l1.weight, l1.bias, l2.weight, l2.bias]
    _replica_check(nk, tdev, 4, ragged, channels=3, gbps=50.0)


@pytest.mark.gpu
def test_gradient_sync_covers_conv_and_attention_parameters(nk, tdev):
    """Parameters whose gradients come from nodes that do NOT hand over pieces early (convolution kernel / bias
    gradients, the attention module's eight parameters whose weight gradients are split-K GEMMs) are exchanged whole
    when their last writer has run - each element exactly once, also when a module is applied twice."""
    def conv_net():
        c1 = nk.nn.Conv2d(tdev, 8, 32, [3, 3], [1, 1], nk.PaddingMode.zero(), [1, 1], [1, 1], 1)
        c2 = nk.nn.Conv2d(tdev, 32, 32, [3, 3], [1, 1], nk.PaddingMode.zero(), [1, 1], [1, 1], 3)
        X = nk.rand(tdev, [4, 8, 12, 12], 5)
        y = c2.forward(c2.forward(c1.forward(X).relu()).relu())          # c2 applied twice: two writers of its gradients
        return y.sum(), [c1.weight, c1.bias, c2.weight, c2.bias]
    _replica_check(nk, tdev, 2, conv_net)

    def attention():
        mha = nk.nn.MultiheadAttention(tdev, 128, 4, 0.0, 11)
        X = nk.rand(tdev, [2 * 64, 128], 3).requires_grad()
        loss = mha.forward(X, 2).mse(nk.rand(tdev, [2 * 64, 128], 4), nk.Reduction.Mean)
        return loss, [getattr(getattr(mha, n), w) for n in "qkvo" for w in ("weight", "bias")]
    _replica_check(nk, tdev, 4, attention, 0.25)


@pytest.mark.gpu
def test_gradient_sync_shared_linear(nk, tdev):
    """A Linear applied twice (tied weights, an unrolled recurrence): its weight gradient has two writers on the tape.
    The piecewise hand-over must wait for the LAST writer - the first one keeps the single-GEMM path and the exchange
    happens once, after the second accumulation (it used to start on the first node's rows while the second node was
    still accumulating into the same buffer on the compute stream, and to reduce those rows twice)."""
    def tied():
        lin = nk.nn.Linear(tdev, 4096, 4096, 1)
        X = nk.rand(tdev, [128, 4096], 5)
        loss = lin.forward(lin.forward(X).relu()).mse(nk.rand(tdev, [128, 4096], 6), nk.Reduction.Mean)
        return loss, [lin.weight, lin.bias]
    sync = _replica_check(nk, tdev, 2, tied, 1.0, "all")
    assert sync.exchanges_issued() == 2 * (2 + 1)              # the last writer's two halves + the bias group, per pass
    sync = _replica_check(nk, tdev, 2, tied)                   # default policy: pass 1 whole, pass 2 the (only) weight gradient in halves
    assert sync.exchanges_issued() == (1 + 1) + (2 + 1)


@pytest.mark.gpu
def test_quickstart_example_trains(nk, tdev):
    """examples/quickstart.py (= the reference's examples/quickstart.rs, config C1) runs end to end - CSV loader,
    serde model, Linear/ReLU graph rebuilt per batch, MSE, SGD - and the epoch loss goes down."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("quickstart", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "quickstart.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    losses = mod.main(epochs=5, seed=1)
    assert len(losses) == 5 and all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.gpu
def test_free_constructors(nk, tdev):
    """lib.rs:160-240: eye / linspace / logspace / geomspace / range with ndarray's element formulas (f32)."""
    f = np.float32
    assert np.array_equal(nk.eye(tdev, 4).data(), np.eye(4, dtype=f))
    lin = lambda a, b, n: f(a) + (f(b) - f(a)) / f(n - 1) * np.arange(n, dtype=f)
    assert np.array_equal(nk.linspace(tdev, -4.0, 4.0, 9).data(), lin(-4, 4, 9))
    assert np.array_equal(nk.linspace(tdev, 2.0, 5.0, 1).data(), np.array([2.0], f))
    close(nk.logspace(tdev, 10.0, 0.0, 3.0, 4).data(), [1, 10, 100, 1000], 1e-6)
    close(nk.logspace(tdev, -2.0, 0.0, 3.0, 4).data(), [-1, -2, -4, -8], 1e-6)
    close(nk.geomspace(tdev, 1.0, 1000.0, 4).data(), [1, 10, 100, 1000], 2e-6)
    close(nk.geomspace(tdev, -1.0, -16.0, 5).data(), [-1, -2, -4, -8, -16], 2e-6)
    with pytest.raises(RuntimeError, match="geomspace"):
        nk.geomspace(tdev, -1.0, 4.0, 3)
    assert np.array_equal(nk.range(tdev, 0.0, 5.0, 1.0).data(), np.arange(5, dtype=f))
    assert np.array_equal(nk.range(tdev, 1.0, 2.0, 0.25).data(), np.array([1.0, 1.25, 1.5, 1.75], f))
    assert nk.range(tdev, 3.0, 1.0, 1.0).shape == [0]


@pytest.mark.gpu
def test_hipgraph_training_step(nk, tdev):
    """A launch-bound training step (quickstart-sized MLP) captured into a hipGraph and replayed gives bit-identical
    parameters to issuing the same steps eagerly."""
    def make():
        lins = [nk.nn.Linear(tdev, 3, 5, 1), nk.nn.Linear(tdev, 5, 5, 2), nk.nn.Linear(tdev, 5, 1, 3)]
        X, T = nk.rand(tdev, [64, 3], 7), nk.rand(tdev, [64, 1], 8)
        loss = lins[2].forward(lins[1].forward(lins[0].forward(X).relu()).relu()).mse(T, nk.Reduction.Mean)
        opt = nk.optim.SGD(0.05, momentum=0.9)
        params = [p for l in lins for p in (l.weight, l.bias)]
        for p in params:
            opt.register(p)

        def step():
            loss.forward()
            loss.no_grad(); loss.with_grad()
            loss.backward(1.0)
            opt.step()
            opt.zero_grad()
        return params, step, loss

    pe, step_e, loss_e = make()
    for _ in range(12):
        step_e()
    want = [p.data().copy() for p in pe]
    pg, step_g, loss_g = make()
    step_g(); step_g()                       # warm the allocator / workspace, reach the steady state
    tdev.graph_begin()
    step_g()
    graph = tdev.graph_end()                 # capturing records the step, it does not run it
    for _ in range(10):
        graph.launch()
    for p, w in zip(pg, want):
        assert np.array_equal(p.data(), w)
    assert np.isfinite(loss_g.item()) and loss_g.item() == loss_e.item()


@pytest.mark.gpu
def test_hipgraph_refuses_to_freeze_a_dropout_mask(nk, tdev):
    """A forward that draws a dropout mask bakes its Philox offset into the kernel arguments: a replayed graph would drop the
    same elements every step.  Training-mode dropout (the plain node, the attention probabilities, the fused attention core)
    refuses to be captured; evaluation mode and p = 0 capture and replay."""
    x = nk.rand(tdev, [64, 64], 3)
    for build in (lambda st: x.dropout(0.5, st),
                  lambda st: x.heads_attention(x, x, 1, 64, 1, 64, 0.125, 0.5, st)):
        st = nk.Status(True)
        y, other = build(st), x.relu()
        y.forward(); other.forward()
        tdev.graph_begin()
        other.forward()                    # (something to capture)
        with pytest.raises(RuntimeError, match="same mask"):
            y.forward()
        g = tdev.graph_end()
        del g
        st.set(False)                      # evaluation mode: nothing is drawn, the step captures
        y.forward(); want = y.data().copy()
        tdev.graph_begin(); y.forward(); g = tdev.graph_end()
        g.launch(); g.launch()
        assert np.array_equal(y.data(), want)
    # the attention-probabilities row kernel, through the C ABI
    from neuronika_amd import capi as c
    dev = c.Device(0)
    sc, out = dev.array(np.random.default_rng(0).random((64, 64), dtype=np.float32)), dev.zeros((64, 64))
    c.check(c.lib.nk_graph_begin(dev.h))
    with pytest.raises(RuntimeError, match="same mask"):
        c.scale_softmax_dropout_fwd(dev, sc, None, out, None, 0.125, 0.5, True, 1, 0)
    c.scale_softmax_dropout_fwd(dev, sc, None, out, None, 0.125, 0.5, False, 1, 0)   # evaluation mode captures
    import ctypes
    gh = ctypes.c_void_p()
    c.check(c.lib.nk_graph_end(dev.h, ctypes.byref(gh)))
    c.check(c.lib.nk_graph_launch(gh)); dev.sync()
    c.check(c.lib.nk_graph_destroy(gh))
    z = sc.numpy().astype(np.float64) * 0.125
    soft = np.exp(z - z.max(1, keepdims=True)); soft /= soft.sum(1, keepdims=True)
    np.testing.assert_allclose(out.numpy(), soft, rtol=2e-6)


@pytest.mark.gpu
def test_hipgraph_refuses_step_dependent_optimizers_and_keeps_workspaces(nk, tdev):
    """An Adam step bakes 1 - beta^step into its kernel arguments: capturing it would freeze the bias correction, so it
    refuses (the SGD step of the test above captures).  And a workspace outgrown AFTER a capture stays valid for the
    graph that has its address baked in: replaying the graph after a much larger reduction gives the same result."""
    lin = nk.nn.Linear(tdev, 8, 8, 1)
    X, T = nk.rand(tdev, [16, 8], 7), nk.rand(tdev, [16, 8], 8)
    loss = lin.forward(X).mse(T, nk.Reduction.Mean)
    adam = nk.optim.Adam(0.01)
    adam.register(lin.weight); adam.register(lin.bias)
    loss.forward(); loss.backward(1.0); adam.step(); adam.zero_grad()
    tdev.graph_begin()
    loss.forward(); loss.no_grad(); loss.with_grad(); loss.backward(1.0)
    with pytest.raises(RuntimeError, match="captured"):
        adam.step()
    g = tdev.graph_end()
    del g
    # workspace: capture a split-K GEMM (uses slabs in the device workspace), then force the workspace to grow
    a, b = nk.rand(tdev, [64, 8192], 1), nk.rand(tdev, [8192, 64], 2)
    c = a.mm(b)
    c.forward()
    want = c.data().copy()
    tdev.graph_begin(); c.forward(); graph = tdev.graph_end()
    big_a, big_b = nk.rand(tdev, [512, 65536], 3), nk.rand(tdev, [65536, 512], 4)     # 64 splits x 1 MB slabs > 64 MB
    big = big_a.mm(big_b); big.forward()
    for _ in range(3):
        graph.launch()
    assert np.array_equal(c.data(), want)


@pytest.mark.gpu
def test_mha_strided_heads_equals_split_merge(nk, tdev):
    """Attention GEMMs addressing the heads inside the (B*S, H*dh) projection layout (no split / merge copies) give
    bit-identical outputs and gradients to the Chunk/cat formulation."""
    B, S, d, H = 2, 64, 128, 4
    x, g = rnd(0, (B * S, d), -1, 1), rnd(5, (B * S, d), -1, 1)
    outs = []
    for strided in (True, False):
        mha = nk.nn.MultiheadAttention(tdev, d, H, 0.0, 11)
        mha.strided_heads, mha.fused_core = strided, False
        X = nk.from_ndarray(tdev, x).requires_grad()
        y = mha.forward(X, B)
        loss = (y * nk.from_ndarray(tdev, g)).sum()
        n_nodes = loss.history_len()
        loss.forward(); loss.backward(1.0)
        outs.append([y.data(), X.grad()] + [getattr(getattr(mha, n), w).grad() for n in "qkvo" for w in ("weight", "bias")] + [n_nodes])
    assert outs[0][-1] == outs[1][-1] - 4                      # 3 split + 1 merge nodes gone
    for a, b in zip(outs[0][:-1], outs[1][:-1]):
        assert np.array_equal(a, b)


def test_conv2d_module_built_folded_survives_a_knob_change(nk, tdev):
    """nn::Conv2d decides at graph-BUILD time to leave the Pad node out (nk_conv_padding_folds said yes under the rules then in force);
    the folded entries decide per CALL.  With the knob turned to "never Winograd" between build and forward() the node's entries fall
    back to the two nodes they stand for (Pad::forward into the device's operand scratch, node/pad/mod.rs:97-129, then the convolution
    entries, convolution/mod.rs:331-355): a valid graph stays valid, and its values are the BITS of the module built with
    `fold_padding = false` under the same knob - forward, input gradient, kernel and bias gradient; the launch counter says no
    Winograd kernel ran."""
    from neuronika_amd import capi
    cdev = capi.Device(handle=tdev.raw())
    N, Cin, Cout, H = 96, 64, 128, 28                                 # (large enough for the rules of all three passes)
    x, gy = rnd(0, (N, Cin, H, H)), rnd(2, (N, Cout, H, H))
    assert capi.conv_padding_folds(cdev, x.shape, (1, 1), (Cout, Cin, 3, 3), (1, 1), (1, 1), 1)
    conv = nk.nn.Conv2d(tdev, Cin, Cout, [3, 3], [1, 1], nk.PaddingMode.zero(), [1, 1], [1, 1], 1)
    X = nk.from_ndarray(tdev, x).requires_grad()
    y = conv.forward(X)                                               # built under the rule: no Pad node
    plain = nk.nn.Conv2d(tdev, Cin, Cout, [3, 3], [1, 1], nk.PaddingMode.zero(), [1, 1], [1, 1], 1)
    plain.fold_padding = False
    X2 = nk.from_ndarray(tdev, x).requires_grad()
    y2 = plain.forward(X2)
    cdev.conv_winograd(0)
    try:
        before = cdev.conv_winograd_launches()
        y.forward(); y.backward_from(nk.from_ndarray(tdev, gy))
        y2.forward(); y2.backward_from(nk.from_ndarray(tdev, gy))
        assert cdev.conv_winograd_launches() == before
    finally:
        cdev.conv_winograd(None)
    assert np.array_equal(conv.weight.data(), plain.weight.data())   # same seed, same parameters
    for a, b in ((y.data(), y2.data()), (X.grad(), X2.grad()), (conv.weight.grad(), plain.weight.grad()), (conv.bias.grad(), plain.bias.grad())):
        assert np.array_equal(a, b)
    # and back under the rule the same graph takes the Winograd kernels again (values: test_conv2d_module_at_a_size_... below)
    X.zero_grad(); conv.weight.zero_grad(); conv.bias.zero_grad()
    before = cdev.conv_winograd_launches()
    y.forward(); y.backward_from(nk.from_ndarray(tdev, gy))
    assert cdev.conv_winograd_launches() - before == 3


def test_conv2d_module_at_an_odd_extent_and_on_a_plain_input(nk, tdev):
    """(i) nn::Conv2d (3 x 3, pad 1) on 13 x 13 planes: since round 6 the Winograd kernels take odd output extents, so the rule folds the
    module's Zero padding here as well (no Pad node; three Winograd launches per step) - values equal to the module built with
    `fold_padding = false` bit for bit, and on integer-valued data to the oracle exactly.
    (ii) The same layer on a NON-differentiable input (`Var`: the data of a network's first layer): the reference builds no
    backward-input node for it (var.rs:1296-1371), and neither does the mirror - the step launches TWO convolution passes (forward,
    kernel gradient), not three; for a 7 x 7 / stride-2 stem that is the 3.8 ms direct input-gradient kernel that never runs."""
    from neuronika_amd import capi
    cdev = capi.Device(handle=tdev.raw())
    N, Cin, Cout, H = 352, 64, 128, 13                                # (enough tiles for the rules of all three passes)
    rng = np.random.default_rng(0)
    x = rng.integers(-3, 4, (N, Cin, H, H)).astype(np.float32)
    gy = rng.integers(-2, 3, (N, Cout, H, H)).astype(np.float32)
    assert capi.conv_padding_folds(cdev, x.shape, (1, 1), (Cout, Cin, 3, 3), (1, 1), (1, 1), 1)
    outs = []
    for fold in (True, False):
        conv = nk.nn.Conv2d(tdev, Cin, Cout, [3, 3], [1, 1], nk.PaddingMode.zero(), [1, 1], [1, 1], 1)
        conv.fold_padding = fold
        w = np.round(conv.weight.data() * 48).astype(np.float32)     # integer-valued parameters: every partial sum is exact in f32
        conv.weight.set_data(w); conv.bias.set_data(np.round(conv.bias.data() * 48).astype(np.float32))
        X = nk.from_ndarray(tdev, x).requires_grad()
        y = conv.forward(X)
        before = cdev.conv_winograd_launches()
        y.forward(); y.backward_from(nk.from_ndarray(tdev, gy))
        assert cdev.conv_winograd_launches() - before == 3
        outs.append([y.data(), X.grad(), conv.weight.grad(), conv.bias.grad(), conv.weight.data(), conv.bias.data()])
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    w, b = outs[0][4], outs[0][5]
    xp = np.zeros((N, Cin, H + 2, H + 2), np.float32); xp[:, :, 1:-1, 1:-1] = x
    yr = np.zeros((N, Cout, H, H), np.float32); O.convolution_forward(xp, w, yr, (1, 1), (1, 1), 1)
    assert np.array_equal(outs[0][0], yr + b)
    dxp = np.zeros_like(xp); O.convolution_backward_input(dxp, gy, w, (1, 1), (1, 1), 1)
    assert np.array_equal(outs[0][1], dxp[:, :, 1:-1, 1:-1])
    dw = np.zeros_like(w); O.convolution_backward_kernel(dw, gy, xp, (1, 1), (1, 1), 1)
    assert np.array_equal(outs[0][2], dw)
    # (ii)
    stem = nk.nn.Conv2d(tdev, 3, 64, [7, 7], [3, 3], nk.PaddingMode.zero(), [2, 2], [1, 1], 1)
    xs = rnd(1, (4, 3, 64, 64))
    for differentiable, passes in ((False, 2), (True, 3)):
        Xs = nk.from_ndarray(tdev, xs)
        if differentiable:
            Xs = Xs.requires_grad()
        ys = stem.forward(Xs)
        cdev.profile_begin()
        ys.forward(); ys.backward_from(nk.from_ndarray(tdev, rnd(2, (4, 64, 32, 32))))
        launches, _, _ = cdev.profile_end(capi.KERNEL_CONV)
        assert launches == passes, (differentiable, launches)
        stem.weight.zero_grad(); stem.bias.zero_grad()


def test_conv2d_module_stride_2_takes_the_fused_phase_input_gradient(nk, tdev):
    """nn::Conv2d (3 x 3, stride 2, pad 1) 64 -> 128 on 56 x 56 planes through the tape - a CNN's down-sampling layer: the input gradient
    lands in the caller's unpadded tensor through `nk_conv_bwd_input_padded`, which by rule is the fused-phase kernel (csrc/nk_conv_s2dx.h);
    forward, input, kernel and bias gradients against the oracle on integer-valued data, exactly; the knob at 0 gives the same bits."""
    from neuronika_amd import capi
    cdev = capi.Device(handle=tdev.raw())
    N, Cin, Cout, H = 16, 64, 128, 56
    rng = np.random.default_rng(0)
    x = rng.integers(-3, 4, (N, Cin, H, H)).astype(np.float32)
    gy = rng.integers(-2, 3, (N, Cout, H // 2, H // 2)).astype(np.float32)
    outs = []
    for mode in (None, 0):
        cdev.conv_s2dx(mode)
        try:
            conv = nk.nn.Conv2d(tdev, Cin, Cout, [3, 3], [1, 1], nk.PaddingMode.zero(), [2, 2], [1, 1], 1)
            conv.weight.set_data(np.round(conv.weight.data() * 48).astype(np.float32))
            conv.bias.set_data(np.round(conv.bias.data() * 48).astype(np.float32))
            X = nk.from_ndarray(tdev, x).requires_grad()
            y = conv.forward(X)
            y.forward(); y.backward_from(nk.from_ndarray(tdev, gy))
            outs.append([y.data(), X.grad(), conv.weight.grad(), conv.bias.grad(), conv.weight.data(), conv.bias.data()])
        finally:
            cdev.conv_s2dx(None)
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    w, b = outs[0][4], outs[0][5]
    xp = np.zeros((N, Cin, H + 2, H + 2), np.float32); xp[:, :, 1:-1, 1:-1] = x
    yr = np.zeros((N, Cout, H // 2, H // 2), np.float32); O.convolution_forward(xp, w, yr, (2, 2), (1, 1), 1)
    assert np.array_equal(outs[0][0], yr + b)
    dxp = np.zeros_like(xp); O.convolution_backward_input(dxp, gy, w, (2, 2), (1, 1), 1)
    assert np.array_equal(outs[0][1], dxp[:, :, 1:-1, 1:-1])
    dw = np.zeros_like(w); O.convolution_backward_kernel(dw, gy, xp, (2, 2), (1, 1), 1)
    assert np.array_equal(outs[0][2], dw)
    assert np.array_equal(outs[0][3].reshape(-1), gy.sum(axis=(0, 2, 3)))


def test_conv2d_module_at_a_size_the_winograd_rule_takes(nk, tdev):
    """nn::Conv2d (3 x 3, pad 1) through the tape at 48 x 64 x 56 x 56 -> 128 channels: by rule all three passes take the Winograd
    kernels (nk_conv_bias_fwd, nk_conv_bwd_input_padded, nk_conv_bwd_kernel_bias - the launch counter says so); the same step with the
    knob at 0 runs the implicit-GEMM kernels.  Both against f64 direct sums at sampled positions, and against each other inside the
    contraction bound (the two orders of summation differ)."""
    from neuronika_amd import capi
    cdev = capi.Device(handle=tdev.raw())
    N, Cin, Cout, H = 48, 64, 128, 56
    x, gy = rnd(0, (N, Cin, H, H)), rnd(2, (N, Cout, H, H))
    got = {}
    for mode in (None, 0):
        cdev.conv_winograd(mode)
        try:
            conv = nk.nn.Conv2d(tdev, Cin, Cout, [3, 3], [1, 1], nk.PaddingMode.zero(), [1, 1], [1, 1], 1)
            X = nk.from_ndarray(tdev, x).requires_grad()
            y = conv.forward(X)
            before = cdev.conv_winograd_launches()
            y.forward()
            y.backward_from(nk.from_ndarray(tdev, gy))
            took = cdev.conv_winograd_launches() - before
            got[mode] = (y.data(), X.grad(), conv.weight.grad(), conv.bias.grad(), conv.weight.data(), conv.bias.data(), took)
        finally:
            cdev.conv_winograd(None)
    assert got[None][6] == 3 and got[0][6] == 0                       # forward, input gradient, kernel gradient
    # the same module with the padded copy as a node of its own (fold_padding = false): the folded node computes the same bits
    conv = nk.nn.Conv2d(tdev, Cin, Cout, [3, 3], [1, 1], nk.PaddingMode.zero(), [1, 1], [1, 1], 1)
    conv.fold_padding = False
    X = nk.from_ndarray(tdev, x).requires_grad()
    y = conv.forward(X)
    y.forward(); y.backward_from(nk.from_ndarray(tdev, gy))
    for a, r in zip((y.data(), X.grad(), conv.weight.grad(), conv.bias.grad()), got[None][:4]):
        assert np.array_equal(a, r)
    w, b = got[None][4], got[None][5]
    assert np.array_equal(w, got[0][4]) and np.array_equal(b, got[0][5])   # same seed, same parameters
    import conv_samples as S
    from tolerance import assert_contraction
    xp = np.zeros((N, Cin, H + 2, H + 2), np.float32); xp[:, :, 1:-1, 1:-1] = x
    xmax, wmax, gmax = float(np.abs(x).max()), float(np.abs(w).max()), float(np.abs(gy).max())
    rng = np.random.default_rng(5)
    for mode in (None, 0):
        y, dx, dw, db = got[mode][:4]
        tag = "winograd" if mode is None else "implicit GEMM"
        idx, r64, r32 = S.forward(xp, w, b, rng, 24)
        assert_contraction(f"Conv2d module 48x64x56x56 ({tag}):y", y[idx], r64, Cin * 9, xmax, wmax, cpu32=r32, epilogue=True)
        idx, r64, r32 = S.input_gradient(gy, w, rng, 24, padded=False)
        assert_contraction(f"Conv2d module 48x64x56x56 ({tag}):dx", dx[idx], r64, Cout * 9, gmax, wmax, cpu32=r32)
        idx, r64, r32 = S.kernel_gradient(gy, xp, rng, 8)
        assert_contraction(f"Conv2d module 48x64x56x56 ({tag}):dw", dw[idx], r64, N * H * H, gmax, xmax, cpu32=r32)
        np.testing.assert_allclose(db.reshape(-1), gy.astype(np.float64).sum(axis=(0, 2, 3)), rtol=2e-6)
    assert not np.array_equal(got[None][0], got[0][0])                # two algorithms, two orders of summation
