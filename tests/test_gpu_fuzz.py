"""Seeded random expression graphs through the tape (broadcast binaries, pointwise nodes, reductions, matmul, softmax,
transpose, shared sub-expressions) against an f64 NumPy evaluation of the same expression; gradients against central
differences of that evaluation.  Exercises the combinations the per-node tests do not: fan-out (several writers into one
gradient: first-write then accumulate), un-broadcast reductions of every pattern, scalars, repeated backward passes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nk():
    import neuronika_amd
    return neuronika_amd.tape


@pytest.fixture(scope="module")
def tdev(nk):
    return nk.Device(0)


SHAPES = [(4, 5), (5,), (4, 1), (1, 5), (), (1,), (4, 5)]

UNARY = {
    "relu": (lambda v: v.relu(), lambda a: np.maximum(a, 0)),
    "sigmoid": (lambda v: v.sigmoid(), lambda a: 1 / (1 + np.exp(-a))),
    "tanh": (lambda v: v.tanh(), np.tanh),
    "softplus": (lambda v: v.softplus(), lambda a: np.log1p(np.exp(a))),
    "exp": (lambda v: v.exp(), np.exp),
    "neg": (lambda v: -v, lambda a: -a),
    "pow2": (lambda v: v.pow(2), lambda a: a ** 2),
    "sqrt_abs": (lambda v: (v.pow(2) + 1.0).sqrt(), lambda a: np.sqrt(a ** 2 + 1.0)),
    "ln_pos": (lambda v: (v.pow(2) + 1.0).ln(), lambda a: np.log(a ** 2 + 1.0)),
}
BINARY = {
    "add": (lambda a, b: a + b, lambda a, b: a + b),
    "sub": (lambda a, b: a - b, lambda a, b: a - b),
    "mul": (lambda a, b: a * b, lambda a, b: a * b),
    "div": (lambda a, b: a / (b.pow(2) + 1.0), lambda a, b: a / (b ** 2 + 1.0)),
}


def build(rng, nk, tdev, leaves_np):
    """Returns (device expression builder result, numpy evaluator) for one random program."""
    n_ops = int(rng.integers(4, 9))
    prog = []
    n_vals = len(leaves_np)
    shapes = [a.shape for a in leaves_np]
    for _ in range(n_ops):
        kind = rng.choice(["un", "bin", "bin", "mm", "softmax", "t", "mean_keep"])
        if kind == "un":
            i = int(rng.integers(0, n_vals)); op = str(rng.choice(list(UNARY)))
            prog.append(("un", op, i)); shapes.append(shapes[i])
        elif kind == "bin":
            i, j = int(rng.integers(0, n_vals)), int(rng.integers(0, n_vals)); op = str(rng.choice(list(BINARY)))
            try:
                out = np.broadcast_shapes(shapes[i], shapes[j])
            except ValueError:
                continue
            prog.append(("bin", op, i, j)); shapes.append(out)
        elif kind == "mm":
            cands = [(i, j) for i in range(n_vals) for j in range(n_vals)
                     if len(shapes[i]) == 2 and len(shapes[j]) == 2 and shapes[i][1] == shapes[j][1] and min(shapes[i] + shapes[j]) > 0]
            if not cands:
                continue
            i, j = cands[int(rng.integers(0, len(cands)))]
            prog.append(("mm_t", i, j)); shapes.append((shapes[i][0], shapes[j][0]))
        elif kind == "softmax":
            cands = [i for i in range(n_vals) if len(shapes[i]) == 2]
            if not cands:
                continue
            i = cands[int(rng.integers(0, len(cands)))]; ax = int(rng.integers(0, 2)); log = bool(rng.integers(0, 2))
            prog.append(("softmax", i, ax, log)); shapes.append(shapes[i])
        elif kind == "t":
            cands = [i for i in range(n_vals) if len(shapes[i]) == 2]
            if not cands:
                continue
            i = cands[int(rng.integers(0, len(cands)))]
            prog.append(("t", i)); shapes.append(shapes[i][::-1])
        else:
            i = int(rng.integers(0, n_vals))
            prog.append(("mean_bcast", i)); shapes.append(shapes[i])
        n_vals += 1
    return prog


def run_np(prog, leaves):
    vals = list(leaves)
    for ins in prog:
        if ins[0] == "un":
            vals.append(UNARY[ins[1]][1](vals[ins[2]]))
        elif ins[0] == "bin":
            vals.append(BINARY[ins[1]][1](vals[ins[2]], vals[ins[3]]))
        elif ins[0] == "mm_t":
            vals.append(vals[ins[1]] @ vals[ins[2]].T)
        elif ins[0] == "softmax":
            a = vals[ins[1]]; m = a.max(axis=ins[2], keepdims=True); e = np.exp(a - m); s = e.sum(axis=ins[2], keepdims=True)
            vals.append(a - m - np.log(s) if ins[3] else e / s)
        elif ins[0] == "t":
            vals.append(vals[ins[1]].T)
        else:
            vals.append(vals[ins[1]] - vals[ins[1]].mean())       # x - mean(x): scalar broadcast back
    total = 0.0
    for k, v in enumerate(vals[len(leaves):]):
        total = total + (k + 1) * 0.25 * np.sum(v)
    return total


def run_dev(nk, prog, leaves):
    vals = list(leaves)
    for ins in prog:
        if ins[0] == "un":
            vals.append(UNARY[ins[1]][0](vals[ins[2]]))
        elif ins[0] == "bin":
            vals.append(BINARY[ins[1]][0](vals[ins[2]], vals[ins[3]]))
        elif ins[0] == "mm_t":
            vals.append(vals[ins[1]].mm_t(vals[ins[2]]))
        elif ins[0] == "softmax":
            vals.append(vals[ins[1]].log_softmax(ins[2]) if ins[3] else vals[ins[1]].softmax(ins[2]))
        elif ins[0] == "t":
            vals.append(vals[ins[1]].t())
        else:
            vals.append(vals[ins[1]] - vals[ins[1]].mean())
    total = None
    for k, v in enumerate(vals[len(leaves):]):
        term = v.sum() * float((k + 1) * 0.25)
        total = term if total is None else total + term
    return total


@pytest.mark.parametrize("seed", range(24))
def test_random_graph(nk, tdev, seed):
    rng = np.random.default_rng(1000 + seed)
    leaves_np = [rng.uniform(-1.5, 1.5, s).astype(np.float32) for s in SHAPES[: int(rng.integers(3, len(SHAPES) + 1))]]
    prog = build(rng, nk, tdev, leaves_np)
    if not prog:
        pytest.skip("empty program")
    leaves = [nk.from_ndarray(tdev, a).requires_grad() for a in leaves_np]
    out = run_dev(nk, prog, leaves)
    out.forward()
    l64 = [a.astype(np.float64) for a in leaves_np]
    want = run_np(prog, l64)
    np.testing.assert_allclose(out.item(), want, rtol=2e-4, atol=2e-4)
    out.backward(1.0)
    for li, (leaf, a) in enumerate(zip(leaves, l64)):
        num = np.zeros_like(a)
        it = np.nditer(a, flags=["multi_index"]) if a.ndim else [None]
        for _ in it:
            idx = it.multi_index if a.ndim else ()
            old = a[idx]
            a[idx] = old + 1e-5; fp = run_np(prog, l64)
            a[idx] = old - 1e-5; fm = run_np(prog, l64)
            a[idx] = old
            num[idx] = (fp - fm) / 2e-5
        got = np.asarray(leaf.grad(), np.float64).reshape(a.shape)
        np.testing.assert_allclose(got, num, rtol=3e-3, atol=3e-3, err_msg=f"leaf {li} of seed {seed}: {prog}")
    # a second backward without zero_grad doubles the leaf gradients only if intermediates are re-armed
    g1 = [np.asarray(l.grad()).copy() for l in leaves]
    for l in leaves:
        l.zero_grad()
    out.no_grad(); out.with_grad()
    out.backward(1.0)
    for l, g in zip(leaves, g1):
        np.testing.assert_allclose(np.asarray(l.grad()), g, rtol=1e-6, atol=1e-6)


def test_partial_coverage_and_fanout(nk, tdev):
    """Nodes whose backward covers only part of a gradient (Chunk) or that feed one gradient from several nodes must
    leave exact zeros / exact sums behind the lazily zeroed buffers."""
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (6, 8)).astype(np.float32)
    X = nk.from_ndarray(tdev, x).requires_grad()
    parts = X.chunks([3, 4])                                  # 4 tiles; only tiles 1 and 2 are used
    y = (parts[1] * 2.0).sum() + (parts[2].t().t() * 3.0).sum()
    y.forward(); y.backward(1.0)
    want = np.zeros_like(x); want[0:3, 4:8] = 2.0; want[3:6, 0:4] = 3.0
    assert np.array_equal(X.grad(), want)
    # one leaf feeding: a padded conv input, a stack, a cat, a transpose and a plain product
    w = rng.uniform(-1, 1, (2, 1, 3, 3)).astype(np.float32)
    img = rng.uniform(-1, 1, (2, 1, 6, 8)).astype(np.float32)
    I, W = nk.from_ndarray(tdev, img).requires_grad(), nk.from_ndarray(tdev, w).requires_grad()
    conv = W.convolution(I.pad([1, 1], 0.0), [1, 1], [1, 1], 1).sum()
    stk = I.stack([I], 0).sum() * 0.5
    cat = I.cat([I * 2.0], 1).sum() * 0.25
    tot = conv + stk + cat + (I * I).sum()
    tot.forward(); tot.backward(1.0)
    i64, w64 = img.astype(np.float64), w.astype(np.float64)
    g = np.zeros_like(i64)
    pad = np.pad(i64, ((0, 0), (0, 0), (1, 1), (1, 1)))
    gp = np.zeros_like(pad)
    for co in range(2):
        for a in range(3):
            for b in range(3):
                gp[:, 0, a:a + 6, b:b + 8] += w64[co, 0, a, b]
    g += gp[:, :, 1:-1, 1:-1]
    g += 0.5 * 2 + 0.25 * (1 + 2) + 2 * i64
    np.testing.assert_allclose(I.grad(), g, rtol=1e-5, atol=1e-5)
    gw = np.zeros_like(w64)
    for a in range(3):
        for b in range(3):
            gw[:, 0, a, b] = pad[:, 0, a:a + 6, b:b + 8].sum()
    np.testing.assert_allclose(W.grad(), gw, rtol=1e-5, atol=1e-4)
