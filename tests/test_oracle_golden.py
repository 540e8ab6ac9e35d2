"""Pin the CPU oracle against the reference's own known-answer vectors.

Every expected number here comes from tests/golden/reference_fixtures.json, which
tests/golden/extract_reference_fixtures.py parsed out of the reference's Rust unit tests;
the `cite` field of each fixture names the reference file:line."""
import numpy as np
import pytest

from oracle import neuronika_oracle as O

F16_EPSILON = 4.88e-4  # reference tolerance, neuronika-variable/src/utils.rs:500


def f32(v, shape=None):
    a = np.asarray(v, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


def close(a, b, tol=F16_EPSILON):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=0, atol=tol)


# ------------------------------------------------------------------ convolution (exact)
CONV = ["conv1d", "conv2d", "conv3d", "conv1d_strided", "conv2d_strided", "conv3d_strided",
        "conv1d_dilated", "conv2d_dilated", "conv3d_dilated",
        "grouped_conv1d", "grouped_conv2d", "grouped_conv3d"]


def conv_case(golden, name):
    c = next(c for c in golden["convolution"] if c["name"] == name)
    x = np.arange(c["input"]["arange"], dtype=np.float32).reshape(c["input"]["shape"])
    w = np.ones(c["kernel"]["ones"], dtype=np.float32)
    return c, x, w


@pytest.mark.parametrize("name", CONV)
def test_conv_golden(golden, name):
    c, x, w = conv_case(golden, name)
    s, d, g = c["stride"], c["dilation"], c["groups"]
    O.check_conv_args(x.shape, (w.shape[0], w.shape[1] * g, *w.shape[2:]), s, d)
    O.check_groups_args(x.shape, w.shape, g)
    oshape = O.conv_out_shape(x.shape, w.shape, s, d)
    y = np.zeros(oshape, dtype=np.float32)
    O.convolution_forward(x, w, y, s, d, g)
    assert np.array_equal(y, f32(c["output"], oshape)), c["cite"]
    dx, dw = np.zeros_like(x), np.zeros_like(w)
    go = np.ones(oshape, dtype=np.float32)
    O.convolution_backward_input(dx, go, w, s, d, g)
    O.convolution_backward_kernel(dw, go, x, s, d, g)
    assert np.array_equal(dx, f32(c["input_grad"], x.shape)), c["cite"]
    assert np.array_equal(dw, f32(c["kernel_grad"], w.shape)), c["cite"]
    # backward accumulates (`+=`): a second call doubles the gradients
    O.convolution_backward_input(dx, go, w, s, d, g)
    O.convolution_backward_kernel(dw, go, x, s, d, g)
    assert np.array_equal(dx, 2 * f32(c["input_grad"], x.shape))
    assert np.array_equal(dw, 2 * f32(c["kernel_grad"], w.shape))


def test_im2col_layout(golden):
    c = golden["im2col"]
    img = f32(c["image"], c["image_shape"])
    x = np.stack([img, img])  # (2, 3, 4, 4)
    cols = O.im2col(x, c["kernel_shape"], c["stride"], c["dilation"])
    want = f32(c["im2col_T"], c["im2col_T_shape"]).T  # (L=4, K=27)
    assert cols.shape == (2, 4, 27)
    assert np.array_equal(cols[0], want) and np.array_equal(cols[1], want)


def test_conv_arg_checks():
    # convolution/test.rs:118-142
    O.check_conv_args((1, 2, 4, 4), (1, 2, 2, 2), (1, 1), (1, 1))
    with pytest.raises(AssertionError, match=r"Invalid kernel shape \[1, 2, 2\] for 2d conv"):
        O.check_conv_args((1, 2, 4, 4), (1, 2, 2), (1, 1), (1, 1))
    O.check_groups_args((3, 3, 10, 10), (3, 3, 3, 3), 3)
    with pytest.raises(AssertionError):
        O.check_groups_args((3, 3, 10, 10), (3, 3, 3, 3), 5)


# ------------------------------------------------------------------ matmul
def test_mm_backward_golden(golden):
    c = golden["nodes"]["mm_backward"]
    l = c["left_bwd"]
    right = np.linspace(*l["right"]["linspace"][:2], int(l["right"]["linspace"][2]), dtype=np.float32).reshape(3, 3)
    g = np.ones((3, 3), np.float32)
    da = np.zeros((3, 3), np.float32)
    O.mm_backward_left(da, g, right)
    close(da, f32(l["left_grad_once"], (3, 3)))
    O.mm_backward_left(da, g, right)
    close(da, f32(l["left_grad_twice"], (3, 3)))
    r = c["right_bwd"]
    left = np.linspace(1.0, 9.0, 9, dtype=np.float32).reshape(3, 3)
    db = np.zeros((3, 3), np.float32)
    O.mm_backward_right(db, g, left)
    close(db, f32(r["right_grad_once"], (3, 3)))
    O.mm_backward_right(db, g, left)
    close(db, f32(r["right_grad_twice"], (3, 3)))


def test_mm_t_golden(golden):
    c = golden["nodes"]["mm_t_forward"]
    a, b = f32(c["left"], c["left_shape"]), f32(c["right"], c["right_shape"])
    out = np.zeros(c["out_shape"], np.float32)
    O.mm_t_forward(a, b, out)
    close(out, f32(c["out"], c["out_shape"]))
    lit = golden["nodes"]["mm_t_backward"]["literals"]
    a, b, g = f32(lit[2], (3, 3)), f32(lit[3], (2, 3)), f32(lit[4], (3, 2))
    da, db = f32(lit[0], (3, 3)).copy(), f32(lit[1], (2, 3)).copy()
    O.mm_t_backward_left(da, g, b); O.mm_t_backward_right(db, g, a)
    close(da, f32(lit[6], (3, 3))); close(db, f32(lit[7], (2, 3)))
    O.mm_t_backward_left(da, g, b); O.mm_t_backward_right(db, g, a)
    close(da, f32(lit[8], (3, 3))); close(db, f32(lit[9], (2, 3)))


# ------------------------------------------------------------------ softmax / log-softmax
@pytest.mark.parametrize("op", ["softmax", "logsoftmax"])
@pytest.mark.parametrize("which", ["rows", "columns"])
def test_softmax_golden(golden, op, which):
    fwd = O.softmax_forward if op == "softmax" else O.log_softmax_forward
    bwd = O.softmax_backward if op == "softmax" else O.log_softmax_backward
    c = golden["nodes"][f"{op}_forward_{which}"]
    x = f32(c["input"], c["shape"])
    y = np.zeros_like(x)
    fwd(x, y, c["axis"])
    close(y, f32(c["out"], c["shape"]))
    b = golden["nodes"][f"{op}_backward_{which}"]
    lit = b["literals"]
    xin, g = f32(lit[1], (3, 3)), f32(lit[2], (3, 3))
    data = np.zeros_like(xin)
    fwd(xin, data, b["axis"])
    dx = f32(lit[0], (3, 3)).copy()
    bwd(dx, g, data, b["axis"])
    close(dx, f32(lit[4], (3, 3)))
    bwd(dx, g, data, b["axis"])
    close(dx, f32(lit[5], (3, 3)))


# ------------------------------------------------------------------ relu / sum / mean / mse / transpose
def test_relu_golden(golden):
    lit = golden["nodes"]["relu_forward"]["literals"]
    x = f32(lit[0], (3, 3)); y = np.zeros_like(x)
    O.relu_forward(x, y)
    assert np.array_equal(y, f32(lit[1], (3, 3)))
    lit = golden["nodes"]["relu_backward"]["literals"]
    dx, xin, g = f32(lit[0]).copy(), f32(lit[1]), f32(lit[2])
    O.relu_backward(dx, g, xin); assert np.array_equal(dx, f32(lit[4]))
    O.relu_backward(dx, g, xin); assert np.array_equal(dx, f32(lit[5]))


def test_sum_mean_golden(golden):
    n = golden["nodes"]
    x = f32(n["sum_forward"]["literals"][0], (3, 3))
    out = np.zeros((), np.float32)
    O.sum_forward(x, out); close(out, n["sum_forward"]["scalars"][0])
    x = f32(n["mean_forward"]["literals"][0], (3, 3))
    O.mean_forward(x, out); close(out, n["mean_forward"]["scalars"][0])
    for op, bwd in (("sum", O.sum_backward), ("mean", O.mean_backward)):
        lit = n[f"{op}_backward"]["literals"]
        dx = f32(lit[0], (10, 10)).copy()
        g = np.float32(n[f"{op}_backward"]["scalars"][0])
        bwd(dx, g); close(dx, f32(lit[1], (10, 10)))
        bwd(dx, g); close(dx, f32(lit[2], (10, 10)))


@pytest.mark.parametrize("red", ["mean", "sum"])
def test_squared_error_golden(golden, red):
    c = golden["nodes"][f"squared_error_{red}"]
    lit = c["literals"]
    t, x = f32(lit[0], (3, 3)), f32(lit[1], (3, 3))
    out = np.zeros((), np.float32)
    O.squared_error_forward(x, t, out, red)
    close(out, c["scalars"][0])
    dx = f32(lit[2], (3, 3)).copy()
    g = np.float32(c["scalars"][1])
    O.squared_error_backward(dx, g, x, t, red); close(dx, f32(lit[3], (3, 3)))
    O.squared_error_backward(dx, g, x, t, red); close(dx, f32(lit[4], (3, 3)))


def test_transpose_golden(golden):
    lit = golden["nodes"]["transpose_forward"]["literals"]
    x = f32(lit[0], (3, 3)); y = np.zeros_like(x)
    O.transpose_forward(x, y)
    assert np.array_equal(y, f32(lit[1], (3, 3)))


# ------------------------------------------------------------------ pointwise unary nodes (row f-2)
UNARY_FIX = {"neg": "negation", "sqrt": "sqrt", "sigmoid": "sigmoid", "tanh": "tanh", "softplus": "softplus",
             "leaky_relu": "leaky_relu", "pow": "power"}


@pytest.mark.parametrize("op", list(UNARY_FIX))
def test_unary_golden(golden, op):
    n = golden["nodes"]
    fw = n[f"{UNARY_FIX[op]}_forward"]
    e = fw["exp"][0] if fw["exp"] else 0
    x = f32(fw["literals"][0], (3, 3)); y = np.zeros_like(x)
    O.unary_forward(op, x, y, e)
    close(y, f32(fw["literals"][1], (3, 3)))
    cases = [n[f"{UNARY_FIX[op]}_backward"]] + ([n["power_backward_negative_exp"]] if op == "pow" else [])
    for bw in cases:
        lit = bw["literals"]
        e = bw["exp"][0] if bw["exp"] else 0
        if op == "neg":
            dx, g = f32(lit[0]).copy(), f32(lit[1])
            O.unary_backward(op, dx, g, None); close(dx, f32(lit[3]))
            O.unary_backward(op, dx, g, None); close(dx, f32(lit[4]))
            continue
        dx, xin, g = f32(lit[0]).copy(), f32(lit[1]), f32(lit[2])
        ref = xin
        if op in ("sigmoid", "tanh"):       # their tests run the forward node; sqrt's hands over sqrt(x) itself
            ref = np.zeros_like(xin); O.unary_forward(op, xin, ref, e)
        O.unary_backward(op, dx, g, ref, e); close(dx, f32(lit[4]))
        O.unary_backward(op, dx, g, ref, e); close(dx, f32(lit[5]))


def test_exp_ln_golden():
    """exp/test.rs:22-36,60-74 and logn/test.rs (enabled, new API): linspace(-4,4,9) inputs,
    expectations computed by ndarray itself (`mapv(f32::exp)`), backward = data / data*2."""
    x = np.linspace(-4.0, 4.0, 9, dtype=np.float32).reshape(3, 3)
    y = np.zeros_like(x); O.unary_forward("exp", x, y); close(y, np.exp(x))
    dx = np.zeros_like(x); g = np.ones_like(x)
    O.unary_backward("exp", dx, g, y); close(dx, y)
    O.unary_backward("exp", dx, g, y); close(dx, 2 * y)
    xp = np.linspace(1.0, 9.0, 9, dtype=np.float32).reshape(3, 3)
    O.unary_forward("ln", xp, y); close(y, np.log(xp))
    dx[...] = 0; O.unary_backward("ln", dx, g, xp); close(dx, 1 / xp)


# ------------------------------------------------------------------ loss criteria (row f-4)
def _lin(spec):
    a, b, n, *shape = spec
    return np.linspace(a, b, int(n), dtype=np.float32).reshape(shape)


@pytest.mark.parametrize("red", ["mean", "sum"])
def test_bce_mae_golden(golden, red):
    """bce/test.rs and absolute_error/test.rs (enabled): forward scalars and backward gradients."""
    n = golden["nodes"]
    for name, fwd, bwd in (("bce", O.bce_forward, O.bce_backward), ("absolute_error", O.mae_forward, O.mae_backward)):
        c = n[f"{name}_forward_base_case_{red}"]
        x, t = (_lin(sp) for sp in c["linspace_start_stop_n_shape"])
        close(fwd(x, t, red), c["scalars"][-1], c["tol"] * max(1.0, abs(c["scalars"][-1]) * 1e-3))
        c = n[f"{name}_backward_base_case_{red}"]
        x, t = (_lin(sp) for sp in c["linspace_start_stop_n_shape"])
        dx = np.zeros_like(x)
        bwd(dx, c["scalars"][0], x, t, red)
        want = f32(c["literals"][0], (3, 3)) if c["literals"] else np.full((3, 3), c["from_elem"][0][0], np.float32)
        np.testing.assert_allclose(dx, want, rtol=2e-6, atol=c["tol"])


@pytest.mark.parametrize("red", ["mean", "sum"])
def test_nll_kldiv_bcelogits_golden(golden, red):
    """nll / kldiv / bce_with_logits test.rs (modules commented out; numbers valid): pin the documented
    NLL layout and the masked KLDiv."""
    n = golden["nodes"]
    c = n[f"nll_{red}"]; lit = c["literals"]
    t, x = f32(lit[0]), f32(lit[1], (3, 5))
    lx = np.zeros_like(x); O.log_softmax_forward(x, lx, 1)
    close(O.nll_forward(lx, t, red), c["scalars"][0])
    dx = f32(lit[2], (3, 5)).copy()
    O.nll_backward(dx, c["scalars"][1], t, red); close(dx, f32(lit[3], (3, 5)))
    O.nll_backward(dx, c["scalars"][1], t, red); close(dx, 2 * f32(lit[4], (3, 5)))

    c = n[f"kldiv_{red}"]; lit = c["literals"]
    t, x = f32(lit[0], (2, 3)), np.log(f32(lit[1], (2, 3)))
    close(O.kldiv_forward(x, t, red), c["scalars"][0])
    dx = f32(lit[2], (2, 3)).copy()
    O.kldiv_backward(dx, c["scalars"][1], t, red); close(dx, f32(lit[3], (2, 3)))
    O.kldiv_backward(dx, c["scalars"][1], t, red); close(dx, 2 * f32(lit[4], (2, 3)))

    c = n[f"bce_with_logits_{red}"]; lit = c["literals"]
    t, x = f32(lit[0], (3, 3)), f32(lit[1], (3, 3))
    close(O.bce_with_logits_forward(x, t, red), c["scalars"][0], 1e-3)
    dx = f32(lit[2], (3, 3)).copy()
    O.bce_with_logits_backward(dx, c["scalars"][1], x, t, red); close(dx, f32(lit[3], (3, 3)))
    O.bce_with_logits_backward(dx, c["scalars"][1], x, t, red); close(dx, 2 * f32(lit[4], (3, 3)))


def test_nll_target_cast():
    """`target as usize` (nll/mod.rs:57): NaN / negative -> class 0, fraction dropped, >= C selects nothing."""
    x = np.log(np.array([[0.2, 0.8], [0.5, 0.5], [0.9, 0.1], [0.3, 0.7]], np.float32))
    t = np.array([-3.0, 1.9, np.nan, 7.0], np.float32)
    close(O.nll_forward(x, t, "sum"), -(x[0, 0] + x[1, 1] + x[2, 0]))


# ------------------------------------------------------------------ GEMV / dot (row f-4)
def test_gemv_dot_golden(golden):
    n = golden["nodes"]
    lit = n["matrix_vector_mul_forward"]["literals"]
    y = np.zeros(3, np.float32); O.mv_forward(f32(lit[0], (3, 3)), f32(lit[1]), y); close(y, f32(lit[2]))
    O.mv_forward(f32(lit[0], (3, 3)), f32(lit[3]), y); close(y, f32(lit[6]))
    lit = n["matrix_vector_mul_backward"]["literals"]
    dA, dx, A, x, g = f32(lit[0], (3, 3)).copy(), f32(lit[1]).copy(), f32(lit[2], (3, 3)), f32(lit[3]), f32(lit[4])
    O.mv_backward(dA, dx, g, A, x); close(dA, f32(lit[6], (3, 3))); close(dx, f32(lit[7]))
    O.mv_backward(dA, dx, g, A, x); close(dA, f32(lit[8], (3, 3))); close(dx, f32(lit[9]))
    lit = n["vector_matrix_mul_forward"]["literals"]
    O.vm_forward(f32(lit[0]), f32(lit[1], (3, 3)), y); close(y, f32(lit[2]))
    lit = n["vector_matrix_mul_backward"]["literals"]
    dv, dB, v, B, g = f32(lit[0]).copy(), f32(lit[1], (3, 3)).copy(), f32(lit[2]), f32(lit[3], (3, 3)), f32(lit[4])
    O.vm_backward(dv, dB, g, v, B); close(dv, f32(lit[6])); close(dB, f32(lit[7], (3, 3)))
    O.vm_backward(dv, dB, g, v, B); close(dv, f32(lit[8])); close(dB, f32(lit[9], (3, 3)))
    c = n["vector_vector_mul_forward"]
    close(O.vv_forward(f32(c["literals"][0]), f32(c["literals"][1])), c["scalars"][0])
    c = n["vector_vector_mul_backward"]; lit = c["literals"]
    dl, dr, l, r = f32(lit[0]).copy(), f32(lit[1]).copy(), f32(lit[2]), f32(lit[3])
    O.vv_backward(dl, r, c["scalars"][0]); O.vv_backward(dr, l, c["scalars"][0])
    close(dl, f32(lit[4])); close(dr, f32(lit[5]))
    O.vv_backward(dl, r, c["scalars"][0]); O.vv_backward(dr, l, c["scalars"][0])
    close(dl, f32(lit[6])); close(dr, f32(lit[7]))


# ------------------------------------------------------------------ broadcast binaries
OPS = {"addition": "add", "subtraction": "sub", "multiplication": "mul", "division": "div"}
NP = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide}


@pytest.mark.parametrize("name", list(OPS))
def test_binary_forward_broadcast(golden, name):
    op = OPS[name]
    for fn in ("left_broadcast", "right_broadcast"):
        c = golden["nodes"][f"{name}_forward_{fn}"]
        a0, a1, n, *shape = c["linspace_start_stop_n_shape"][0]
        small = np.linspace(a0, a1, int(n), dtype=np.float32).reshape(shape)   # (1, 3)
        big = np.ones(c["constructors"][0]["shape"], np.float32)                # (2, 2, 3)
        l, r = (small, big) if fn == "left_broadcast" else (big, small)
        assert O.cobroadcast(l.shape, r.shape) == (2, 2, 3)
        out = np.zeros((2, 2, 3), np.float32)
        O.binary_forward(op, l, r, out)
        close(out, NP[op](l, r))


def _expected_pair(c):
    # the two `from_elem(3, v)` expectations (after one and two backward() calls) are the
    # last two rank-1 constructors with non-trivial values of the test body
    vals = [k["value"] for k in c["constructors"] if k["shape"] == [3]]
    return vals[-2], vals[-1]


def test_binary_backward_reduction_golden(golden):
    n = golden["nodes"]
    g = np.ones((3, 3), np.float32)
    # addition/test.rs:110-124,158-172  (3,3) -> (3): 3 then 6
    for side, fn in (("left", O.binary_backward_left), ("right", O.binary_backward_right)):
        once, twice = _expected_pair(n[f"addition_backward_{side}_reduction"])
        d = np.zeros(3, np.float32)
        fn("add", d, g, None, None); close(d, np.full(3, once))
        fn("add", d, g, None, None); close(d, np.full(3, twice))
    # subtraction: left 3/6, right -3/-6
    once, twice = _expected_pair(n["subtraction_backward_right_reduction"])
    d = np.zeros(3, np.float32)
    O.binary_backward_right("sub", d, g, None, None); close(d, np.full(3, once))
    O.binary_backward_right("sub", d, g, None, None); close(d, np.full(3, twice))
    assert once < 0
    # multiplication/test.rs:121-139: right_data = 5 -> 15 / 30
    c = n["multiplication_backward_left_reduction"]
    once, twice = _expected_pair(c)
    other = np.full((3, 3), c["constructors"][0]["value"], np.float32)
    d = np.zeros(3, np.float32)
    O.binary_backward_left("mul", d, g, None, other); close(d, np.full(3, once))
    O.binary_backward_left("mul", d, g, None, other); close(d, np.full(3, twice))
    # division/test.rs:121-139: g / 5 summed over 3 rows -> 0.6 / 1.2
    c = n["division_backward_left_reduction"]
    once, twice = _expected_pair(c)
    other = np.full((3, 3), c["constructors"][0]["value"], np.float32)
    d = np.zeros(3, np.float32)
    O.binary_backward_left("div", d, g, None, other); close(d, np.full(3, once))
    O.binary_backward_left("div", d, g, None, other); close(d, np.full(3, twice))
    # division/test.rs:188-207: left = 3 (3,3), right = 5 (3) -> -g*l/r^2 summed: -0.36 / -0.72
    c = n["division_backward_right_reduction"]
    once, twice = _expected_pair(c)
    left = np.full((3, 3), c["constructors"][1]["value"], np.float32)
    right = np.full(3, c["constructors"][2]["value"], np.float32)
    d = np.zeros(3, np.float32)
    O.binary_backward_right("div", d, g, left, right); close(d, np.full(3, once))
    O.binary_backward_right("div", d, g, left, right); close(d, np.full(3, twice))


def test_accumulate_intended_semantics():
    """The reference's `accumulate` is defective for non-square shapes; the
    oracle implements NumPy un-broadcast.  Pin the intended behaviour on non-square shapes."""
    rng = np.random.default_rng(0)
    src = rng.random((5, 7), dtype=np.float32)
    t = np.zeros(7, np.float32); O.accumulate(t, src); close(t, src.sum(0), 1e-5)
    t = np.zeros((5, 1), np.float32); O.accumulate(t, src); close(t, src.sum(1, keepdims=True), 1e-5)
    t = np.zeros((), np.float32); O.accumulate(t, src); close(t, src.sum(), 1e-4)
    src4 = rng.random((2, 3, 4, 5), dtype=np.float32)
    t = np.zeros((3, 1, 1), np.float32); O.accumulate(t, src4)
    close(t, src4.sum((0, 2, 3)).reshape(3, 1, 1), 1e-4)


# ------------------------------------------------------------------ pad / chunk
@pytest.mark.parametrize("mode", ["zero", "constant"])
def test_pad_golden(golden, mode):
    c = golden["nodes"][f"pad_{mode}_test"]
    want = f32(c["literals"][0], (7, 9))
    base = np.arange(25, dtype=np.float32).reshape(1, 1, 5, 5)  # Array::range(0,25,1) (5,5)
    out = np.zeros((1, 1, 7, 9), np.float32)
    value = float(want[0, 0])
    O.pad_constant_forward(base, out, (1, 2), value)
    assert np.array_equal(out[0, 0], want), c["cite"]
    dx = np.zeros_like(base)
    O.pad_backward(dx, out, (1, 2))
    assert np.array_equal(dx, base)


@pytest.mark.parametrize("mode", ["reflective", "replicative"])
@pytest.mark.parametrize("fn", ["test_1d", "test_2d", "test_3d"])
def test_pad_mode_golden(golden, mode, fn):
    """pad/{reflective,replicative}/test.rs: exact (`assert_eq!`) expectations for 1-, 2- and 3-d samples."""
    c = golden["nodes"][f"pad_{mode}_{fn}"]
    base = np.arange(c["arange"], dtype=np.float32).reshape([1, 1] + c["base_shape"])
    out = np.zeros([1, 1] + c["padded_shape"], np.float32)
    O.pad_mode_forward(base, out, c["padding"], mode)
    assert np.array_equal(out[0, 0], f32(c["expected"], c["padded_shape"])), c["cite"]
    with pytest.raises(IndexError):
        O.pad_mode_forward(base, np.zeros([1, 1] + [n + 2 * n for n in c["base_shape"]], np.float32), c["base_shape"], "reflective")


def test_chunk_golden(golden):
    lit = golden["nodes"]["chunk_forward_base_case"]["literals"]
    x = np.linspace(-4.0, 4.0, 9, dtype=np.float32).reshape(3, 3)
    for i in range(3):
        out = np.zeros((1, 3), np.float32)
        O.chunk_forward(x, out, i)
        assert np.array_equal(out, f32(lit[i], (1, 3)))
    lit = golden["nodes"]["chunk_backward_base_case"]["literals"]
    g = np.ones((1, 3), np.float32)
    for i in range(3):          # chunk/test.rs:71-121: fresh zero grad per chunk, once then twice
        dx = np.zeros((3, 3), np.float32)
        O.chunk_backward(dx, g, i); assert np.array_equal(dx, f32(lit[2 * i], (3, 3)))
        O.chunk_backward(dx, g, i); assert np.array_equal(dx, f32(lit[2 * i + 1], (3, 3)))


# ------------------------------------------------------------------ dropout (dropout/test.rs:57-158)
def test_dropout_reference_properties():
    x = np.arange(9, dtype=np.float32).reshape(3, 3) + 1
    y = np.ones_like(x)
    noise = np.zeros_like(x)
    O.dropout_forward(x, y, noise, 1.0, True); assert np.array_equal(y, np.zeros_like(x))
    O.dropout_forward(x, y, noise, 0.0, True); assert np.array_equal(y, x)
    noise = O.dropout_noise(x.size, 0.5, seed=7, offset=0).reshape(x.shape)
    O.dropout_forward(x, y, noise, 0.5, True)
    assert np.all(y <= 2 * x) and set(np.unique(noise)) <= {0.0, 1.0}
    dx = np.zeros_like(x)
    O.dropout_backward(dx, np.ones_like(x), noise, 0.5, True)
    assert np.array_equal(dx, noise)          # NOT divided by (1-p): reference quirk
    dx[...] = 0
    O.dropout_backward(dx, np.ones_like(x), np.zeros_like(x), 1.0, True)
    assert np.array_equal(dx, np.zeros_like(x))
    dx[...] = 0
    O.dropout_backward(dx, np.ones_like(x), noise, 0.0, True)
    assert np.array_equal(dx, np.ones_like(x))
    with pytest.raises(AssertionError, match="Wrong probability"):
        O.dropout_forward(x, y, noise, 1.5, True)


def test_philox_known_answer():
    """Philox4x32-10 known-answer test from the Random123 distribution (kat_vectors):
    counter=0,key=0 and counter=ff..,key=ff.. ."""
    r = O.philox4x32_10(np.zeros((1, 4), np.uint32), np.zeros(2, np.uint32))[0]
    assert [hex(int(v)) for v in r] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    ff = np.full((1, 4), 0xFFFFFFFF, np.uint32)
    r = O.philox4x32_10(ff, np.full(2, 0xFFFFFFFF, np.uint32))[0]
    assert [hex(int(v)) for v in r] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]


def test_dropout_noise_rate():
    n = O.dropout_noise(1 << 16, 0.1, seed=7, offset=3)
    assert abs(n.mean() - 0.9) < 0.01


def test_dropout_noise_draw_layout():
    """The draw layout restated element by element (scalar loops): call i // 8 + offset, word (i % 8) // 2, rotated by 16
    bits for odd i, kept iff below floor((1 - p) * 2^32) - rand 0.8's Bernoulli construction on 32-bit values."""
    n, p, seed, off = 203, 0.3, 0x0123456789ABCDEF, (1 << 32) - 5        # ragged tail, counter carry into the high word
    got = O.dropout_noise(n, p, seed, off)
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint32)
    thr = int(np.floor((1.0 - p) * 2.0 ** 32))
    assert int(O.bernoulli_threshold(1.0 - p)) == thr
    for i in range(n):
        c = i // 8 + off
        w = int(O.philox4x32_10(np.array([[c & 0xFFFFFFFF, c >> 32, 0, 0]], dtype=np.uint32), key)[0][(i % 8) // 2])
        if i % 2:
            w = ((w << 16) | (w >> 16)) & 0xFFFFFFFF
        assert got[i] == (1.0 if w < thr else 0.0), i
    assert O.dropout_draws_calls(203) == 26 and O.dropout_draws_calls(208) == 26 and O.dropout_draws_calls(209) == 27
    # a later forward (offset advanced by the calls consumed) continues the stream: no element of the two masks shares a call
    a, b = O.dropout_noise(64, p, seed, 10), O.dropout_noise(64, p, seed, 10 + O.dropout_draws_calls(64))
    assert np.array_equal(O.dropout_noise(128, p, seed, 10), np.concatenate([a, b]))
    assert int(O.bernoulli_threshold(1.0)) == 0xFFFFFFFF and int(O.bernoulli_threshold(0.0)) == 0
    assert int(O.bernoulli_threshold(0.5)) == 1 << 31


@pytest.mark.parametrize("p", [0.1, 0.25, 1.0 / 3.0, 0.5, 0.9])
def test_dropout_noise_statistics(p):
    """Keep rate within 4 sigma of 1 - p over 2^20 draws, and the two draws that share a Philox word (one reads it as
    it is, the other rotated by 16 bits) are uncorrelated: P(both kept) = (1 - p)^2 within 4 sigma."""
    n = 1 << 20
    z = O.dropout_noise(n, p, seed=2026, offset=11).astype(np.float64)
    keep = 1.0 - p
    assert abs(z.mean() - keep) < 4 * np.sqrt(keep * p / n)
    both = z[0::2] * z[1::2]
    assert abs(both.mean() - keep * keep) < 4 * np.sqrt(keep * keep * (1 - keep * keep) / (n / 2))
    nb = z[1:-1:2] * z[2::2]                       # neighbours from different words
    assert abs(nb.mean() - keep * keep) < 4 * np.sqrt(keep * keep * (1 - keep * keep) / (n / 2))


def test_dropout_scale_is_the_f32_subtraction():
    """`(1. - self.p as f32)` (dropout/mod.rs:76): the cast binds tighter than the subtraction - f32(1) - f32(p), which
    differs from f32(1 - p) by one ulp for e.g. p = 0.09."""
    for p in (0.09, 0.16, 0.33, 0.1, 0.5):
        x = np.linspace(0.5, 2.0, 64, dtype=np.float32)
        y = np.zeros_like(x)
        O.dropout_forward(x, y, np.ones_like(x), p, True)
        assert np.array_equal(y, x / (np.float32(1.0) - np.float32(p)))
    assert np.float32(1.0 - 0.09) != np.float32(1.0) - np.float32(0.09)


# ------------------------------------------------------------------ tape composition sanity
def test_mlp_step_matches_f64_autograd_free_formula():
    rng = np.random.default_rng(0)
    x = rng.random((8, 3), dtype=np.float32); t = rng.random((8, 1), dtype=np.float32)
    params = [(rng.random((5, 3), dtype=np.float32) - 0.5, rng.random(5, dtype=np.float32) - 0.5),
              (rng.random((1, 5), dtype=np.float32) - 0.5, rng.random(1, dtype=np.float32) - 0.5)]
    loss, grads = O.mlp_step(x, t, params)
    p64 = [(w.astype(np.float64), b.astype(np.float64)) for w, b in params]
    eps = 1e-6
    def L(p):
        h = x.astype(np.float64)
        h = np.maximum(h @ p[0][0].T + p[0][1], 0)
        h = h @ p[1][0].T + p[1][1]
        return ((h - t) ** 2).mean()
    close(loss, L(p64), 1e-5)
    w = p64[0][0]; num = np.zeros_like(w)
    for i in np.ndindex(*w.shape):
        w[i] += eps; up = L(p64); w[i] -= 2 * eps; dn = L(p64); w[i] += eps
        num[i] = (up - dn) / (2 * eps)
    close(grads[0][0], num, 1e-4)
