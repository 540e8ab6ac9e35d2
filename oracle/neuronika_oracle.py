"""CPU oracle for the neuronika dense-tensor hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the reference's ndarray node kernels
(`/root/reference/neuronika-variable/src/node/*/mod.rs`).  It exists to CHECK the HIP
backend; nothing in the product path (`neuronika_amd/`, `include/`) may import it.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it.

Parity status: PINNED.  The reference is Rust and cannot be compiled here (no rustc/cargo),
so the restatement is pinned against the known-answer vectors of the reference's own unit
tests, transcribed by `tests/golden/extract_reference_fixtures.py` into
`tests/golden/*.json` and replayed by `tests/test_oracle_golden.py`.

Third-party arithmetic restated (not under /root/reference): `ndarray ^0.15.4`
`linalg::general_mat_mul` -> `matrixmultiply ^0.3` sgemm (C = alpha*A*B + beta*C);
`ndarray::ArrayBase::{sum,mean,fold}`; Rust std f32 `exp`/`ln`; `rand 0.8` Bernoulli.
Summation order of those libraries is not reproducible bit for bit, so contractions and
reductions are compared with a stated tolerance; data movement, masks and index-like
behaviour are compared bit-exactly.

Conventions restated from the reference:
  * every forward OVERWRITES its output buffer (GEMM beta = 0),
  * every backward ACCUMULATES (`+=`) into the operand gradient (GEMM beta = 1),
  * all arrays are C-contiguous; dtype is whatever the caller passes (f32 twin / f64 yardstick).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------
# a-1 / a-2  MatMul, MatMulT      node/matrix_matrix_mul/mod.rs, node/matrix_matrix_mul_t/mod.rs
# --------------------------------------------------------------------------------------


def mm_forward(left, right, out):
    """`MatrixMatrixMul::forward` (matrix_matrix_mul/mod.rs:31-41): out = left . right, beta=0."""
    np.matmul(left, right, out=out)


def mm_backward_left(left_grad, grad, right):
    """`MatrixMatrixMulBackwardLeft::backward` (:63-73): dA += G . B^T."""
    left_grad += grad @ right.T


def mm_backward_right(right_grad, grad, left):
    """`MatrixMatrixMulBackwardRight::backward` (:95-105): dB += A^T . G."""
    right_grad += left.T @ grad


def mm_t_forward(left, right, out):
    """`MatrixMatrixMulT::forward` (matrix_matrix_mul_t/mod.rs:31-41): out = left . right^T."""
    np.matmul(left, right.T, out=out)


def mm_t_backward_left(left_grad, grad, right):
    """`MatrixMatrixMulTBackwardLeft::backward` (:63-73): dA += G . B."""
    left_grad += grad @ right


def mm_t_backward_right(right_grad, grad, left):
    """`MatrixMatrixMulTBackwardRight::backward` (:95-105): dB += G^T . A."""
    right_grad += grad.T @ left


# --------------------------------------------------------------------------------------
# a-3  Convolution (1/2/3-d, stride, dilation, groups)        node/convolution/mod.rs
# --------------------------------------------------------------------------------------


def conv_out_shape(input_shape, kernel_shape, stride, dilation):
    """`conv_out_shape` (utils.rs:207-237): no internal padding."""
    spatial = [
        (i - d * (k - 1) - 1) // s + 1
        for i, k, s, d in zip(input_shape[2:], kernel_shape[2:], stride, dilation)
    ]
    return (input_shape[0], kernel_shape[0], *spatial)


def check_conv_args(input_shape, kernel_shape, stride, dilation):
    """`check_conv_args` (utils.rs:427-474); panics become AssertionError."""
    nd = len(input_shape) - 2
    assert nd == len(stride), f"Invalid stride {list(stride)} for {nd}d conv."
    assert nd == len(dilation), f"Invalid dilation {list(dilation)} for {nd}d conv."
    assert len(kernel_shape) == len(input_shape), (
        f"Invalid kernel shape {list(kernel_shape)} for {nd}d conv"
    )
    for i, k, d in zip(input_shape[2:], kernel_shape[2:], dilation):
        assert i >= (k - 1) * d + 1, "The kernel size can't be greater than actual input size."


def check_groups_args(input_shape, kernel_shape, groups):
    """`check_groups_args` (utils.rs:481-497)."""
    assert input_shape[1] % groups == 0, (
        f"In channels {input_shape[1]} is not divisible by groups {groups}"
    )
    assert kernel_shape[0] % groups == 0, (
        f"Out channels {kernel_shape[0]} is not divisible by groups {groups}"
    )


def im2col(x, kernel_shape, stride, dilation):
    """Rolling-window view -> dense columns, `as_windows` + `columns_shape`
    (utils.rs:249-353, 403-420): result (N, L, Cin*prod(k)) with the window axis ordered
    (ci, k0, k1, ...), L = prod(out spatial) in row-major order."""
    nd = x.ndim - 2
    ks = tuple(kernel_shape[2:])
    out_sp = conv_out_shape(x.shape, kernel_shape, stride, dilation)[2:]
    n, c = x.shape[:2]
    cols = np.empty((n, *out_sp, c, *ks), dtype=x.dtype)
    for kidx in np.ndindex(*ks):
        sl = tuple(
            slice(kidx[a] * dilation[a], kidx[a] * dilation[a] + stride[a] * (out_sp[a] - 1) + 1, stride[a])
            for a in range(nd)
        )
        patch = x[(slice(None), slice(None), *sl)]  # (N, C, *out_sp)
        cols[(slice(None), *([slice(None)] * nd), slice(None), *kidx)] = np.moveaxis(patch, 1, -1)
    return cols.reshape(n, int(np.prod(out_sp)), c * int(np.prod(ks)))


def _col2im_add(dest, cols, kernel_shape, stride, dilation):
    """`assign_from_cols` (convolution/mod.rs:63-83): dest windows += cols."""
    nd = dest.ndim - 2
    ks = tuple(kernel_shape[2:])
    out_sp = conv_out_shape(dest.shape, kernel_shape, stride, dilation)[2:]
    n, c = dest.shape[:2]
    cols = cols.reshape(n, *out_sp, c, *ks)
    for kidx in np.ndindex(*ks):
        sl = tuple(
            slice(kidx[a] * dilation[a], kidx[a] * dilation[a] + stride[a] * (out_sp[a] - 1) + 1, stride[a])
            for a in range(nd)
        )
        src = cols[(slice(None), *([slice(None)] * nd), slice(None), *kidx)]  # (N,*out_sp,C)
        dest[(slice(None), slice(None), *sl)] += np.moveaxis(src, -1, 1)


def _convolution(x, w, out, stride, dilation):
    """`convolution` (convolution/mod.rs:85-123): out[n] = Wflat . cols[n]^T, beta = 0."""
    wf = w.reshape(w.shape[0], -1)
    cols = im2col(x, w.shape, stride, dilation)
    res = np.einsum("ok,nlk->nol", wf, cols, optimize=True)
    out[...] = res.reshape(out.shape)


def _convolution_backward_input(x_grad, grad, w, stride, dilation):
    """`convolution_backward_input` (:146-189): buf[n] = Wflat^T . G[n]; col2im `+=`."""
    wf = w.reshape(w.shape[0], -1)
    g = grad.reshape(grad.shape[0], grad.shape[1], -1)  # (N, Cout, L)
    buf = np.einsum("ok,nol->nlk", wf, g, optimize=True)  # windows order (N, L, K)
    _col2im_add(x_grad, buf, w.shape, stride, dilation)


def _convolution_backward_kernel(w_grad, grad, x, stride, dilation):
    """`convolution_backward_kernel` (:191-226): dW[c,:] += G[:,c,:] . cols."""
    cols = im2col(x, w_grad.shape, stride, dilation)  # (N, L, K)
    g = grad.reshape(grad.shape[0], grad.shape[1], -1)  # (N, Cout, L)
    dw = np.einsum("nol,nlk->ok", g, cols, optimize=True)
    w_grad += dw.reshape(w_grad.shape)


def _groups(a, axis, groups):
    step = a.shape[axis] // groups
    for g in range(groups):
        idx = [slice(None)] * a.ndim
        idx[axis] = slice(g * step, (g + 1) * step)
        yield tuple(idx)


def convolution_forward(x, w, out, stride, dilation, groups=1):
    """`Convolution::forward` (:331-355) -> `grouped_convolution` (:125-144): input split on
    axis 1, kernel on axis 0, output on axis 1."""
    for xs, ws, os in zip(_groups(x, 1, groups), _groups(w, 0, groups), _groups(out, 1, groups)):
        o = np.zeros(out[os].shape, dtype=out.dtype)
        _convolution(np.ascontiguousarray(x[xs]), np.ascontiguousarray(w[ws]), o, stride, dilation)
        out[os] = o


def convolution_backward_input(x_grad, grad, w, stride, dilation, groups=1):
    """`grouped_convolution_backward_input` (:256-274)."""
    for xs, gs, ws in zip(_groups(x_grad, 1, groups), _groups(grad, 1, groups), _groups(w, 0, groups)):
        xg = np.ascontiguousarray(x_grad[xs])
        _convolution_backward_input(xg, np.ascontiguousarray(grad[gs]), np.ascontiguousarray(w[ws]), stride, dilation)
        x_grad[xs] = xg


def convolution_backward_kernel(w_grad, grad, x, stride, dilation, groups=1):
    """`grouped_convolution_backward_kernel` (:276-294)."""
    for ws, gs, xs in zip(_groups(w_grad, 0, groups), _groups(grad, 1, groups), _groups(x, 1, groups)):
        wg = np.ascontiguousarray(w_grad[ws])
        _convolution_backward_kernel(wg, np.ascontiguousarray(grad[gs]), np.ascontiguousarray(x[xs]), stride, dilation)
        w_grad[ws] = wg


# --------------------------------------------------------------------------------------
# a-4 / a-5  broadcast binaries + un-broadcast accumulate
# --------------------------------------------------------------------------------------


def cobroadcast(left_shape, right_shape):
    """`cobroadcast` (utils.rs:97-125): NumPy right-aligned broadcasting, ndim = max."""
    bigger, smaller = (left_shape, right_shape) if len(left_shape) >= len(right_shape) else (right_shape, left_shape)
    out = list(bigger)
    off = len(bigger) - len(smaller)
    for i, r in enumerate(smaller):
        l = out[off + i]
        if l != r:
            if l == 1:
                out[off + i] = r
            else:
                assert r == 1, "The two tensors have incompatible shape."
    return tuple(out)


def accumulate(target, source):
    """INTENDED semantics of `utils::accumulate` (utils.rs:152-192): target += source summed
    over every axis `target` lacks or has with extent 1.  The reference's lane-axis choice is
    defective for non-square shapes; it is NOT replicated.  The
    reference's own fixtures (addition/test.rs:110-124, multiplication/test.rs:121-139) pass
    against this."""
    if source.shape == target.shape:
        target += source
        return
    k = source.ndim - target.ndim
    red = source.sum(axis=tuple(range(k))) if k > 0 else source
    keep = tuple(i for i, (t, s) in enumerate(zip(target.shape, red.shape)) if t == 1 and s != 1)
    if keep:
        red = red.sum(axis=keep, keepdims=True)
    target += red.reshape(target.shape)


_BIN = {
    "add": lambda l, r: l + r,
    "sub": lambda l, r: l - r,
    "mul": lambda l, r: l * r,
    "div": lambda l, r: l / r,
}


def binary_forward(op, left, right, out):
    """`Addition|Subtraction|Multiplication|Division::forward` (node/<op>/mod.rs:39-50)."""
    out[...] = _BIN[op](left, right)


def binary_backward_left(op, left_grad, grad, left, right):
    """`<Op>BackwardLeft::backward`: addition/mod.rs:86-91, subtraction/mod.rs:87-92,
    multiplication/mod.rs:91-103 (buf = g*r), division/mod.rs:90-99 (buf = g/r)."""
    if op in ("add", "sub"):
        local = grad
    elif op == "mul":
        local = grad * right
    else:
        local = grad / right
    accumulate(left_grad, np.broadcast_to(local, grad.shape) if np.ndim(local) else local)


def binary_backward_right(op, right_grad, grad, left, right):
    """`<Op>BackwardRight::backward`: addition/mod.rs:129-134, subtraction/mod.rs:130-136
    (-g), multiplication/mod.rs:138-149 (g*l), division/mod.rs:139-149 (-g*l/r^2)."""
    if op == "add":
        local = grad
    elif op == "sub":
        local = -grad
    elif op == "mul":
        local = grad * left
    else:
        local = -grad * left / (right * right)
    accumulate(right_grad, local)


# --------------------------------------------------------------------------------------
# a-6  Sum / Mean          node/sum/mod.rs, node/mean/mod.rs
# --------------------------------------------------------------------------------------


def sum_forward(x, out):
    """`Sum::forward` (sum/mod.rs:28-35): full reduction to Ix0."""
    out[...] = x.sum(dtype=x.dtype)


def sum_backward(x_grad, grad):
    """`SumBackward::backward` (:60-67): dx += g."""
    x_grad += grad


def mean_forward(x, out):
    """`Mean::forward` (mean/mod.rs:28-35): sum/len."""
    out[...] = x.sum(dtype=x.dtype) / x.dtype.type(x.size)


def mean_backward(x_grad, grad):
    """`MeanBackward::backward` (:60-72): dx += g/len."""
    x_grad += grad / x_grad.dtype.type(x_grad.size)


# --------------------------------------------------------------------------------------
# a-7  Softmax / LogSoftmax      node/softmax/mod.rs, node/logsoftmax/mod.rs
# --------------------------------------------------------------------------------------


def softmax_forward(x, out, axis):
    """`Softmax::forward` (softmax/mod.rs:37-53): m = max, e = exp(x-m), y = e/sum(e)."""
    m = np.maximum(x.max(axis=axis, keepdims=True), np.finfo(np.float32).min).astype(x.dtype)
    e = np.exp(x - m)
    out[...] = e / e.sum(axis=axis, keepdims=True, dtype=x.dtype)


def softmax_backward(x_grad, grad, data, axis):
    """`SoftmaxBackward::backward` (:84-104): dx += y*(g - sum(g*y))."""
    s = (grad * data).sum(axis=axis, keepdims=True, dtype=data.dtype)
    x_grad += data * (grad - s)


def log_softmax_forward(x, out, axis):
    """`LogSoftmax::forward` (logsoftmax/mod.rs:37-53): y = x - ln(sum exp(x-m)) - m."""
    m = np.maximum(x.max(axis=axis, keepdims=True), np.finfo(np.float32).min).astype(x.dtype)
    lse = np.log(np.exp(x - m).sum(axis=axis, keepdims=True, dtype=x.dtype))
    out[...] = x - lse - m


def log_softmax_backward(x_grad, grad, data, axis):
    """`LogSoftmaxBackward::backward` (:84-102): dx += g - exp(y)*sum(g)."""
    x_grad += grad - np.exp(data) * grad.sum(axis=axis, keepdims=True, dtype=data.dtype)


# --------------------------------------------------------------------------------------
# a-8  Dropout            node/dropout/mod.rs
# --------------------------------------------------------------------------------------

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al., SC'11), vectorised.  `counter` is (n,4) uint32, `key`
    (2,) uint32.  This is the device RNG of the HIP backend (the reference uses
    `rand::thread_rng`, which is non-reproducible by design, dropout/mod.rs:68-70), restated
    here so the keep/drop pattern can be checked bit-exactly."""
    c = counter.astype(np.uint32).copy()
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _PHILOX_M0 * c[:, 0].astype(np.uint64)
            p1 = _PHILOX_M1 * c[:, 2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & mask).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & mask).astype(np.uint32)
            c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], axis=1)
            k0 = np.uint32(k0 + _PHILOX_W0)
            k1 = np.uint32(k1 + _PHILOX_W1)
    return c


DRAWS_PER_PHILOX_CALL = 8


def dropout_draws_calls(n):
    """Philox calls one forward over `n` elements consumes = how far the host advances the counter offset per forward."""
    return (n + DRAWS_PER_PHILOX_CALL - 1) // DRAWS_PER_PHILOX_CALL


def bernoulli_threshold(keep):
    """`Bernoulli::new(1. - p)` (dropout/mod.rs:46) is rand 0.8's integer construction - `p_int = (p * 2^64) as u64`,
    sample = `rng.gen::<u64>() < p_int` (crate rand 0.8.x, src/distributions/bernoulli.rs; not under /root/reference) -
    restated on 32-bit words: keep iff v < floor(keep * 2^32), `keep` in f64 as the reference passes it."""
    t = int(np.floor(np.float64(keep) * 4294967296.0))
    return np.uint32(min(t, 0xFFFFFFFF))


def dropout_noise(n, p, seed, offset):
    """Bernoulli(1-p) 0/1 noise for `n` elements as the HIP backend draws it.  One Philox4x32-10 call
    (counter = (i // 8 + offset) as 64-bit lo/hi, 0, 0; key = seed lo/hi) serves EIGHT consecutive elements: element
    i takes word (i % 8) // 2, as it is for even i and rotated by 16 bits for odd i, and is kept iff that 32-bit value
    is below `bernoulli_threshold(1 - p)`.  Every draw thus has the keep probability to 2^-32 (its own 16 bits decide,
    the partner's 16 bits only break ties); the two elements sharing a word are independent except on those 2^-16 ties.
    Half the Philox rounds per element of the one-word-per-element layout of rounds 1-2 - the fused attention forward
    pays ~2000 of a masked tile's ~6700 issue cycles for them."""
    nblk = dropout_draws_calls(n)
    idx = np.arange(nblk, dtype=np.uint64) + np.uint64(offset)
    ctr = np.zeros((nblk, 4), dtype=np.uint32)
    ctr[:, 0] = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (idx >> np.uint64(32)).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    words = philox4x32_10(ctr, key)                                         # (nblk, 4)
    rot = (words << np.uint32(16)) | (words >> np.uint32(16))
    draws = np.stack([words, rot], axis=2).reshape(-1)[:n]                  # w0, rot(w0), w1, rot(w1), ...
    return (draws < bernoulli_threshold(1.0 - p)).astype(np.float32)


def dropout_forward(x, out, noise, p, train=True):
    """`Dropout::forward` (dropout/mod.rs:53-79) with the 0/1 `noise` supplied by the caller
    (the reference resamples it with thread_rng on every forward).  eval or p == 0: copy;
    p == 1: zeros (noise untouched); else y = (x*noise)/(1-p)."""
    assert 0.0 <= p <= 1.0, f"Wrong probability received: {p}."
    if (not train) or p == 0.0:
        out[...] = x
        return
    if 1.0 - p == 0.0:
        out[...] = 0
        return
    # `(1. - self.p as f32)` (dropout/mod.rs:76): the cast binds first - an f32 subtraction of the rounded p
    out[...] = (x * noise) / x.dtype.type(np.float32(1.0) - np.float32(p))


def dropout_backward(x_grad, grad, noise, p, train=True):
    """`DropoutBackward::backward` (:113-128): eval or p == 0: dx += g; else dx += g*noise —
    NOT divided by (1-p): reference behaviour, replicated on purpose."""
    if (not train) or p == 0.0:
        x_grad += grad
        return
    x_grad += grad * noise


# --------------------------------------------------------------------------------------
# a-9  ReLU               node/relu/mod.rs
# --------------------------------------------------------------------------------------


def relu_forward(x, out):
    """`ReLU::forward` (relu/mod.rs:29-38): `o.max(0.)`.  Rust's `f32::max` returns the other operand when one is NaN, so a
    NaN input gives 0 (NumPy's `maximum` would propagate it); -inf -> 0, +inf -> +inf."""
    out[...] = np.where(x > 0, x, x.dtype.type(0))


def relu_backward(x_grad, grad, x):
    """`ReLUBackward::backward` (:67-79): dx += (x > 0) * g, strict, on the INPUT."""
    x_grad += (x > 0).astype(grad.dtype) * grad


# --------------------------------------------------------------------------------------
# f-2  pointwise unary nodes   node/{negation,exp,logn,sqrt,sigmoid,tanh,softplus,leaky_relu,power}
# --------------------------------------------------------------------------------------


def _powi_arr(x, e):
    r, b, k = np.ones_like(x), x.copy(), abs(int(e))
    while k:
        if k & 1:
            r = r * b
        b = b * b
        k >>= 1
    return (1 / r) if e < 0 else r


UNARY_KEEPS_OUTPUT = {"exp", "sqrt", "sigmoid", "tanh"}   # the others keep the operand (input)


def unary_forward(op, x, out, exp=0):
    """node/<op>/mod.rs:35 (power: :44, leaky_relu: :36-38)."""
    dt = x.dtype.type
    out[...] = {
        "neg": lambda: -x, "exp": lambda: np.exp(x), "ln": lambda: np.log(x), "sqrt": lambda: np.sqrt(x),
        "sigmoid": lambda: dt(1) / (dt(1) + np.exp(-x)), "tanh": lambda: np.tanh(x),
        "softplus": lambda: np.log(dt(1) + np.exp(x)),
        "leaky_relu": lambda: (x > 0).astype(x.dtype) * x + (x <= 0).astype(x.dtype) * (dt(0.01) * x),
        "pow": lambda: _powi_arr(x, exp),
    }[op]()


def unary_backward(op, x_grad, grad, ref, exp=0):
    """node/<op>/mod.rs:73-86; `ref` = the buffer the node keeps (see UNARY_KEEPS_OUTPUT).
    leaky_relu replicates the reference (adds 0.01, not 0.01*g, where x <= 0: leaky_relu/mod.rs:77-80)."""
    dt = grad.dtype.type
    if op == "neg":
        x_grad -= grad
        return
    x_grad += {
        "exp": lambda: grad * ref, "ln": lambda: grad / ref, "sqrt": lambda: grad / (ref * dt(2)),
        "sigmoid": lambda: grad * ref * (dt(1) - ref), "tanh": lambda: grad * (dt(1) - ref * ref),
        "softplus": lambda: grad / (dt(1) + np.exp(-ref)),
        "leaky_relu": lambda: (ref > 0).astype(grad.dtype) * grad + (ref <= 0).astype(grad.dtype) * dt(0.01),
        "pow": lambda: grad * _powi_arr(ref, exp - 1) * dt(exp),
    }[op]()


# --------------------------------------------------------------------------------------
# a-10  glue: SquaredError, Pad(Constant/Zero), Chunk, MultiConcatenate, Transpose
# --------------------------------------------------------------------------------------


def squared_error_forward(x, target, out, reduction="mean"):
    """`SquaredError::forward` (squared_error/mod.rs:42-59)."""
    total = ((x - target) ** 2).sum(dtype=x.dtype)
    out[...] = total / x.dtype.type(x.size) if reduction == "mean" else total


def squared_error_backward(x_grad, grad, x, target, reduction="mean"):
    """`SquaredErrorBackward::backward` (:94-123): dx += 2(x-t)*g[/n]."""
    local = (2 * (x - target)) * grad
    if reduction == "mean":
        local = local / x.dtype.type(x.size)
    x_grad += local


def pad_constant_forward(x, out, padding, value=0.0):
    """`Pad::forward` with `Constant(value)` / `Zero` (pad/mod.rs:97-129,
    pad/constant/mod.rs:14-39, pad/zero/mod.rs:12-22): symmetric padding[i] on both sides of
    spatial axis i (axes 2..), fill then assign the centre."""
    out[...] = value
    centre = (slice(None), slice(None)) + tuple(slice(p, out.shape[2 + i] - p) for i, p in enumerate(padding))
    out[centre] = x


def pad_mode_forward(x, out, padding, mode):
    """`Pad<Reflective|Replicative>::forward` (pad/mod.rs:97-129 per (N*C) sample; index maps of
    pad/reflective/mod.rs:9-136 and pad/replicative/mod.rs:9-134), restated as one gather per axis:
    reflective: i<pad -> pad-i, i>=len+pad -> 2(len-1)-(i-pad); replicative: clamp to [0, len-1]."""
    src = x
    for ax, pad in enumerate(padding):
        n = x.shape[2 + ax]
        c = np.arange(n + 2 * pad) - pad
        if mode == "reflective":
            if pad and pad >= n:
                raise IndexError("reflective padding needs input extent > padding")  # reference: slice index panic
            idx = np.where(c < 0, -c, np.where(c >= n, 2 * (n - 1) - c, c))
        elif mode == "replicative":
            idx = np.clip(c, 0, n - 1)
        else:
            raise ValueError(mode)
        src = np.take(src, idx, axis=2 + ax)
    out[...] = src


def pad_backward(x_grad, grad, padding):
    """`PadBackward::backward` (pad/mod.rs:157-181): dx += centre slice of g."""
    centre = (slice(None), slice(None)) + tuple(slice(p, grad.shape[2 + i] - p) for i, p in enumerate(padding))
    x_grad += grad[centre]


def _chunk_slices(operand_shape, chunk_shape, chunk_no):
    """`exact_chunks(shape)` iteration order: row-major over the chunk grid, remainder skipped."""
    grid = [o // c for o, c in zip(operand_shape, chunk_shape)]
    idx = np.unravel_index(chunk_no, grid)
    return tuple(slice(i * c, (i + 1) * c) for i, c in zip(idx, chunk_shape))


def chunk_forward(x, out, chunk_no):
    """`Chunk::forward` (chunk/mod.rs:48-64)."""
    out[...] = x[_chunk_slices(x.shape, out.shape, chunk_no)]


def chunk_backward(x_grad, grad, chunk_no):
    """`ChunkBackward::backward` (:99-113): tile `+=`."""
    x_grad[_chunk_slices(x_grad.shape, grad.shape, chunk_no)] += grad


def multi_concatenate_forward(operands, out, axis):
    """`MultiConcatenate::forward` (multi_concatenate/mod.rs:37-50)."""
    off = 0
    for o in operands:
        idx = [slice(None)] * out.ndim
        idx[axis] = slice(off, off + o.shape[axis])
        out[tuple(idx)] = o
        off += o.shape[axis]


def multi_concatenate_backward(operand_grads, grad, axis):
    """`MultiConcatenateBackward::backward` (:81-97): slice `+=`."""
    off = 0
    for og in operand_grads:
        idx = [slice(None)] * grad.ndim
        idx[axis] = slice(off, off + og.shape[axis])
        og += grad[tuple(idx)]
        off += og.shape[axis]


def transpose_forward(x, out):
    """`Transpose::forward` (transpose/mod.rs:28-37): reversed axes, materialised."""
    out[...] = x.T


def transpose_backward(x_grad, grad):
    """`TransposeBackward::backward` (:62-69): dx += g^T."""
    x_grad += grad.T


# --------------------------------------------------------------------------------------
# f-1  optimizer steps (first "next" row): SGD / Adam / AMSGrad / Adagrad / RMSProp + penalties
# --------------------------------------------------------------------------------------


def penalty_grad(w, l1=0.0, l2=0.0):
    """`Penalty::penalize` (neuronika-optim/src/penalty.rs:63-79): L1 -> l1*signum(w) with Rust's
    `f32::signum` (+0 -> 1, -0 -> -1), L2 -> 2*l2*w, ElasticNet -> both."""
    dt = w.dtype.type
    g = np.zeros_like(w)
    if l1 != 0.0:
        g = g + dt(l1) * np.copysign(dt(1), w)
    if l2 != 0.0:
        g = g + dt(2) * dt(l2) * w
    return g


def sgd_step(w, grad, lr, velocity=None, momentum=0.0, dampening=0.0, nesterov=False, l1=0.0, l2=0.0):
    """`SGDParam::optimize` (sgd/mod.rs:186-236).  In place: grad += penalty(w); plain
    (`velocity is None`, i.e. momentum <= f32::EPSILON): w -= grad*lr.  Otherwise
    buffer = buffer*momentum + grad*(1-dampening) (the buffer starts at ZERO; there is no
    "first step: buffer = grad" special case in the reference); nesterov:
    w -= (grad + buffer*momentum)*lr, else w -= buffer*lr."""
    dt = w.dtype.type
    grad += penalty_grad(w, l1, l2)
    if velocity is None:
        w -= grad * dt(lr)
        return
    velocity[...] = velocity * dt(momentum) + grad * dt(1.0 - dampening)
    if nesterov:
        w -= (grad + velocity * dt(momentum)) * dt(lr)
    else:
        w -= velocity * dt(lr)


def _powi(b, e, dt):
    r, b, k = dt(1), dt(b), abs(int(e))
    while k:
        if k & 1:
            r = dt(r * b)
        b = dt(b * b)
        k >>= 1
    return r


def adam_step(w, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, max_exp_avg_sq=None, l1=0.0, l2=0.0):
    """`AdamParam::optimize` (adam/mod.rs:131-169) / `AMSGradParam::optimize`
    (amsgrad/mod.rs:163-205, when `max_exp_avg_sq` is given).  `step` is 1-based."""
    dt = w.dtype.type
    bc1, bc2 = dt(1) - _powi(beta1, step, dt), dt(1) - _powi(beta2, step, dt)
    grad += penalty_grad(w, l1, l2)
    exp_avg[...] = exp_avg * dt(beta1) + grad * dt(1.0 - beta1)
    exp_avg_sq[...] = exp_avg_sq * dt(beta2) + grad * grad * dt(1.0 - beta2)
    den = exp_avg_sq
    if max_exp_avg_sq is not None:
        max_exp_avg_sq[...] = np.maximum(max_exp_avg_sq, exp_avg_sq)
        den = max_exp_avg_sq
    w -= exp_avg / ((np.sqrt(den) / np.sqrt(bc2)) + dt(eps)) * (dt(lr) / bc1)


def adagrad_step(w, grad, grad_sq, lr, lr_decay, eps, step, l1=0.0, l2=0.0):
    """`AdagradParam::optimize` (adagrad/mod.rs:113-140)."""
    dt = w.dtype.type
    clr = dt(lr) / (dt(1) + dt(step - 1) * dt(lr_decay))
    grad += penalty_grad(w, l1, l2)
    grad_sq += grad * grad
    w -= grad / (np.sqrt(grad_sq) + dt(eps)) * clr


def rmsprop_step(w, grad, square_avg, lr, alpha, eps, grad_avg=None, buffer=None, momentum=0.0, l1=0.0, l2=0.0):
    """`RMSPropParam::optimize` (rmsprop/mod.rs:193-296): `grad_avg` given = centered,
    `buffer` given = momentum (> f32::EPSILON)."""
    dt = w.dtype.type
    grad += penalty_grad(w, l1, l2)
    square_avg[...] = square_avg * dt(alpha) + grad * grad * dt(1.0 - alpha)
    if grad_avg is not None:
        grad_avg[...] = grad_avg * dt(alpha) + grad * dt(1.0 - alpha)
        den = np.sqrt(square_avg + (-grad_avg * grad_avg)) + dt(eps)
    else:
        den = np.sqrt(square_avg) + dt(eps)
    if buffer is not None:
        buffer[...] = buffer * dt(momentum) + grad / den
        w -= buffer * dt(lr)
    else:
        w -= grad / den * dt(lr)


# --------------------------------------------------------------------------------------
# a-11  tape-level compositions used by the BASELINE configs
# --------------------------------------------------------------------------------------


def linear_forward(x, w, b):
    """`nn::Linear::forward` (neuronika-nn/src/lib.rs:441-447): x.mm_t(W) + b."""
    z = np.zeros((x.shape[0], w.shape[0]), dtype=x.dtype)
    mm_t_forward(x, w, z)
    out = np.zeros_like(z)
    binary_forward("add", z, b, out)
    return out


def mlp_step(x, target, params, seed=1.0):
    """One `loss.forward(); loss.backward(seed)` of the C1/C4 MLP on a fresh graph:
    Linear -> ReLU -> ... -> Linear -> MSE(mean).  `params` = [(W1,b1),(W2,b2),...].
    Returns (loss, [(dW1,db1),...]) — input `x` is a non-differentiable `Var`, so no dX for
    layer 1 (var.rs:1081-1094).  Tape: [MMT,Add,ReLU]* MMT,Add,SquaredError."""
    dt = x.dtype
    acts, pre = [x], []
    h = x
    for i, (w, b) in enumerate(params):
        z = linear_forward(h, w, b)
        pre.append(z)
        if i + 1 < len(params):
            a = np.zeros_like(z)
            relu_forward(z, a)
            h = a
            acts.append(a)
        else:
            h = z
    loss = np.zeros((), dtype=dt)
    squared_error_forward(h, target, loss, "mean")

    g_root = np.full((), seed, dtype=dt)
    g = np.zeros_like(h)
    squared_error_backward(g, g_root, h, target, "mean")
    grads = [None] * len(params)
    for i in reversed(range(len(params))):
        w, b = params[i]
        # Addition backward: left (N,out) same shape, right bias un-broadcast.
        g_mm = np.zeros_like(g)
        accumulate(g_mm, g)
        db = np.zeros_like(b)
        accumulate(db, g)
        dw = np.zeros_like(w)
        mm_t_backward_right(dw, g_mm, acts[i])
        grads[i] = (dw, db)
        if i > 0:
            da = np.zeros_like(acts[i])
            mm_t_backward_left(da, g_mm, w)
            g = np.zeros_like(pre[i - 1])
            relu_backward(g, da, pre[i - 1])
    return loss, grads


# --------------------------------------------------------------------------------------------
# loss criteria (row f-4).  `out` is a 0-d array; gradients accumulate.
# --------------------------------------------------------------------------------------------
def _rust_as_usize(t):
    """`target as usize` (nll/mod.rs:57): saturating cast - NaN/negatives -> 0, fraction dropped."""
    t = np.asarray(t, dtype=np.float64)
    return np.where(np.isnan(t) | (t <= 0), 0, np.trunc(np.minimum(t, 2.0 ** 62))).astype(np.int64)


def mae_forward(x, t, reduction="mean"):
    """`AbsoluteError::forward` absolute_error/mod.rs:42-58."""
    s = np.abs(x - t).sum(dtype=x.dtype)
    return s / x.dtype.type(x.size) if reduction == "mean" else s


def mae_backward(x_grad, g, x, t, reduction="mean"):
    """`AbsoluteErrorBackward::backward` :93-123: (diff != 0) * (signum(diff) * g / n)."""
    diff = x - t
    v = np.copysign(1, diff).astype(x.dtype) * x.dtype.type(g)
    if reduction == "mean":
        v = v / x.dtype.type(x.size)
    x_grad += (diff != 0).astype(x.dtype) * v


def bce_forward(x, t, reduction="mean"):
    """`BinaryCrossEntropy::forward` bce/mod.rs:42-62: -t*clamp(ln x,-100) + (t-1)*clamp(ln(1-x),-100)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        lx = np.maximum(np.log(x), -100, where=~np.isnan(x), out=np.log(x))
        l1x = np.maximum(np.log(1 - x), -100, where=~np.isnan(x), out=np.log(1 - x))
    s = (-t * lx + (t - 1) * l1x).sum(dtype=x.dtype)
    return s / x.dtype.type(x.size) if reduction == "mean" else s


def bce_backward(x_grad, g, x, t, reduction="mean"):
    """`BinaryCrossEntropyBackward::backward` :97-127: (x-t)/max((1-x)x, f32::EPSILON) * g / n."""
    v = (x - t) / np.maximum((1 - x) * x, x.dtype.type(np.finfo(np.float32).eps)) * x.dtype.type(g)
    x_grad += v / x.dtype.type(x.size) if reduction == "mean" else v


def bce_with_logits_forward(x, t, reduction="mean"):
    """`BCEWithLogits::forward` bce_with_logits/mod.rs:42-66."""
    m = np.maximum(-x, 0)
    s = ((1 - t) * x + m + np.log(np.exp(-m) + np.exp(-x - m))).sum(dtype=x.dtype)
    return s / x.dtype.type(x.size) if reduction == "mean" else s


def bce_with_logits_backward(x_grad, g, x, t, reduction="mean"):
    """`BCEWithLogitsBackward::backward` :101-131: (sigmoid(x) - t) * g / n."""
    v = (1 / (1 + np.exp(-x)) - t) * x.dtype.type(g)
    x_grad += v / x.dtype.type(x.size) if reduction == "mean" else v


def kldiv_forward(x, t, reduction="mean"):
    """`KLDiv::forward` kldiv/mod.rs:42-59 with the mask applied BEFORE the product (the reference's
    `t*(ln t - x)*(t>0)` is NaN at t = 0, yet kldiv/test.rs:10-21 expects 0.1530 with a 0.0 target);
    Mean divides by len_of(Axis(0))."""
    with np.errstate(divide="ignore", invalid="ignore"):
        term = np.where(t > 0, t * (np.log(np.where(t > 0, t, 1)) - x), 0).astype(x.dtype)
    s = term.sum(dtype=x.dtype)
    return s / x.dtype.type(x.shape[0]) if reduction == "mean" else s


def kldiv_backward(x_grad, g, t, reduction="mean"):
    """`KLDivBackward::backward` :92-113: -t * g / len_of(Axis(0))."""
    v = -t * x_grad.dtype.type(g)
    x_grad += v / x_grad.dtype.type(t.shape[0]) if reduction == "mean" else v


def nll_forward(x, t, reduction="mean"):
    """`NegativeLogLikelihood::forward` nll/mod.rs:43-69 on the documented layout (var.rs:645-661):
    x (N, C, d...) log-probabilities, t (N, d...) class indices.  The node as written zips each
    outer slice of x with the whole target (only shape-consistent when N == C); its vectors
    (nll/test.rs:11-27: target [2,0,4], x (3,5), mean 1.52222) pin the documented meaning built here.
    Mean divides by len_of(Axis(0)) (:64)."""
    cls = _rust_as_usize(t)
    ok = cls < x.shape[1]
    picked = np.take_along_axis(x, np.expand_dims(np.where(ok, cls, 0), 1), axis=1)[:, 0]
    s = -np.where(ok, picked, 0).sum(dtype=x.dtype)
    return s / x.dtype.type(x.shape[0]) if reduction == "mean" else s


def nll_backward(x_grad, g, t, reduction="mean"):
    """`NegativeLogLikelihoodBackward::backward` :104-137: grad -= g * [class match] (/ target.len() for Mean, :114)."""
    cls = _rust_as_usize(t)
    ok = cls < x_grad.shape[1]
    v = x_grad.dtype.type(g) / x_grad.dtype.type(t.size) if reduction == "mean" else x_grad.dtype.type(g)
    upd = np.zeros_like(x_grad)
    np.put_along_axis(upd, np.expand_dims(np.where(ok, cls, 0), 1), np.expand_dims(np.where(ok, v, 0).astype(x_grad.dtype), 1), axis=1)
    x_grad -= upd


# --------------------------------------------------------------------------------------------
# matrix-vector / vector-matrix / vector-vector products (row f-4)
# --------------------------------------------------------------------------------------------
def mv_forward(a, x, out):
    """`MatrixVectorMul::forward` matrix_vector_mul/mod.rs:31-41."""
    out[...] = a @ x


def mv_backward(a_grad, x_grad, g, a, x):
    """BackwardLeft :63-69 (dA += g (x) x), BackwardRight :92-102 (dx += A^T g); either may be None."""
    if a_grad is not None:
        a_grad += np.outer(g, x)
    if x_grad is not None:
        x_grad += a.T @ g


def vm_forward(v, b, out):
    """`VectorMatrixMul::forward` vector_matrix_mul/mod.rs:31-41."""
    out[...] = v @ b


def vm_backward(v_grad, b_grad, g, v, b):
    """BackwardLeft :63-73 (dv += B g), BackwardRight :95-101 (dB += v (x) g)."""
    if v_grad is not None:
        v_grad += b @ g
    if b_grad is not None:
        b_grad += np.outer(v, g)


def vv_forward(l, r):
    """`VectorVectorMul::forward` vector_vector_mul/mod.rs:31-34."""
    return l.dot(r)


def vv_backward(op_grad, other, g):
    """`VectorVectorMulBackwardUnary::backward` :57-63: d_op += other * g."""
    op_grad += other * op_grad.dtype.type(g)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_cell_forward(cell_state, hidden, x, w_ih, w_hh, b_ih, b_hh):
    """`LSTMCell::forward` neuronika-nn/src/lib.rs:512-541, composition kept as written there:
    gates = h.mm_t(W_hh) + b_hh + x.mm_t(W_ih) + b_ih; chunk k of 4 along axis 1 ->
    sigmoid, tanh, sigmoid, sigmoid (:528-533); c' = g1*c + g0*g2; h' = g3*tanh(c').  Returns (c', h')."""
    gates = hidden @ w_hh.T + b_hh + x @ w_ih.T + b_ih
    hsz = gates.shape[1] // 4
    g = [gates[:, k * hsz:(k + 1) * hsz] for k in range(4)]
    input_gate, forget_gate, cell_gate, output_gate = _sigmoid(g[0]), np.tanh(g[1]), _sigmoid(g[2]), _sigmoid(g[3])
    new_cell = forget_gate * cell_state + input_gate * cell_gate
    return new_cell, output_gate * np.tanh(new_cell)


def gru_cell_forward(hidden, x, w_ih, w_hh, b_ih, b_hh):
    """`GRUCell::forward` neuronika-nn/src/lib.rs:602-624: r = sigmoid(hg0+ig0); z = sigmoid(hg1+ig1);
    n = tanh(ig2 + hg2*r); out = (h - n)*z + n."""
    ig, hg = x @ w_ih.T + b_ih, hidden @ w_hh.T + b_hh
    hsz = hg.shape[1] // 3
    ci = [ig[:, k * hsz:(k + 1) * hsz] for k in range(3)]
    ch = [hg[:, k * hsz:(k + 1) * hsz] for k in range(3)]
    r, z = _sigmoid(ch[0] + ci[0]), _sigmoid(ch[1] + ci[1])
    n = np.tanh(ci[2] + ch[2] * r)
    return (hidden - n) * z + n


def numeric_grad(f, x, eps=1e-6):
    """Central finite differences of a scalar function of an f64 array (test helper for composed modules)."""
    g = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        old = x[i]
        x[i] = old + eps; fp = f()
        x[i] = old - eps; fm = f()
        x[i] = old
        g[i] = (fp - fm) / (2 * eps)
    return g


def _heads_split(t, batch, heads):  # (B*S, d) -> (B*H, S, dh)   chunks((S,dh)) row-major order (var.rs:401-417)
    bs, d = t.shape
    s, dh = bs // batch, d // heads
    return np.ascontiguousarray(t.reshape(batch, s, heads, dh).transpose(0, 2, 1, 3)).reshape(batch * heads, s, dh)


def _heads_merge(t, batch, heads):  # cat(axis 1) per batch then cat(axis 0)  (var.rs:564-584)
    bh, s, dh = t.shape
    return np.ascontiguousarray(t.reshape(batch, heads, s, dh).transpose(0, 2, 1, 3)).reshape(batch * s, heads * dh)


def attention_core_forward(q, k, v, heads, batch, p, noise):
    """The per-(sample, head) chain of the composed MHA on the (B*S, H*dh) projection layout: a-2 `mm_t`
    (node/matrix_matrix_mul_t/mod.rs:31-41), a-4 scalar Multiplication, a-7 Softmax (node/softmax/mod.rs:37-53), a-8
    Dropout (node/dropout/mod.rs:53-79), a-1 `mm`, with Chunk / MultiConcatenate as the head split and merge.
    `noise` is (batch*heads, S, S).  Returns (context, cache for attention_core_backward)."""
    dt = q.dtype
    scale = dt.type(1.0 / np.sqrt(q.shape[1] // heads))
    qh, kh, vh = (_heads_split(t, batch, heads) for t in (q, k, v))
    sc = np.matmul(qh, kh.transpose(0, 2, 1))
    scs = sc * scale
    pr = np.zeros_like(scs)
    softmax_forward(scs, pr, axis=2)
    pd = np.zeros_like(pr)
    dropout_forward(pr, pd, noise, p, True)
    o = _heads_merge(np.matmul(pd, vh), batch, heads)
    return o, dict(qh=qh, kh=kh, vh=vh, scores=sc, probs=pr, dropped=pd, noise=noise, p=p, scale=scale, heads=heads, batch=batch)


def attention_core_backward(cache, g):
    """Backward nodes of attention_core_forward for the context gradient `g`: MatrixMatrixMulBackward{Left,Right},
    DropoutBackward (mask only, node/dropout/mod.rs:113-128), SoftmaxBackward (node/softmax/mod.rs:84-104),
    MultiplicationBackwardLeft, MatrixMatrixMulTBackward{Left,Right}.  Returns dict(d_scores, dq, dk, dv)."""
    c = cache
    doh = _heads_split(g, c["batch"], c["heads"])
    dpd = np.matmul(doh, c["vh"].transpose(0, 2, 1))
    dvh = np.matmul(c["dropped"].transpose(0, 2, 1), doh)
    dpr = np.zeros_like(dpd); dropout_backward(dpr, dpd, c["noise"], c["p"], True)
    dscs = np.zeros_like(dpd); softmax_backward(dscs, dpr, c["probs"], axis=2)
    dsc = dscs * c["scale"]
    m = lambda t: _heads_merge(t, c["batch"], c["heads"])
    return dict(d_scores=dsc, dq=m(np.matmul(dsc, c["kh"])), dk=m(np.matmul(dsc.transpose(0, 2, 1), c["qh"])), dv=m(dvh))


def mha_forward_backward(x, wq, bq, wk, bk, wv, bv, wo, bo, heads, batch, p, noise, g_out):
    """The composed MHA of SURVEY.md section 8a (module absent from the reference; oracle =
    composition of a-2, a-4, a-7, a-8, a-1, a-10):
      Q,K,V = x.mm_t(W)+b; per (b,h): P = dropout(softmax((Q_bh.mm_t(K_bh))*dh^-1/2, axis=1));
      O_bh = P.mm(V_bh); out = cat(O).mm_t(Wo)+bo.
    x is a differentiable leaf here.  `noise` has shape (batch*heads, S, S).  Returns out and
    a dict of gradients."""
    dt = x.dtype
    bs, d = x.shape
    s = bs // batch
    dh = d // heads
    scale = dt.type(1.0 / np.sqrt(dh))
    q, k, v = linear_forward(x, wq, bq), linear_forward(x, wk, bk), linear_forward(x, wv, bv)

    o, cache = attention_core_forward(q, k, v, heads, batch, p, noise)
    out = linear_forward(o, wo, bo)

    # backward
    g = g_out
    dbo = np.zeros_like(bo); accumulate(dbo, g)
    dwo = np.zeros_like(wo); mm_t_backward_right(dwo, g, o)
    do = np.zeros_like(o); mm_t_backward_left(do, g, wo)
    core = attention_core_backward(cache, do)
    dq, dk, dv = core["dq"], core["dk"], core["dv"]
    grads = {}
    dx = np.zeros_like(x)
    for name, w, b, dz in (("q", wq, bq, dq), ("k", wk, bk, dk), ("v", wv, bv, dv)):
        db = np.zeros_like(b); accumulate(db, dz)
        dw = np.zeros_like(w); mm_t_backward_right(dw, dz, x)
        mm_t_backward_left(dx, dz, w)
        grads["w" + name], grads["b" + name] = dw, db
    grads.update(wo=dwo, bo=dbo, x=dx)
    return out, grads
