"""Run-to-run determinism: every reduction in the library has a fixed order (split-K slabs summed in order, two-pass
reductions, no float atomics), so repeating a launch on the same inputs must reproduce the same BITS - the property the
data-parallel replicas rely on (identical weights on every rank after identical updates)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def capi():
    from neuronika_amd import capi as c
    return c


@pytest.fixture(scope="module")
def dev():
    return capi().Device(0)


def rnd(seed, shape):
    return (np.random.default_rng(seed).random(shape, dtype=np.float32) * 2 - 1)


def twice(run):
    a = run()
    for _ in range(3):
        assert np.array_equal(a, run())


@pytest.mark.parametrize("M,N,K,ta,tb", [(1024, 1024, 8192, 0, 0), (512, 512, 4096, 1, 0), (256, 256, 2048, 0, 1),
                                         (2048, 2048, 2048, 0, 1), (300, 200, 5000, 1, 1)])
def test_gemm_split_k_bits_repeat(dev, M, N, K, ta, tb):
    c = capi()
    A, B = dev.array(rnd(1, (K, M) if ta else (M, K))), dev.array(rnd(2, (N, K) if tb else (K, N)))

    def run():
        C = dev.zeros((M, N))
        c.sgemm(dev, ta, tb, M, N, K, 1.0, A, M if ta else K, B, K if tb else N, 0.0, C, N)
        return C.numpy()
    twice(run)


@pytest.mark.parametrize("xs,ws,s,g", [((16, 64, 30, 30), (128, 64, 3, 3), (1, 1), 1), ((8, 64, 31, 29), (64, 32, 3, 3), (2, 2), 2),
                                       ((8, 20, 17, 18), (24, 20, 3, 3), (1, 1), 1), ((8, 16, 20, 20), (16, 1, 3, 3), (1, 1), 16)])
def test_conv_passes_bits_repeat(dev, xs, ws, s, g):
    c = capi()
    from oracle import neuronika_oracle as O
    ys = O.conv_out_shape(xs, ws, s, (1, 1))
    X, W, G = dev.array(rnd(3, xs)), dev.array(rnd(4, ws)), dev.array(rnd(5, ys))

    def fwd():
        Y = dev.zeros(ys); c.conv_fwd(dev, X, W, Y, s, (1, 1), g); return Y.numpy()

    def bwd_in():
        D = dev.zeros(xs); c.conv_bwd_input(dev, D, G, W, s, (1, 1), g, assign=True); return D.numpy()

    def bwd_k():
        D = dev.zeros(ws); c.conv_bwd_kernel(dev, D, G, X, s, (1, 1), g, assign=True); return D.numpy()
    twice(fwd); twice(bwd_in); twice(bwd_k)


def test_reductions_bits_repeat(dev):
    c = capi()
    x = dev.array(rnd(6, (3000, 1025)))

    def total():
        out = dev.zeros(()); c.sum_fwd(dev, x, out); return out.numpy()

    def cols():   # bias-gradient pattern: (rows, cols) -> (cols)
        d = dev.zeros((1025,)); c.unbroadcast_add(dev, d, x, assign=True); return d.numpy()
    twice(total); twice(cols)
