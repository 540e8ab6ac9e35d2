"""Seeded random GEMM problems through nk_sgemm / nk_sgemm_batched: ragged and aligned shapes, leading dimensions larger
than the rows, every layout, alpha / beta, two-level batch strides, and reduction lengths on both sides of the look-ahead
threshold (48 k-tiles, even and odd counts) - against f64 NumPy under the suite's one contraction bound (tests/tolerance.py;
err_cpu32 from OpenBLAS's f32 product of the same operands)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def capi():
    from neuronika_amd import capi as c
    return c


@pytest.fixture(scope="module")
def dev():
    return capi().Device(0)


def case(seed):
    rng = np.random.default_rng(5000 + seed)
    kind = seed % 4
    if kind == 0:      # small ragged
        M, N, K = (int(rng.integers(1, 300)) for _ in range(3))
    elif kind == 1:    # aligned tiles, long reductions around the look-ahead threshold
        M, N = 128 * int(rng.integers(1, 4)), 128 * int(rng.integers(1, 4))
        K = 32 * int(rng.integers(44, 56))
    elif kind == 2:    # aligned rows, ragged K (interior tiles on unaligned 16-byte loads, last k-tile guarded)
        M, N = 64 * int(rng.integers(1, 6)), 64 * int(rng.integers(1, 6))
        K = int(rng.integers(33, 700))
    else:              # tall / wide
        M, N, K = int(rng.choice([1, 7, 64, 1000])), int(rng.choice([1, 5, 64, 777])), int(rng.integers(1, 2100))
    ta, tb = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    pad_a, pad_b, pad_c = (int(rng.choice([0, 0, 4, 3])) for _ in range(3))
    alpha = float(rng.choice([1.0, 1.0, -0.5, 2.0]))
    beta = float(rng.choice([0.0, 1.0, 0.25]))
    return M, N, K, ta, tb, pad_a, pad_b, pad_c, alpha, beta


@pytest.mark.parametrize("seed", range(120))
def test_sgemm_random_problem(dev, seed):
    c = capi()
    M, N, K, ta, tb, pa, pb, pc, alpha, beta = case(seed)
    rng = np.random.default_rng(seed)
    ar, ac = (K, M) if ta else (M, K)
    br, bc = (N, K) if tb else (K, N)
    a_full = (rng.random((ar, ac + pa), dtype=np.float32) * 2 - 1)
    b_full = (rng.random((br, bc + pb), dtype=np.float32) * 2 - 1)
    c_full = (rng.random((M, N + pc), dtype=np.float32) * 2 - 1)
    a, b = a_full[:, :ac], b_full[:, :bc]
    opa, opb = (a.T if ta else a).astype(np.float64), (b.T if tb else b).astype(np.float64)
    A, B, Cd = dev.array(a_full), dev.array(b_full), dev.array(c_full)
    c.sgemm(dev, ta, tb, M, N, K, alpha, A, ac + pa, B, bc + pb, beta, Cd, N + pc)
    got = Cd.numpy()
    ref = alpha * (opa @ opb) + beta * c_full[:, :N].astype(np.float64)
    from tolerance import assert_contraction
    cpu32 = np.float32(alpha) * ((a.T if ta else a) @ (b.T if tb else b)) + np.float32(beta) * c_full[:, :N]
    assert_contraction("sgemm_random_problem", got[:, :N], ref, K, cpu32=cpu32, scale=alpha, epilogue=True)
    assert np.array_equal(got[:, N:], c_full[:, N:])            # the padding columns of C are not touched


@pytest.mark.parametrize("seed", range(24))
def test_sgemm_batched_random_strides(dev, seed):
    """Two-level batch strides (outer x inner), operands addressed inside larger buffers - the per-head attention form."""
    c = capi()
    rng = np.random.default_rng(9000 + seed)
    bo, bi = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    M, N, K = (int(rng.choice([1, 16, 64, 100, 128])) for _ in range(3))
    ta, tb = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    beta = float(rng.choice([0.0, 1.0]))
    ar, ac = (K, M) if ta else (M, K)
    br, bc = (N, K) if tb else (K, N)
    a = (rng.random((bo, bi, ar, ac), dtype=np.float32) * 2 - 1)
    b = (rng.random((bo, bi, br, bc), dtype=np.float32) * 2 - 1)
    c0 = (rng.random((bo, bi, M, N), dtype=np.float32) * 2 - 1)
    A, B, Cd = dev.array(a), dev.array(b), dev.array(c0)
    c.sgemm_batched(dev, ta, tb, M, N, K, 1.0, A, ac, bi * ar * ac, ar * ac, B, bc, bi * br * bc, br * bc, beta, Cd, N,
                    bi * M * N, M * N, bo, bi)
    opa = np.swapaxes(a, 2, 3) if ta else a
    opb = np.swapaxes(b, 2, 3) if tb else b
    ref = opa.astype(np.float64) @ opb.astype(np.float64) + beta * c0.astype(np.float64)
    from tolerance import assert_contraction
    assert_contraction("sgemm_batched_random_strides", Cd.numpy(), ref, K, cpu32=opa @ opb + np.float32(beta) * c0, epilogue=True)


@pytest.mark.parametrize("seed", range(40))
def test_sgemm_pair_random_problem(dev, seed):
    """nk_sgemm_pair on seeded pairs - the layouts of the MatMul / MatMulT backward ((NN | NT | TN) first, TN second) and layouts
    the pair kernel does not take, tile-aligned and ragged extents, leading dimensions with padding, beta 0 / 1: whatever the
    rule decides (one launch, or two), forced one launch and forced two launches give the SAME bits as two nk_sgemm calls
    without k-pair blocks, the padding columns stay untouched, and the values match f64 NumPy."""
    c = capi()
    rng = np.random.default_rng(9000 + seed)
    unit = int(rng.choice([64, 128, 128, 32, 1]))
    dims = lambda: tuple(unit * int(rng.integers(1, 9)) if unit > 1 else int(rng.integers(1, 300)) for _ in range(3))
    probs = []
    for which in range(2):
        M, N, K = dims()
        if which == 1 or seed % 5 == 4:
            ta, tb = (1, 0) if seed % 7 else (int(rng.integers(0, 2)), int(rng.integers(0, 2)))
        else:
            ta, tb = [(0, 0), (0, 1), (1, 0)][seed % 3]
        pa, pb, pc = (int(rng.choice([0, 0, 4])) for _ in range(3))
        ar, ac = (K, M) if ta else (M, K)
        br, bc = (N, K) if tb else (K, N)
        a = rng.random((ar, ac + pa), dtype=np.float32) * 2 - 1
        b = rng.random((br, bc + pb), dtype=np.float32) * 2 - 1
        cm = rng.random((M, N + pc), dtype=np.float32) * 2 - 1
        probs.append((ta, tb, M, N, K, a, ac + pa, b, bc + pb, float(rng.choice([0.0, 1.0])), cm, N + pc, ac, bc))

    def run(mode):
        dev.gemm_pair(0 if mode == "singles" else mode); dev.gemm_kpair(0)
        try:
            bufs = [(dev.array(p[5]), dev.array(p[7]), dev.array(p[10])) for p in probs]
            if mode == "singles":
                for p, (A, B, Cd) in zip(probs, bufs):
                    c.sgemm(dev, p[0], p[1], p[2], p[3], p[4], 1.0, A, p[6], B, p[8], p[9], Cd, p[11])
            else:
                (p0, (A0, B0, C0)), (p1, (A1, B1, C1)) = zip(probs, bufs)
                c.sgemm_pair(dev, p0[0], p0[1], p0[2], p0[3], p0[4], A0, p0[6], B0, p0[8], p0[9], C0, p0[11],
                             p1[0], p1[1], p1[2], p1[3], p1[4], A1, p1[6], B1, p1[8], p1[9], C1, p1[11])
            return [b[2].numpy() for b in bufs]
        finally:
            dev.gemm_pair(None); dev.gemm_kpair(None)

    ref = None
    for mode in ("singles", 0, 1, -1):
        got = run(mode)
        if ref is None:
            ref = got
        for x, y in zip(got, ref):
            assert np.array_equal(x, y), mode
    for p, got in zip(probs, ref):
        ta, tb, M, N, K, a, lda, b, ldb, beta, cm, ldc, ac, bc = p
        opa, opb = (a[:, :ac].T if ta else a[:, :ac]).astype(np.float64), (b[:, :bc].T if tb else b[:, :bc]).astype(np.float64)
        want = opa @ opb + beta * cm[:, :N].astype(np.float64)
        from tolerance import assert_contraction
        cpu32 = (a[:, :ac].T if ta else a[:, :ac]) @ (b[:, :bc].T if tb else b[:, :bc]) + np.float32(beta) * cm[:, :N]
        assert_contraction("sgemm_pair_random_problem", got[:, :N], want, K, cpu32=cpu32, epilogue=True)
        assert np.array_equal(got[:, N:], cm[:, N:])
