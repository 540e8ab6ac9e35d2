"""GPU parity of the fused attention core (nk_attention_fwd / nk_attention_bwd, the per-(sample, head) chain of the
composed multi-head attention, SURVEY.md 8a note) against the oracle's node-by-node composition
`attention_core_forward/backward`, through the C ABI.

Tolerance (SURVEY.md 8c ii): the kernels and the f32 oracle are both measured against the f64 oracle fed the SAME Philox
mask; pass iff err_gpu <= max(2 * err_cpu32, 1e-6 * max|reference|) per tensor (SURVEY 8c ii as stated; measured margins:
profiles/r03_tolerance_margins.json).  The raw scores come out of the same MFMA
k-order as nk_sgemm_batched and must equal it bit for bit; the dropped-probability tensor must have exactly the mask's
zero pattern."""
import numpy as np
import pytest

from oracle import neuronika_oracle as O

pytestmark = pytest.mark.gpu


def capi():
    from neuronika_amd import capi as c
    return c


def rnd(seed, shape, lo, hi):
    a = np.random.default_rng(seed).random(shape, dtype=np.float32)
    return np.asarray(a * np.float32(hi - lo) + np.float32(lo), dtype=np.float32)


def _check(got, want64, want32, what, floor=0.0):
    scale = max(np.abs(want64).max(), floor)
    err_gpu, err_cpu = np.abs(got - want64).max(), np.abs(want32 - want64).max()
    from conftest import record_margin
    record_margin("attention_core:" + what.split("[")[0].strip(), err_gpu, err_cpu, 1e-6 * scale)
    assert err_gpu <= max(2 * err_cpu, 1e-6 * scale), (what, err_gpu, err_cpu, scale)   # SURVEY 8c (ii) as stated


def _run(dev, B, S, H, p, train, seed, offset, assign, q, k, v, g, dq0, dh=64):
    """Runs forward + backward; the (B*H, SP, SP) scratch tensors (SP = S rounded up to 32) come back cut to (B*H, S, S), the
    padded part as `pad_*` for the ragged-length checks."""
    c = capi()
    scale = float(np.float32(1.0 / np.sqrt(dh)))
    SP = c.attention_padded(S)
    Q, K, V, G = (dev.array(t) for t in (q, k, v, g))
    scores, stats, out = dev.full((B * H, SP, SP), 7.0), dev.zeros((B * H, SP, 2)), dev.zeros((B * S, H * dh))
    bits = dev.zeros((B * H, SP, SP // 32))
    c.attention_fwd(dev, Q, K, V, scores, stats, bits, out, B, S, H, dh, scale, p, train, seed, offset)
    dS, dropped, dQ = dev.full((B * H, SP, SP), 7.0), dev.full((B * H, SP, SP), 7.0), dev.array(dq0)
    dK, dV = dev.zeros((B * S, H * dh)), dev.zeros((B * S, H * dh))
    c.attention_bwd(dev, dQ, dK, dV, dS, dropped, G, out, scores, stats, bits, Q, K, V, B, S, H, dh, scale, p, train,
                    assign=(assign, True, True))
    cut = lambda t: np.ascontiguousarray(t.numpy()[:, :S, :S])
    pad = lambda t: t.numpy()[:, :S, S:]          # padded KEYS of the real queries
    return dict(scores=cut(scores), stats=stats.numpy()[:, :S], out=out.numpy(), bits=bits.numpy().view(np.uint32), d_scores=cut(dS),
                dropped=cut(dropped), dq=dQ.numpy(), dk=dK.numpy(), dv=dV.numpy(), pad_scores=pad(scores), pad_d_scores=pad(dS),
                pad_dropped=pad(dropped)), (Q, K)


# head dimension 64 (C5) at six geometries; 32 and 128 (the kernels' other two instantiations: one / four output column tiles,
# one / four staged float4 per thread and operand) at three each
# ragged sequence lengths (S % 32 != 0: padded scratch, clamped loads, -inf scores for the padded keys) for every head dimension (S = 1: test_attention_core_single_key)
@pytest.mark.parametrize("B,S,H,dh", [(2, 128, 2, 64), (1, 160, 3, 64), (3, 32, 1, 64), (1, 256, 2, 64), (2, 96, 2, 64), (1, 384, 1, 64),
                                       (2, 128, 2, 32), (1, 160, 3, 32), (1, 32, 1, 32), (2, 128, 2, 128), (1, 160, 3, 128), (1, 32, 1, 128),
                                       (2, 40, 2, 64), (3, 100, 3, 64), (2, 129, 1, 64), (2, 7, 2, 64), (2, 2, 1, 64), (1, 197, 2, 64),
                                       (2, 72, 2, 32), (2, 17, 1, 32), (2, 200, 2, 128), (3, 33, 1, 128), (1, 1000, 2, 64)])
@pytest.mark.parametrize("p,train", [(0.1, True), (0.0, True), (0.35, False), (0.5, True)])
def test_attention_core_equals_oracle(dev, B, S, H, dh, p, train):
    c = capi()
    seed, offset = 0x1234567890ABCDEF, 4242
    q, k, v, g = (rnd(s, (B * S, H * dh), -1, 1) for s in (1, 2, 3, 4))
    dq0 = rnd(9, (B * S, H * dh), -1, 1)
    got, (Q, K) = _run(dev, B, S, H, p, train, seed, offset, False, q, k, v, g, dq0, dh)
    SP = c.attention_padded(S)
    masked = train and p != 0.0
    # score (bh, r, k) takes draw (bh * SP + r) * SP + k of the shared layout (SP == S for whole tiles)
    noise = (np.ascontiguousarray(O.dropout_noise(B * H * SP * SP, p, seed, offset).reshape(B * H, SP, SP)[:, :S, :S]) if masked
             else np.ones((B * H, S, S), np.float32))
    pe = p if masked else 0.0
    ref, ref32 = {}, {}
    for dt, dst in ((np.float64, ref), (np.float32, ref32)):
        o, cache = O.attention_core_forward(q.astype(dt), k.astype(dt), v.astype(dt), H, B, pe, noise.astype(dt))
        dst.update(O.attention_core_backward(cache, g.astype(dt)), out=o, scores=cache["scores"], dropped=cache["dropped"])
    # raw scores: same MFMA reduction order as the batched GEMM -> identical bits
    ref_scores = dev.zeros((B * H, S, S))
    c.sgemm_batched(dev, 0, 1, S, S, dh, 1.0, Q, H * dh, S * H * dh, dh, K, H * dh, S * H * dh, dh, 0.0, ref_scores, S, H * S * S, S * S, B, H)
    assert np.array_equal(got["scores"], ref_scores.numpy())
    _check(got["scores"], ref["scores"], ref32["scores"], "scores")
    assert np.array_equal(got["dropped"] == 0, noise == 0)   # dropped exactly where the mask says (no probability underflows here)
    if masked:   # the packed draws the backward kernel reads: bit j of word t of a row = key 32 t + j kept
        # words are laid out [bh][query tile][key tile][query in tile] (one 128-byte line per wave and tile)
        w = got["bits"].reshape(B * H, SP // 32, SP // 32, 32).transpose(0, 1, 3, 2).reshape(B * H, SP, SP // 32)
        unpacked = ((w[..., None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(B * H, SP, SP)[:, :S, :S]
        assert np.array_equal(unpacked, noise != 0)
    if S % 32:   # padded keys: a score of -inf, probability and score gradient exactly 0
        assert np.all(np.isneginf(got["pad_scores"])) and not got["pad_d_scores"].any() and not got["pad_dropped"].any()
    # O, dQ, dK, dV are contractions over the S keys / queries: SURVEY 8c (ii)'s yardstick for a contraction is the size of the summed
    # TERMS (its K * max|a| * max|b|; here the sharper sum_k |a_k| * max|b|), not the size of the cancelling result.  It only matters
    # for long rows: at S = 1000 O reads 1.3e-7 and dQ 3.5e-8 against 2 x 4.9e-8 / 2 x 1.7e-8 of the NumPy f32 oracle (pairwise BLAS sums).
    terms = {"out": np.abs(v).max() / (1 - pe),                                            # sum_k Pd_k = 1 / (1 - p) at most
             "dq": np.abs(ref["d_scores"]).sum(2).max() * np.abs(k).max(),
             "dk": np.abs(ref["d_scores"]).sum(1).max() * np.abs(q).max(),
             "dv": np.abs(ref["dropped"]).sum(1).max() * np.abs(g).max()}
    for name in ("out", "dropped", "d_scores", "dk", "dv"):
        _check(got[name], ref[name], ref32[name], name, floor=float(terms.get(name, 0.0)))
    _check(got["dq"] - dq0, ref["dq"], ref32["dq"], "dq (accumulated)", floor=max(np.abs(dq0).max(), float(terms["dq"])))
    # row statistics (shift m2 in log2 units, 1 / sum): together with the scores they reproduce the softmax
    c1 = np.float64(np.float32(1.0 / np.sqrt(dh))) * np.log2(np.e)
    sc2 = ref["scores"] * c1
    m2, inv = got["stats"][..., 0].astype(np.float64), got["stats"][..., 1].astype(np.float64)
    assert (m2 >= sc2.max(2) - 6.0 - 1e-4).all() and (m2 <= sc2.max(2) + 1e-4).all()     # a bound within 2^6, never above the max
    z = ref["scores"] * np.float64(np.float32(1.0 / np.sqrt(dh)))
    soft = np.exp(z - z.max(2, keepdims=True)); soft /= soft.sum(2, keepdims=True)
    np.testing.assert_allclose(np.exp2(sc2 - m2[..., None]) * inv[..., None], soft, rtol=2e-5, atol=1e-9)
    # first-write form: dQ assigned, whatever the buffer held
    got2, _ = _run(dev, B, S, H, p, train, seed, offset, True, q, k, v, g, dq0, dh)
    assert np.array_equal(got2["dq"] + dq0, got["dq"]) or np.abs(got2["dq"] + dq0 - got["dq"]).max() <= 1e-6 * np.abs(dq0).max()
    _check(got2["dq"], ref["dq"], ref32["dq"], "dq (assigned)", floor=float(terms["dq"]))


@pytest.mark.parametrize("B,S,H,dh", [(2, 128, 2, 64), (1, 160, 3, 32), (2, 100, 2, 64), (1, 96, 2, 128), (3, 33, 1, 64), (1, 1024, 16, 64)])
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_attention_core_reads_packed_qkv_bit_for_bit(dev, B, S, H, dh, p):
    """nk_attention_qkv_fwd / _bwd - Q, K, V (dQ, dK, dV) as the three column blocks of ONE (B*S, 3*H*dh) array, the output of
    `nn::MultiheadAttention`'s packed projection GEMM - give the bits of nk_attention_fwd / _bwd on three separate arrays:
    output, kept scores / statistics / mask words, dS, Pd and all three input gradients, whole and ragged sequence lengths."""
    c = capi()
    seed, offset = 77, 5
    d = H * dh
    q, k, v, g = (rnd(s_, (B * S, d), -1, 1) for s_ in (11, 12, 13, 14))
    ref, _ = _run(dev, B, S, H, p, True, seed, offset, True, q, k, v, g, np.zeros((B * S, d), np.float32), dh)
    scale = float(np.float32(1.0 / np.sqrt(dh)))
    SP = c.attention_padded(S)
    QKV, G = dev.array(np.concatenate([q, k, v], axis=1)), dev.array(g)
    scores, stats, out = dev.full((B * H, SP, SP), 7.0), dev.zeros((B * H, SP, 2)), dev.zeros((B * S, d))
    bits = dev.zeros((B * H, SP, SP // 32))
    c.attention_qkv_fwd(dev, QKV, scores, stats, bits, out, B, S, H, dh, scale, p, True, seed, offset)
    dS, dropped = dev.full((B * H, SP, SP), 7.0), dev.full((B * H, SP, SP), 7.0)
    dQKV = dev.full((B * S, 3 * d), np.nan)
    c.attention_qkv_bwd(dev, dQKV, dS, dropped, G, out, scores, stats, bits, QKV, B, S, H, dh, scale, p, True, assign=True)
    cut = lambda t: np.ascontiguousarray(t.numpy()[:, :S, :S])
    assert np.array_equal(out.numpy(), ref["out"]) and np.array_equal(cut(scores), ref["scores"])
    assert np.array_equal(stats.numpy()[:, :S], ref["stats"]) and np.array_equal(bits.numpy().view(np.uint32), ref["bits"])
    assert np.array_equal(cut(dS), ref["d_scores"]) and np.array_equal(cut(dropped), ref["dropped"])
    dqkv = dQKV.numpy()
    for i, name in enumerate(("dq", "dk", "dv")):
        assert np.array_equal(dqkv[:, i * d:(i + 1) * d], ref[name]), name
    # accumulating form: onto a non-zero start
    start = rnd(15, (B * S, 3 * d), -1, 1)
    D2 = dev.array(start)
    c.attention_qkv_bwd(dev, D2, dS, dropped, G, out, scores, stats, bits, QKV, B, S, H, dh, scale, p, True, assign=False)
    np.testing.assert_allclose(D2.numpy() - start, dqkv, rtol=0, atol=2e-6 * max(1.0, float(np.abs(dqkv).max())))


@pytest.mark.parametrize("dh", [64, 32, 128])
def test_attention_core_single_key(dev, dh):
    """S = 1 (one valid key in a 32 x 32 tile, 31 padded keys and queries): P = 1, so O = V * noise / (1 - p), dV = Pd * dO, and the
    score gradient cancels - dS = P * (dP - sum dP P) is exactly 0 in the oracle, here the rounding of dP (the kernel forms the sum
    as keep * dO . O), and dQ / dK inherit it: measured against dP's size, the terms that cancel."""
    B, H, p, seed, offset = 3, 2, 0.25, 5, 0
    q, k, v, g = (rnd(s_, (B, H * dh), -1, 1) for s_ in (41, 42, 43, 44))
    got, _ = _run(dev, B, 1, H, p, True, seed, offset, True, q, k, v, g, np.zeros((B, H * dh), np.float32), dh)
    noise = O.dropout_noise(B * H * 32 * 32, p, seed, offset).reshape(B * H, 32, 32)[:, 0, 0]          # draw (bh * 32 + 0) * 32 + 0
    keep = np.repeat(noise.reshape(B, H), dh, axis=1) / (np.float32(1) - np.float32(p))
    np.testing.assert_allclose(got["out"], v * keep, rtol=2e-7, atol=0)
    np.testing.assert_allclose(got["dv"], g * keep, rtol=2e-7, atol=0)
    assert np.array_equal(got["dropped"].reshape(-1) != 0, noise != 0)
    dp_max = np.abs((g.astype(np.float64) * v).reshape(B, H, dh).sum(2)).max() / (1 - p)
    for name in ("d_scores", "dq", "dk"):
        assert np.abs(got[name]).max() <= 1e-6 * max(dp_max, 1.0), (name, np.abs(got[name]).max(), dp_max)


@pytest.mark.parametrize("dh", [64, 32, 128])
def test_attention_core_mask_is_the_row_kernels_mask(dev, dh):
    """Same seed / offset -> the fused core drops exactly the elements nk_scale_softmax_dropout_fwd drops, and its output
    equals the node-by-node device path (batched GEMM -> fused probabilities -> batched GEMM) to f32 rounding."""
    c = capi()
    B, S, H, p, seed, offset = 2, 128, 2, 0.2, 99, 17
    scale = float(np.float32(1.0 / np.sqrt(dh)))
    q, k, v, g = (rnd(s, (B * S, H * dh), -1, 1) for s in (11, 12, 13, 14))
    got, (Q, K) = _run(dev, B, S, H, p, True, seed, offset, True, q, k, v, g, np.zeros((B * S, H * dh), np.float32), dh)
    V = dev.array(v)
    sc, pd, ctx = dev.zeros((B * H, S, S)), dev.zeros((B * H, S, S)), dev.zeros((B * S, H * dh))
    d, so, po, pi = H * dh, S * H * dh, H * S * S, S * S
    c.sgemm_batched(dev, 0, 1, S, S, dh, 1.0, Q, d, so, dh, K, d, so, dh, 0.0, sc, S, po, pi, B, H)
    c.scale_softmax_dropout_fwd(dev, sc, None, pd, None, scale, p, True, seed, offset)
    c.sgemm_batched(dev, 0, 0, S, dh, S, 1.0, pd, S, po, pi, V, d, so, dh, 0.0, ctx, d, so, dh, B, H)
    assert np.array_equal(got["dropped"] == 0, pd.numpy() == 0)
    np.testing.assert_allclose(got["dropped"], pd.numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(got["out"], ctx.numpy(), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("S,p,dh", [(128, 0.2, 64), (96, 0.0, 64), (256, 0.4, 64), (128, 0.2, 32), (96, 0.3, 128), (100, 0.2, 64), (45, 0.0, 32), (70, 0.3, 128)])
def test_attention_forward_without_kept_state_is_the_same_forward(dev, S, p, dh):
    """Inference form (scores = stats = mask_bits = NULL: no (B*H, S, S) tensor exists): the output is bit-identical to
    the training-graph form's, dropout included (same Philox stream)."""
    c = capi()
    B, H, seed, offset = 2, 2, 31337, 9
    scale = float(np.float32(1.0 / np.sqrt(dh)))
    q, k, v = (rnd(s_, (B * S, H * dh), -1, 1) for s_ in (21, 22, 23))
    Q, K, V = dev.array(q), dev.array(k), dev.array(v)
    SP = c.attention_padded(S)
    scores, stats, bits = dev.zeros((B * H, SP, SP)), dev.zeros((B * H, SP, 2)), dev.zeros((B * H, SP, SP // 32))
    kept, lean = dev.zeros((B * S, H * dh)), dev.zeros((B * S, H * dh))
    c.attention_fwd(dev, Q, K, V, scores, stats, bits, kept, B, S, H, dh, scale, p, True, seed, offset)
    c.attention_fwd(dev, Q, K, V, None, None, None, lean, B, S, H, dh, scale, p, True, seed, offset)
    assert np.array_equal(kept.numpy(), lean.numpy())
    with pytest.raises(RuntimeError, match="kept together"):
        c.attention_fwd(dev, Q, K, V, scores, None, None, lean, B, S, H, dh, scale, p, True, seed, offset)


def test_attention_core_is_deterministic_at_benchmark_width(dev):
    """S = 1024 (the C5 width: 32 key tiles per block, 8 blocks per head): two runs on the same inputs agree BIT for bit in
    every output - scores, statistics, dropout bits, O, dS, Pd, dQ, dK, dV - (no atomics anywhere: the query-block
    reduction of dK / dV is two batched products in a fixed order), and a different Philox offset changes the mask."""
    B, S, H, p, seed = 1, 1024, 2, 0.1, 4711
    q, k, v, g = (rnd(s_, (B * S, H * 64), -1, 1) for s_ in (31, 32, 33, 34))
    dq0 = np.zeros((B * S, H * 64), np.float32)
    a, _ = _run(dev, B, S, H, p, True, seed, 0, True, q, k, v, g, dq0)
    b, _ = _run(dev, B, S, H, p, True, seed, 0, True, q, k, v, g, dq0)
    for name in a:
        assert np.array_equal(a[name], b[name]), name
    c2, _ = _run(dev, B, S, H, p, True, seed, 1 << 20, True, q, k, v, g, dq0)
    assert not np.array_equal(a["bits"], c2["bits"])
    keep = np.unpackbits(a["bits"].view(np.uint8)).mean()
    assert abs(keep - (1 - p)) < 2e-3          # 2 M draws: the keep rate is 1 - p to three digits


def test_attention_core_rejects_what_it_cannot_do(dev):
    c = capi()
    assert c.attention_supported(1024, 64, 0.1) and c.attention_supported(32, 64, 0.0)
    assert c.attention_supported(1024, 32, 0.1) and c.attention_supported(1024, 128, 0.1)
    assert not c.attention_supported(1024, 96, 0.1) and not c.attention_supported(1024, 256, 0.1) and c.attention_supported(100, 64, 0.1)
    assert not c.attention_supported(64, 64, 1.0, True) and c.attention_supported(64, 64, 1.0, False)
    z = dev.zeros((64, 96))
    sc, st = dev.zeros((1, 64, 64)), dev.zeros((1, 64, 2))
    with pytest.raises(RuntimeError, match="fused attention needs"):
        c.attention_fwd(dev, z, z, z, sc, st, None, z, 1, 64, 1, 96, 0.1, 0.0)
    with pytest.raises(RuntimeError, match="Wrong probability"):
        c.attention_fwd(dev, z, z, z, sc, st, None, z, 1, 64, 1, 64, 0.1, 1.5)
    z64 = dev.zeros((64, 64))
    with pytest.raises(RuntimeError, match="mask_bits buffer is needed"):
        c.attention_fwd(dev, z64, z64, z64, sc, st, None, z64, 1, 64, 1, 64, 0.1, 0.5)
    with pytest.raises(RuntimeError, match="positive finite scale"):
        c.attention_fwd(dev, z64, z64, z64, sc, st, None, z64, 1, 64, 1, 64, -0.1, 0.0)
