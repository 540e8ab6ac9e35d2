"""THE parity bound for contractions - one place, every contraction assert of the suite goes through it.

Policy (SURVEY.md 8c ii), both f32 results measured against the f64 oracle, AS THE SURVEY STATES IT:

    err_gpu <= max(2 * err_cpu32,  1e-6 * K * max|a| * max|b|)

K = contraction length.  (Rounds 4 - 5 multiplied the absolute term by sqrt(L / 2048) for device chains longer than 2048 - the
unsplit 4096 / 8192-long GEMMs.  Since round 6 the library cuts such reductions into chained launches of at most 2048
(nk_gemm.hip, GEMM_CHAIN_K) and the amendment is gone: no factor, no exception list.)  A GEMM with an epilogue
(alpha, beta * C, bias) is a contraction followed by elementwise work: the contraction term scales with |alpha| and the
stated elementwise tolerance (rtol 1e-5, atol 1e-6) is added for the epilogue's own roundings.

No test spells a contraction tolerance by hand: `grep -nE "[0-9]e-6 \\* ?(K|n|m|o|[0-9])" tests/` finds only this file's
docstring and sums (not products).  Every call records its margin (tests/conftest.py -> gpurun_out/tolerance_margins.json)."""
import numpy as np

ABS = 1e-6          # the survey's absolute coefficient
CPU_FACTOR = 2.0    # ... and its factor on the CPU restatement's own error
ELEMENTWISE_RTOL, ELEMENTWISE_ATOL = 1e-5, 1e-6


def abs_term(K, amax, bmax, scale=1.0):
    """the absolute term of the policy; `scale` = |alpha| (or any factor the product is multiplied by afterwards)"""
    return ABS * K * float(amax) * float(bmax) * abs(float(scale))


def assert_contraction(label, got, ref64, K, amax=1.0, bmax=1.0, *, cpu32=None, scale=1.0, epilogue=False):
    """got: the device's f32 result; ref64: the f64 oracle; cpu32: the f32 CPU restatement of the same product (OpenBLAS /
    the oracle's f32 twin) or None when the test has none (then only the absolute term applies).
    epilogue=True adds the elementwise tolerance on top (results that went through alpha / beta / bias arithmetic).
    Returns err_gpu / bound."""
    from conftest import record_margin
    got64 = np.asarray(got, dtype=np.float64)
    ref64 = np.asarray(ref64, dtype=np.float64)
    err_gpu = float(np.abs(got64 - ref64).max()) if got64.size else 0.0
    err_cpu = float(np.abs(np.asarray(cpu32, dtype=np.float64) - ref64).max()) if cpu32 is not None and got64.size else 0.0
    a = abs_term(K, amax, bmax, scale)
    extra = (ELEMENTWISE_ATOL + ELEMENTWISE_RTOL * float(np.abs(ref64).max())) if (epilogue and ref64.size) else 0.0
    bound = max(CPU_FACTOR * err_cpu, a) + extra
    if label:
        record_margin(label, err_gpu, err_cpu, a + extra)
    assert err_gpu <= bound, (label, err_gpu, err_cpu, a, extra)
    return err_gpu / bound if bound > 0 else 0.0
