"""CPU model of the C4 full-size parity check (tests/test_gpu_fullsize.py::test_C4_mlp_full_size): which part of
err_gpu / err_cpu32 is ReLU mask flips and which is summation order, and what K-blocked accumulation changes.

The device GEMM is modelled bit for bit in its summation order by oracle/device_order_sgemm.c (one fmaf chain per output,
folded every kc values of k).  Runs on the CPU (about a minute per variant at n = 4096):
    python tools/c4_tolerance_model.py [n]            -> prints a table, writes profiles/r04_c4_tolerance_model.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.build_c import sgemm_device_order  # noqa: E402


def rnd(seed, shape, lo=0.0, hi=1.0):
    a = np.random.default_rng(seed).random(shape, dtype=np.float32)
    return np.asarray(a * np.float32(hi - lo) + np.float32(lo), dtype=np.float32)


def run(n, mm, dt, x, t, W, masks=None):
    """mm(a, b) -> a @ b in the variant's arithmetic; masks: (m1, m2) to impose, or None for the variant's own."""
    h0, tt = x.astype(dt), t.astype(dt)
    Wd = [(w.astype(dt), b.astype(dt)) for w, b in W]
    z1 = mm(h0, Wd[0][0].T) + Wd[0][1]
    m1 = (z1 > 0) if masks is None else masks[0]
    a1 = np.where(m1, z1, 0).astype(dt)
    z2 = mm(a1, Wd[1][0].T) + Wd[1][1]
    m2 = (z2 > 0) if masks is None else masks[1]
    a2 = np.where(m2, z2, 0).astype(dt)
    z3 = mm(a2, Wd[2][0].T) + Wd[2][1]
    g3 = (2 * (z3 - tt) / dt(z3.size)).astype(dt)
    g2 = (mm(g3, Wd[2][0]) * m2).astype(dt)
    g1 = (mm(g2, Wd[1][0]) * m1).astype(dt)
    grads = [(mm(g1.T, h0), g1.sum(0)), (mm(g2.T, a1), g2.sum(0)), (mm(g3.T, a2), g3.sum(0))]
    bounds = [float(np.abs(g).max() * np.abs(a).max()) for g, a in ((g1, h0), (g2, a1), (g3, a2))]
    return grads, bounds, (m1, m2)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    x, t = rnd(100, (n, n)), rnd(200, (n, n))
    k = 1.0 / np.sqrt(n)
    W = [(rnd(s, (n, n), -k, k), rnd(s + 1, (n,), -k, k)) for s in (1, 3, 5)]
    blas = lambda a, b: a @ b
    g64, ab, m64 = run(n, blas, np.float64, x, t, W)
    out = {"n": n, "variants": {}}
    variants = {"openblas_f32": blas}
    for kc in (0, 2048, 1024, 512, 256):
        variants[f"device_order_kc{kc}"] = (lambda kc: lambda a, b: sgemm_device_order(np.ascontiguousarray(a), np.ascontiguousarray(b), kc))(kc)
    res = {}
    for name, mm in variants.items():
        for label, masks in (("own_masks", None), ("f64_masks", m64)):
            g, _, m = run(n, mm, np.float32, x, t, W, masks)
            flips = [int((a != b).sum()) for a, b in zip(m, m64)]
            errs = [(float(np.abs(dw - dw64).max()), float(np.abs(db - db64).max())) for (dw, db), (dw64, db64) in zip(g, g64)]
            res[(name, label)] = errs
            out["variants"][f"{name}:{label}"] = {"mask_flips_vs_f64": flips, "err_dW": [e[0] for e in errs], "err_db": [e[1] for e in errs]}
            print(f"{name:22s} {label:10s} flips {flips}  err_dW " + " ".join(f"{e[0]:.3e}" for e in errs), flush=True)
    out["abs_term_1e-6_K_g_a"] = [1e-6 * n * b for b in ab]
    print("abs term (1e-6 K |g||a|):", " ".join(f"{v:.3e}" for v in out["abs_term_1e-6_K_g_a"]))
    for label in ("own_masks", "f64_masks"):
        cpu = res[("openblas_f32", label)]
        for name in variants:
            if name == "openblas_f32":
                continue
            r = [res[(name, label)][i][0] / (2 * cpu[i][0]) for i in range(3)]
            ra = [res[(name, label)][i][0] / out["abs_term_1e-6_K_g_a"][i] for i in range(3)]
            stated = max(min(a, b) for a, b in zip(r, ra))
            out["variants"][f"{name}:{label}"]["dW_vs_2x_cpu32"] = r
            out["variants"][f"{name}:{label}"]["dW_vs_abs"] = ra
            out["variants"][f"{name}:{label}"]["dW_vs_stated_policy"] = stated
            print(f"{name:22s} {label:10s} dW err/(2 err_cpu32) " + " ".join(f"{v:.2f}" for v in r) + "   err/abs " + " ".join(f"{v:.2f}" for v in ra) + f"   vs stated policy {stated:.2f}")
    with open(os.path.join(ROOT, "profiles", "r04_c4_tolerance_model.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
