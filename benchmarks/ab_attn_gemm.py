"""The batched attention GEMMs of C5 (512 heads, S = 1024, dh = 64) and the projection GEMMs, under the library's own
rules or a forced configuration (NK_GEMM_FORCE="ti,tj,splits[,chunk]"): scores / dP (K = 64, output S x S), context /
dV / dQ / dK (N = 64, K = S), projections (32768 x 1024 x 1024).

    python benchmarks/ab_attn_gemm.py                 # sweep (one process per point)
    python benchmarks/ab_attn_gemm.py scores          # one measurement with the current environment
"""
import json
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(what):
    from neuronika_amd import capi as c
    from benchmarks.microbench import timeit, rand
    dev = c.Device(0)
    BH, S, D = 512, 1024, 64
    if what in ("scores", "dP", "scores_samec"):
        Q, K, SC = rand(dev, (BH, S, D), 0), rand(dev, (BH, S, D), 1), dev.zeros((BH, S, S))
        beta = 1.0 if what == "dP" else 0.0
        sc = 0 if what == "scores_samec" else S * S     # every batch writes the SAME 4 MB: no HBM write traffic
        f = lambda: c.sgemm_batched(dev, 0, 1, S, S, D, 1.0, Q, D, S * D, 0, K, D, S * D, 0, beta, SC, S, sc, 0, BH, 1)
        flop = 2.0 * BH * S * S * D
    elif what in ("context", "dV"):
        P, V, O = rand(dev, (BH, S, S), 2, 0, 1), rand(dev, (BH, S, D), 1), dev.zeros((BH, S, D))
        ta = 0 if what == "context" else 1
        f = lambda: c.sgemm_batched(dev, ta, 0, S, D, S, 1.0, P, S, S * S, 0, V, D, S * D, 0, 0.0, O, D, S * D, 0, BH, 1)
        flop = 2.0 * BH * S * S * D
    else:
        M, N, Kk = 32768, 1024, 1024
        X, W, G = rand(dev, (M, Kk), 0), rand(dev, (N, Kk), 1), rand(dev, (M, N), 2)
        Y, dX, dW, bias = dev.zeros((M, N)), dev.zeros((M, Kk)), dev.zeros((N, Kk)), rand(dev, (N,), 3)
        f = {"proj_fwd": lambda: c.linear_fwd(dev, X, W, bias, Y),
             "proj_fwd_nobias": lambda: c.mm_t_fwd(dev, X, W, Y),
             "proj_dx": lambda: c.sgemm(dev, 0, 0, M, Kk, N, 1.0, G, N, W, Kk, 0.0, dX, Kk),
             "proj_dw": lambda: c.sgemm(dev, 1, 0, N, Kk, M, 1.0, G, N, X, Kk, 0.0, dW, Kk)}[what]
        flop = 2.0 * M * N * Kk
    timeit(dev, f, 3)
    ms = timeit(dev, f, 10)
    print(json.dumps({"op": what, "force": os.environ.get("NK_GEMM_FORCE", "rules"), "us": round(ms * 1e3, 1), "tflops": round(flop / ms / 1e9, 1)}), flush=True)


def sweep():
    if os.environ.get("NK_SWEEP") == "pf2":
        pts = [(op, f) for op in ("context", "dV", "proj_fwd", "proj_dx", "proj_dw") for f in (None, "2,1,1,1,8,0,8", "2,1,1,1,8,0,16", "2,2,1,1,8,0,16", "2,2,1,1,8,0,24")]
        pts = [(op, f) for op, f in pts if not (op in ("context", "dV") and f and f.startswith("2,2")) and not (op.startswith("proj") and f and f.startswith("2,1"))]
        for op, force in pts:
            env = dict(os.environ)
            env.pop("NK_GEMM_FORCE", None)
            if force:
                env["NK_GEMM_FORCE"] = force
            subprocess.run([sys.executable, os.path.abspath(__file__), op], env=env)
        return
    pts = [("scores", f) for f in (None, "2,2,1,1", "2,2,1,1,1", "2,2,1,8,1", "2,2,1,8,2", "2,2,1,16,1", "2,2,1,4,1", "1,2,1,8,1", "1,1,1,8,1", "1,1,1,1,1")]
    pts += [("scores_samec", f) for f in (None, "2,2,1,1", "2,2,1,8,1")]
    pts += [("dP", f) for f in (None, "2,2,1,8,1")]
    pts += [("context", f) for f in (None, "2,1,1,1,1")] + [("dV", f) for f in (None, "2,1,1,1,1")]
    pts += [(op, f) for op in ("proj_fwd", "proj_fwd_nobias", "proj_dx", "proj_dw") for f in (None,)]
    for op, force in pts:
        env = dict(os.environ)
        env.pop("NK_GEMM_FORCE", None)
        if force:
            env["NK_GEMM_FORCE"] = force
        subprocess.run([sys.executable, os.path.abspath(__file__), op], env=env)


if __name__ == "__main__":
    one(sys.argv[1]) if len(sys.argv) > 1 else sweep()
