"""Calibration sweep: every (ti, tj, splits) candidate for a few GEMM shapes (NK_GEMM_FORCE="ti,tj,splits[,chunk]" is read by the library at run time)."""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    from neuronika_amd import capi as c
    from benchmarks.microbench import timeit, rand
    dev = c.Device(0)
    ta, tb, M, N, K = map(int, sys.argv[1:6])
    A = rand(dev, (K, M) if ta else (M, K), 0, 0, 1); B = rand(dev, (N, K) if tb else (K, N), 1, 0, 1); C = dev.zeros((M, N))
    lda, ldb = (M if ta else K), (K if tb else N)
    f = lambda: c.sgemm(dev, ta, tb, M, N, K, 1.0, A, lda, B, ldb, 1.0, C, N)
    timeit(dev, f, 30); ms = timeit(dev, f, 30)
    print(round(2.0 * M * N * K / ms / 1e9, 1))
    sys.exit(0)
shapes = [(0, 0, 1024, 1024, 1024), (0, 0, 1536, 1536, 1536), (0, 0, 2048, 2048, 2048), (0, 0, 512, 512, 8192), (1, 0, 1024, 1024, 32768),
          (0, 1, 64, 4096, 4096), (0, 0, 256, 256, 4096), (0, 0, 768, 768, 768), (0, 0, 4096, 256, 1024)]
env = dict(os.environ)
for sh in shapes:
    res = {}
    for ti, tj in ((2, 2), (2, 1), (1, 2), (1, 1)):
        if sh[2] <= 64 and ti == 2 or sh[3] <= 64 and tj == 2:
            continue
        kt = sh[4] // 32
        for sp in (1, 2, 3, 4, 6, 8, 16, 32):
            if sp > 1 and kt // sp < 4:
                continue
            blocks = -(-sh[2] // (64 * ti)) * -(-sh[3] // (64 * tj)) * sp
            if blocks > 4096 or (sp > 1 and blocks > 1024):
                continue
            r = subprocess.run([sys.executable, __file__, *map(str, sh)], env=dict(env, NK_GEMM_FORCE=f"{ti},{tj},{sp}"), capture_output=True, text=True)
            res[f"{ti}{tj}s{sp}"] = float(r.stdout.strip() or -1)
    best = max(res, key=res.get)
    print(sh, "best", best, res[best], json.dumps(res), flush=True)
