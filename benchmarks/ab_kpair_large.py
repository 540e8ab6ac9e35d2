"""Round 3: k-pair blocks on grids of MORE than one block per CU (where two plain 256-thread blocks share a CU anyway): the skewed
wave groups enforce the complementary phases two independent blocks only drift into.  Alternating runs, one process per point:
    python benchmarks/ab_kpair_large.py [reps]      -> one JSON line per shape: TFLOP/s of every run, plain vs pair"""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shapes = [(0, 1, 4096, 4096, 4096, "4096^3 NT"), (0, 0, 4096, 4096, 4096, "4096^3 NN"), (1, 0, 4096, 4096, 4096, "4096^3 TN"),
          (0, 1, 32768, 1024, 1024, "proj fwd NT 32768x1024x1024"), (0, 0, 32768, 1024, 1024, "proj dX NN 32768x1024x1024"),
          (0, 1, 8192, 8192, 8192, "8192^3 NT"), (0, 1, 2048, 2048, 2048, "2048^3 NT"), (0, 0, 2048, 2048, 2048, "2048^3 NN"),
          (1, 0, 2048, 2048, 2048, "2048^3 TN"), (0, 1, 3072, 3072, 3072, "3072^3 NT (576 tiles)"), (0, 1, 2560, 2560, 2560, "2560^3 NT (400 tiles)")]
for sh in shapes:
    res = {"0": [], "2": []}
    for _ in range(reps):
        for kp in ("0", "2"):
            env = dict(os.environ, NK_GEMM_KPAIR=kp)
            env.pop("NK_GEMM_FORCE", None)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "ab_force.py"), *map(str, sh[:5])], env=env, capture_output=True, text=True)
            res[kp].append(float(r.stdout.strip() or -1))
    print(json.dumps({"shape": sh[5], "plain": res["0"], "pair_skewed": res["2"]}), flush=True)
