"""Round 3: k-pair GEMM blocks (512 threads, two wave groups on the two halves of the reduction, accumulators added through LDS)
against the one-block-per-CU unsplit launch and split-K + second pass, same box, one process per point:
    python benchmarks/ab_kpair.py            -> one JSON line per (shape, variant), TFLOP/s
NK_GEMM_KPAIR = 0 never / 1 lock-step groups / 2 group 1 half a k-tile out of phase;
(alternate k-tiles per group instead of halves: measured equal, 125.1 / 127.4 / 123.0 vs 125.0 / 127.4 / 122.8 at 2048^3, not kept) NK_GEMM_FORCE as in ab_force.py."""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shapes = [(0, 0, 2048, 2048, 2048, "2048^3 NN"), (0, 1, 2048, 2048, 2048, "2048^3 NT"), (1, 0, 2048, 2048, 2048, "2048^3 TN"),
          (0, 1, 1024, 4096, 4096, "1024x4096x4096 NT"), (1, 0, 1024, 1024, 32768, "proj dW TN 1024x1024x32768"),
          (0, 1, 4096, 4096, 4096, "4096^3 NT"), (0, 0, 1024, 1024, 1024, "1024^3 NN")]
variants = [("rules, no pair", {"NK_GEMM_KPAIR": "0"}),
            ("pair lock-step", {"NK_GEMM_KPAIR": "1"}),
            ("pair skewed", {"NK_GEMM_KPAIR": "2"}),
            ("split 2 + second pass", {"NK_GEMM_KPAIR": "0", "NK_GEMM_FORCE": "2,2,2"})]
extra = {"proj dW TN 1024x1024x32768": [("split 8, no pair", {"NK_GEMM_KPAIR": "0", "NK_GEMM_FORCE": "2,2,8"}),
                                       ("split 4 x pair skewed", {"NK_GEMM_KPAIR": "2", "NK_GEMM_FORCE": "2,2,4"}),
                                       ("split 2 x pair skewed", {"NK_GEMM_KPAIR": "2", "NK_GEMM_FORCE": "2,2,2"})],
         "1024^3 NN": [("128x128 split 2 x pair skewed", {"NK_GEMM_KPAIR": "2", "NK_GEMM_FORCE": "2,2,2"}),
                       ("128x128 split 4, no pair", {"NK_GEMM_KPAIR": "0", "NK_GEMM_FORCE": "2,2,4"})]}
for sh in shapes:
    for name, add in variants + extra.get(sh[5], []):
        env = dict(os.environ)
        env.pop("NK_GEMM_FORCE", None)
        env.update(add)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "ab_force.py"), *map(str, sh[:5])], env=env, capture_output=True, text=True)
        print(json.dumps({"shape": sh[5], "variant": name, "tflops": float(r.stdout.strip() or -1)}), flush=True)
