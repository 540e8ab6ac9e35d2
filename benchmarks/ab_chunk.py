"""Same-box sweep of the tiles-per-block chunk (NK_GEMM_FORCE's 4th field) and the look-ahead threshold (6th field) for the
projection-shaped and mid-size GEMMs: does a block that walks several output tiles (next tile's first loads issued in front
of the last MFMA block, stores draining under the next tile) beat one tile per block at 32 k-tiles?
    python benchmarks/ab_chunk.py            -> one JSON line per (shape, variant), TFLOP/s, each in its own process"""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shapes = [(0, 1, 32768, 1024, 1024, "proj fwd NT"), (0, 0, 32768, 1024, 1024, "proj dX NN"), (1, 0, 1024, 1024, 32768, "proj dW TN"),
          (0, 0, 2048, 2048, 2048, "2048^3 NN"), (0, 1, 2048, 2048, 2048, "2048^3 NT"), (0, 0, 1024, 1024, 1024, "1024^3 NN"),
          (0, 1, 4096, 4096, 4096, "4096^3 NT")]
variants = [("rules", None), ("2,2,1,2 chunk2 pf=99", "2,2,1,2,8,99"), ("2,2,1,4 chunk4 pf=99", "2,2,1,4,8,99"), ("2,2,1,1 pf=16", "2,2,1,1,8,16"),
            ("2,2,1,1 pf=24", "2,2,1,1,8,24"), ("2,2,1,1 pf=99", "2,2,1,1,8,99")]
for sh in shapes:
    for name, force in variants:
        env = dict(os.environ)
        env.pop("NK_GEMM_FORCE", None)
        if force:
            env["NK_GEMM_FORCE"] = force
        r = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "ab_force.py"), *map(str, sh[:5])], env=env, capture_output=True, text=True)
        print(json.dumps({"shape": sh[5], "variant": name, "tflops": float(r.stdout.strip() or -1)}), flush=True)
