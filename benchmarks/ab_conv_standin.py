"""Does a block that walks several output tiles (chunk loop: next tile's first loads in front of the last MFMA block, stores
draining under the next tile) help the conv-shaped GEMMs?  Dense stand-ins of the C3 passes through nk_sgemm with
NK_GEMM_FORCE's chunk field: forward NN 128 x 401408 x 576 (18 k-tiles), input gradient NN 64 x 401408 x 1152 (36 k-tiles).
    python benchmarks/ab_conv_standin.py"""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shapes = [(0, 0, 128, 401408, 576, "fwd stand-in NN 128x401408x576", "2,2"), (0, 0, 64, 401408, 1152, "bwd-input stand-in NN 64x401408x1152", "1,2")]
for sh in shapes:
    for name, force in (("rules", None), ("chunk1 pf99", f"{sh[6]},1,1,8,99"), ("chunk2 pf99", f"{sh[6]},1,2,8,99"), ("chunk3 pf99", f"{sh[6]},1,3,8,99"),
                        ("chunk1 pf8", f"{sh[6]},1,1,8,8"), ("chunk1 pf16", f"{sh[6]},1,1,8,16")):
        env = dict(os.environ)
        env.pop("NK_GEMM_FORCE", None)
        if force:
            env["NK_GEMM_FORCE"] = force
        r = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "ab_force.py"), *map(str, sh[:5])], env=env, capture_output=True, text=True)
        print(json.dumps({"shape": sh[5], "variant": name, "tflops": float(r.stdout.strip() or -1)}), flush=True)
