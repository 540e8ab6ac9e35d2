"""Round 6: the fused-phase input gradient of the 3 x 3 / stride-2 convolution (csrc/nk_conv_s2dx.h) against the per-phase implicit GEMMs
it replaces (NK_TUNE_CONV_S2DX 0), with narrow and wide blocks forced (2 / 3) and by rule: us per call (HIP events), TFLOP/s, fraction of
the f32 MFMA peak.  `python benchmarks/ab_s2dx.py [N]`, one JSON line per layer shape."""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from neuronika_amd import capi as c  # noqa: E402

PEAK = 157.3e12


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = c.Device(0)
    rng = np.random.default_rng(0)

    def time(fn, reps=20):
        e0, e1 = dev.event(), dev.event()
        for _ in range(3):
            fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.sync()
        return e0.elapsed_ms(e1) / reps * 1e3

    for name, ci, co, h in (("3x3 s2 64->128 @56", 64, 128, 56), ("3x3 s2 128->256 @28", 128, 256, 28), ("3x3 s2 256->512 @14", 256, 512, 14),
                            ("3x3 s2 64->64 @112", 64, 64, 112)):
        ho = (h + 2 - 3) // 2 + 1
        W = dev.array((rng.random((co, ci, 3, 3), dtype=np.float32) * 2 - 1) / 24)
        G = dev.array(rng.random((N, co, ho, ho), dtype=np.float32))
        DX = dev.zeros((N, ci, h, h))
        fn = lambda: c.conv_bwd_input(dev, DX, G, W, (2, 2), (1, 1), 1, assign=True, padding=(1, 1))
        flop = 2.0 * N * co * ho * ho * ci * 9
        if len(sys.argv) > 2 and sys.argv[2] == "fwd":      # the forward twin (csrc/nk_conv_s2fwd.h) on the padded copy, as the module's Pad node leaves it
            XP = dev.array(rng.random((N, ci, h + 2, h + 2), dtype=np.float32))
            Y = dev.zeros((N, co, ho, ho))
            fn = lambda: c.conv_fwd(dev, XP, W, Y, (2, 2), (1, 1), 1)
        row = {"shape": name, "N": N, "gflop": round(flop / 1e9, 2), "pass": "forward" if len(sys.argv) > 2 and sys.argv[2] == "fwd" else "input gradient"}
        for rnd in range(3):
            for label, mode in (("per_phase", 0), ("rule", None), ("narrow", 2), ("wide", 3)):
                dev.conv_s2dx(mode)
                row.setdefault(label + "_us", []).append(round(time(fn), 1))
        dev.conv_s2dx(None)
        for label in ("per_phase", "rule", "narrow", "wide"):
            row[label + "_frac_of_peak"] = round(flop / (min(row[label + "_us"]) * 1e-6) / PEAK, 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
