"""C3's forward (+ bias) and input gradient (padding folded in, assigning) with the Winograd F(2x2, 3x3) kernels and with the
implicit-GEMM kernels (NK_TUNE_CONV_WINOGRAD 1 / 0), alternating on one box: us per call (HIP events), the direct algorithmic
TFLOP/s each time represents, and the fraction of the f32 MFMA peak.  `python benchmarks/ab_winograd.py [N]`"""
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
    Cin, Cout, H = 64, 128, 56
    rng = np.random.default_rng(0)
    XP = dev.array(rng.random((N, Cin, H + 2, H + 2), dtype=np.float32))
    W = dev.array((rng.random((Cout, Cin, 3, 3), dtype=np.float32) * 2 - 1) / 24)
    B = dev.array(rng.random((Cout, 1, 1), dtype=np.float32))
    G = dev.array(rng.random((N, Cout, H, H), dtype=np.float32))
    Y, DX = dev.zeros((N, Cout, H, H)), dev.zeros((N, Cin, H, H))
    flop = 2.0 * N * Cout * H * H * Cin * 9
    calls = {"forward + bias": lambda: c.conv_fwd(dev, XP, W, Y, (1, 1), (1, 1), 1, bias=B),
             "input gradient (pad folded, assign)": lambda: c.conv_bwd_input(dev, DX, G, W, (1, 1), (1, 1), 1, assign=True, padding=(1, 1))}

    def time(fn, reps=20):
        e0, e1 = dev.event(), dev.event()
        for _ in range(3):
            fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.sync()
        return e0.elapsed_ms(e1) / reps * 1e3

    for _ in range(30):                              # clocks
        calls["forward + bias"]()
    if len(sys.argv) > 2 and sys.argv[2] == "dw":        # the kernel gradient: implicit GEMM against Winograd F(3x3, 2x2), dW and db in one call
        geoms = [("C3 64 -> 128 at 56 x 56", 64, 128, 56), ("64 -> 64 at 56 x 56", 64, 64, 56), ("128 -> 128 at 28 x 28", 128, 128, 28),
                 ("256 -> 256 at 14 x 14", 256, 256, 14)]
        for name, ci, co, hh in geoms:
            xp = dev.array(rng.random((N, ci, hh + 2, hh + 2), dtype=np.float32))
            gg = dev.array(rng.random((N, co, hh, hh), dtype=np.float32))
            dww, dbb = dev.zeros((co, ci, 3, 3)), dev.zeros((co, 1, 1))
            fn = lambda: c.conv_bwd_kernel_bias(dev, dww, dbb, gg, xp, (1, 1), (1, 1), 1, assign=(True, True))
            row = {"case": name, "N": N}
            for rnd in range(2):
                for label, mode in (("implicit_gemm", 0), ("winograd", 1)):
                    dev.conv_winograd(mode, None, None, mode)
                    before = dev.conv_winograd_launches()
                    row[label + "_us"] = round(time(fn), 1)
                    row[label + "_winograd_launches"] = dev.conv_winograd_launches() - before
            dev.conv_winograd(None)
            print(json.dumps(row), flush=True)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "dwfold":    # round 6: the kernel gradient with the module's padding FOLDED in (what the C3 step launches)
        for name, ci, co, hh in (("C3 64 -> 128 at 56 x 56", 64, 128, 56), ("128 -> 128 at 28 x 28", 128, 128, 28), ("256 -> 256 at 14 x 14", 256, 256, 14)):
            xx = dev.array(rng.random((N, ci, hh, hh), dtype=np.float32))
            gg = dev.array(rng.random((N, co, hh, hh), dtype=np.float32))
            dww, dbb = dev.zeros((co, ci, 3, 3)), dev.zeros((co, 1, 1))
            fn = lambda: c.conv_bwd_kernel_padded(dev, dww, gg, xx, (1, 1), (1, 1), (1, 1), 1, db=dbb, assign=(True, True))
            row = {"case": name, "N": N, "folded_us": []}
            for rnd in range(3):
                row["folded_us"].append(round(time(fn), 1))
            print(json.dumps(row), flush=True)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "shape":     # narrow (two-wave, 64-channel) against wide (four-wave, 128-channel) blocks
        geoms = [("C3 forward 64 -> 128", 64, 128, True), ("128 -> 128 at 28 x 28, forward", 128, 128, True),
                 ("128 -> 128 at 28 x 28, input gradient", 128, 128, False), ("256 -> 256 at 14 x 14, forward", 256, 256, True),
                 ("256 -> 256 at 14 x 14, input gradient", 256, 256, False)]
        for name, ci, co, fwd in geoms:
            hh = 56 if ci == 64 else (28 if ci == 128 else 14)
            xp = dev.array(rng.random((N, ci, hh + 2, hh + 2), dtype=np.float32))
            ww = dev.array((rng.random((co, ci, 3, 3), dtype=np.float32) * 2 - 1) / 24)
            gg = dev.array(rng.random((N, co, hh, hh), dtype=np.float32))
            yy, dd = dev.zeros((N, co, hh, hh)), dev.zeros((N, ci, hh, hh))
            fn = (lambda: c.conv_fwd(dev, xp, ww, yy, (1, 1), (1, 1), 1)) if fwd else \
                 (lambda: c.conv_bwd_input(dev, dd, gg, ww, (1, 1), (1, 1), 1, assign=True, padding=(1, 1)))
            row = {"case": name, "N": N}
            for rnd in range(2):
                for label, mode, shape in (("implicit_gemm", 0, None), ("narrow", 1, 0), ("wide", 1, 1)):
                    dev.conv_winograd(mode, None, shape)
                    row[label + "_us"] = round(time(fn), 1)
            dev.conv_winograd(None)
            print(json.dumps(row), flush=True)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "stagger":   # the staggered start: off, the rule, explicit units (shader clocks)
        for rnd in range(2):
            for name, fn in calls.items():
                row = {"round": rnd, "pass": name, "N": N}
                for st in (0, -1, 4000, 6000, 12000, 16000):
                    dev.conv_winograd(1, st)
                    row["stagger_%s_us" % ("rule" if st < 0 else st)] = round(time(fn), 1)
                dev.conv_winograd(None)
                print(json.dumps(row), flush=True)
        return
    for rnd in range(3):
        for name, fn in calls.items():
            row = {"round": rnd, "pass": name, "N": N}
            for mode, label in ((0, "implicit_gemm"), (1, "winograd")):
                dev.conv_winograd(mode)
                us = time(fn)
                row[label + "_us"] = round(us, 1)
                row[label + "_direct_tflops"] = round(flop / us / 1e6, 1)
                row[label + "_frac_of_peak_on_direct_flops"] = round(flop / (us * 1e-6) / PEAK, 3)
            dev.conv_winograd(None)
            row["speedup"] = round(row["implicit_gemm_us"] / row["winograd_us"], 3)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
