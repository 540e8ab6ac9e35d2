"""The three convolution passes over a few layer shapes of common CNNs (which kernel variant each falls into, TFLOP/s).
x is the already padded input.  One JSON line per shape."""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import timeit, rand  # noqa: E402


def conv_out_shape(xs, ws, stride, dilation):   # utils.rs:207-237
    return (xs[0], ws[0]) + tuple((i - d * (k - 1) - 1) // s + 1 for i, k, s, d in zip(xs[2:], ws[2:], stride, dilation))


dev = c.Device(0)
SHAPES = [
    # name, x shape (padded), w shape, stride, dilation, groups
    ("C3 3x3 s1 64->128 @56", (128, 64, 58, 58), (128, 64, 3, 3), (1, 1), (1, 1), 1),
    ("3x3 s1 64->64 @56", (128, 64, 58, 58), (64, 64, 3, 3), (1, 1), (1, 1), 1),
    ("3x3 s2 64->128 @56", (128, 64, 58, 58), (128, 64, 3, 3), (2, 2), (1, 1), 1),
    ("3x3 s2 128->256 @28", (128, 128, 30, 30), (256, 128, 3, 3), (2, 2), (1, 1), 1),
    ("1x1 s2 256->512 @28 (projection shortcut)", (128, 256, 28, 28), (512, 256, 1, 1), (2, 2), (1, 1), 1),
    ("stem 7x7 s2 3->64 @224", (64, 3, 230, 230), (64, 3, 7, 7), (2, 2), (1, 1), 1),
    ("1x1 s1 256->64 @56", (64, 256, 56, 56), (64, 256, 1, 1), (1, 1), (1, 1), 1),
    ("3x3 s1 256->256 @14", (128, 256, 16, 16), (256, 256, 3, 3), (1, 1), (1, 1), 1),
    ("3x3 s1 512->512 @7", (128, 512, 9, 9), (512, 512, 3, 3), (1, 1), (1, 1), 1),
    ("depthwise-ish g=32 3x3 128->128 @28", (128, 128, 30, 30), (128, 4, 3, 3), (1, 1), (1, 1), 32),
    ("depthwise 3x3 256 ch @28", (128, 256, 30, 30), (256, 1, 3, 3), (1, 1), (1, 1), 256),
    ("1-d k=9 64->64 L=4096", (64, 64, 4104), (64, 64, 9), (1,), (1,), 1),
]
only = sys.argv[1:] and sys.argv[1]
for name, xs, ws, s, d, g in SHAPES:
    if only and only not in name:
        continue
    ys = conv_out_shape(xs, ws, s, d)
    X, W, G = rand(dev, xs, 0), rand(dev, ws, 1), rand(dev, ys, 2)
    Y, DX, DW = dev.zeros(ys), dev.zeros(xs), dev.zeros(ws)
    flop = 2.0
    for v in ys:
        flop *= v
    flop *= ws[1]
    for v in ws[2:]:
        flop *= v
    out = {"shape": name, "gflop_per_pass": round(flop / 1e9, 2)}
    for key, fn in (("fwd", lambda: c.conv_fwd(dev, X, W, Y, s, d, g)),
                    ("bwd_input", lambda: c.conv_bwd_input(dev, DX, G, W, s, d, g, assign=True)),
                    ("bwd_kernel", lambda: c.conv_bwd_kernel(dev, DW, G, X, s, d, g, assign=True))):
        ms = timeit(dev, fn, 10)
        out[key] = [round(ms * 1e3, 1), round(flop / ms / 1e9, 1)]   # us, TFLOP/s
    if g > 1 and all(v == 1 for v in s) and len(xs) == 4:  # grouped: also the module's form (zero padding 1 cropped away)
        DU = dev.zeros((xs[0], xs[1], xs[2] - 2, xs[3] - 2))
        ms = timeit(dev, lambda: c.conv_bwd_input(dev, DU, G, W, s, d, g, assign=True, padding=(1, 1)), 10)
        out["bwd_input_cropped"] = [round(ms * 1e3, 1), round(flop / ms / 1e9, 1)]
    print(json.dumps(out), flush=True)
    del X, W, G, Y, DX, DW
