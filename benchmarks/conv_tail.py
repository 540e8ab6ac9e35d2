"""How much the ragged last wave of blocks costs the C3-shaped passes: time per sample at batch sizes whose tile counts are
just below / just above a whole number of resident-block waves."""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import timeit, rand  # noqa: E402

dev = c.Device(0)
for N in (94, 96, 125, 128, 156, 157):
    xs, ws, ys = (N, 64, 56, 56), (128, 64, 3, 3), (N, 128, 56, 56)
    W, G = rand(dev, ws, 1), rand(dev, ys, 2)
    DX = dev.zeros(xs)
    XP = rand(dev, (N, 64, 58, 58), 0)
    Y = dev.zeros(ys)
    f_bi = lambda: c.conv_bwd_input(dev, DX, G, W, (1, 1), (1, 1), 1, assign=True, padding=(1, 1))
    f_fw = lambda: c.conv_fwd(dev, XP, W, Y, (1, 1), (1, 1), 1)
    ms_bi, ms_fw = timeit(dev, f_bi, 10), timeit(dev, f_fw, 10)
    tiles = N * 56 * 56 // 128
    print(json.dumps({"N": N, "tiles": tiles, "waves@768": round(tiles / 768, 3), "waves@512": round(tiles / 512, 3),
                      "bwd_input_us_per_sample": round(ms_bi * 1e3 / N, 3), "fwd_us_per_sample": round(ms_fw * 1e3 / N, 3)}), flush=True)
    del W, G, DX, XP, Y
