"""One kernel family per invocation, a few launches with nothing else around them: the target of the rocprofv3 PMC passes
(tools/pmc_profile.sh).

    python benchmarks/pmc_targets.py proj_fwd | gemm4k_k1024 | gemm4k | gemm2k | scores | context | attn_fwd | attn_fwd_nodrop | attn_bwd |
                                     conv_fwd | conv_bwd_input | conv_bwd_input_on_padded | conv_bwd_kernel
"""
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import rand  # noqa: E402

what = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = c.Device(0)
if what == "proj_fwd":
    M, N, K = 32768, 1024, 1024
    X, W, Y = rand(dev, (M, K), 0), rand(dev, (N, K), 1), dev.zeros((M, N))
    f = lambda: c.mm_t_fwd(dev, X, W, Y)
elif what in ("gemm4k_k1024", "gemm4k", "gemm2k"):   # gemm2k: k-pair blocks (NK_GEMM_KPAIR=0 for the plain one-block-per-CU launch)
    M = N = 2048 if what == "gemm2k" else 4096
    K = 1024 if what == "gemm4k_k1024" else M
    X, W, Y = rand(dev, (M, K), 0), rand(dev, (N, K), 1), dev.zeros((M, N))
    f = lambda: c.mm_t_fwd(dev, X, W, Y)
elif what in ("scores", "context"):
    BH, S, D = 512, 1024, 64
    if what == "scores":
        Q, Kk, SC = rand(dev, (BH, S, D), 0), rand(dev, (BH, S, D), 1), dev.zeros((BH, S, S))
        f = lambda: c.sgemm_batched(dev, 0, 1, S, S, D, 1.0, Q, D, S * D, 0, Kk, D, S * D, 0, 0.0, SC, S, S * S, 0, BH, 1)
    else:
        P, V, O = rand(dev, (BH, S, S), 2, 0, 1), rand(dev, (BH, S, D), 1), dev.zeros((BH, S, D))
        f = lambda: c.sgemm_batched(dev, 0, 0, S, D, S, 1.0, P, S, S * S, 0, V, D, S * D, 0, 0.0, O, D, S * D, 0, BH, 1)
elif what in ("attn_fwd", "attn_fwd_nodrop", "attn_bwd"):
    B, S, H, dh = 32, 1024, 16, 64
    p = 0.0 if what == "attn_fwd_nodrop" else 0.1
    Q, Kk, V, G = (rand(dev, (B * S, H * dh), i, -0.5, 0.5) for i in range(4))
    scores, dS, Pd = dev.zeros((B * H, S, S)), dev.zeros((B * H, S, S)), dev.zeros((B * H, S, S))
    stats, bits, out = dev.zeros((B * H, S, 2)), dev.zeros((B * H, S, S // 32)), dev.zeros((B * S, H * dh))
    dQ, dK, dV = dev.zeros((B * S, H * dh)), dev.zeros((B * S, H * dh)), dev.zeros((B * S, H * dh))
    fwd = lambda: c.attention_fwd(dev, Q, Kk, V, scores, stats, bits, out, B, S, H, dh, 0.125, p, True, 7, 0)
    if what == "attn_bwd":
        fwd()
        f = lambda: c.attention_bwd(dev, dQ, dK, dV, dS, Pd, G, out, scores, stats, bits, Q, Kk, V, B, S, H, dh, 0.125, p, True, (True, True, True))
    else:
        f = fwd
elif what == "conv_s2_bwd_input":   # round 6: the fused-phase input gradient of the 3x3 stride-2 layer 64 -> 128 at 56 x 56 (nk_conv_s2dx.h)
    batch = 128
    k = 1.0 / np.sqrt(576.0)
    W = rand(dev, (128, 64, 3, 3), 1, -k, k)
    G = rand(dev, (batch, 128, 28, 28), 2, 0, 1)
    DX = dev.zeros((batch, 64, 56, 56))
    f = lambda: c.conv_bwd_input(dev, DX, G, W, (2, 2), (1, 1), 1, assign=True, padding=(1, 1))
else:
    batch = 128
    x = rand(dev, (batch, 64, 56, 56), 0, 0, 1)
    k = 1.0 / np.sqrt(576.0)
    W = rand(dev, (128, 64, 3, 3), 1, -k, k)
    XP = dev.zeros((batch, 64, 58, 58))
    Y, G = dev.zeros((batch, 128, 56, 56)), rand(dev, (batch, 128, 56, 56), 2, 0, 1)
    DXP, DX, DW = dev.zeros(XP.shape), dev.zeros(x.shape), dev.zeros(W.shape)
    c.pad_const_fwd(dev, x, XP, (1, 1), 0.0)
    f = {"conv_fwd": lambda: c.conv_fwd(dev, XP, W, Y, (1, 1), (1, 1), 1),
         # what the C3 module step launches: the columns are the 56 x 56 UNPADDED input positions (nk_conv_bwd_input_padded), the gradient
         # lands in the caller's tensor; conv_bwd_input_on_padded: the two-kernel form's convolution half, 58 x 60 columns per plane
         "conv_bwd_input": lambda: c.conv_bwd_input(dev, DX, G, W, (1, 1), (1, 1), 1, assign=True, padding=(1, 1)),
         "conv_bwd_input_on_padded": lambda: c.conv_bwd_input(dev, DXP, G, W, (1, 1), (1, 1), 1),
         "conv_bwd_kernel": lambda: c.conv_bwd_kernel(dev, DW, G, XP, (1, 1), (1, 1), 1)}[what]
for _ in range(reps):
    f()
dev.sync()
print("ok", what)
