#!/usr/bin/env python3
"""Benchmark of the neuronika HIP backend on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json's metric, "training-step samples/sec (fwd+bwd+allreduce)", on
the configuration it is quoted on (configs[3], "C4"): a 3-layer MLP, hidden = 4096,
batch = 4096 per GPU, synthetic f32 data, data-parallel over the N GPUs of one node with a
RCCL sum all-reduce of the parameter gradients overlapped with backward on a side stream.
One "step" = forward, re-zero the intermediate gradients, backward(seed = 1/N) with the
overlapped all-reduce, join, SGD step and zero_grad — the whole training step through the
C++ tape mirror (host/neuronika.hpp) and the C-ABI HIP library; nothing is skipped.

One JSON line is printed by rank 0.  `value` = samples/s of the whole job.  `roofline` is for
the dominant kernel (the f32 MFMA GEMM): achieved = algorithmic flop of the GEMM launches in
the timed region / their summed HIP-event durations (events on the compute stream, recorded
inside the library around every launch).  `cpu_baseline` = the CPU oracle (a NumPy/OpenBLAS
restatement of the reference's ndarray path; the Rust reference cannot be built here) timed
on this host on a bounded sample, rank 0, N = 1 only.

Other workloads (parity-test configurations, not the headline): --workload matmul | conv | mha.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL's intra-node P2P needs dmabuf IPC on this driver stack (the image exports this already; keep it for any launcher
# that builds its own environment).  Must be set before the HIP runtime is loaded.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

MFMA_F32_PEAK = 157.3e12   # /opt/skills/guides/MI355X_MICROARCH.md: f32-in MFMA, dense
HBM_PEAK = 8.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="mlp", choices=["mlp", "matmul", "conv", "mha"])
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=4096, help="rows per GPU (mlp)")
    ap.add_argument("--n", type=int, default=4096, help="matrix size (matmul)")
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+bwd+allreduce only (metric's literal definition)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


class Dist:
    """Control plane only (rendezvous, barrier, max over ranks, broadcasting the RCCL id): plain
    TCP over MASTER_ADDR:MASTER_PORT (neuronika_amd/rendezvous.py).  torch is deliberately NOT
    imported in this process: its wheel bundles a second HIP/HSA runtime (see rendezvous.py).
    The data path (gradient all-reduce) is RCCL inside the HIP library."""

    def __init__(self, want):
        from neuronika_amd.rendezvous import Rendezvous
        self.rv = Rendezvous()
        self.rank, self.world, self.local = self.rv.rank, self.rv.world, self.rv.local
        if "NK_BENCH_FORCE_DEVICE" in os.environ:   # debugging aid: several ranks on one GPU (RCCL permitting)
            self.local = int(os.environ["NK_BENCH_FORCE_DEVICE"])
        if want != self.world and self.rank == 0:
            print(f"[bench] --gpus {want} but WORLD_SIZE={self.world}: running on {self.world} process(es)", file=sys.stderr)

    def barrier(self):
        self.rv.barrier()

    def bcast_bytes(self, b):
        return self.rv.broadcast(b)

    def max(self, v: float) -> float:
        return self.rv.max(v)

    def close(self):
        self.rv.close()


def device_sync(tdev):
    """Drain both streams of this rank's device (our equivalent of torch.cuda.synchronize(): the
    work runs on the library's own HIP streams, which torch does not see)."""
    tdev.sync()


def timed_steps(dist, tdev, cdev, step, steps, warmup):
    """W untimed + exactly K timed steps, barrier + sync on both sides; returns (host seconds for
    the K steps, max over ranks), device-event ms, and the GEMM/conv launch statistics."""
    from neuronika_amd import capi
    for _ in range(warmup):
        step()
    device_sync(tdev)
    dist.barrier()
    e0, e1 = cdev.event(), cdev.event()
    cdev.profile_begin()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    device_sync(tdev)
    dist.barrier()
    dt = time.perf_counter() - t0
    gemm = cdev.profile_end(capi.KERNEL_SGEMM)
    conv = cdev.profile_end(capi.KERNEL_CONV)
    ev_ms = e0.elapsed_ms(e1)
    return dist.max(dt), ev_ms, gemm, conv


def read_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/), if any."""
    p = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        return json.load(open(p)).get(kernel)
    except Exception:
        return None


def cpu_baseline_mlp(hidden, sample_rows):
    """The CPU oracle timed on this host: one full training step (fwd+bwd) of the same MLP on
    `sample_rows` rows, 1 BLAS thread = the reference's default (single-threaded matrixmultiply
    sgemm for mm/mm_t, neuronika-variable/Cargo.toml:25-29)."""
    from threadpoolctl import threadpool_limits
    from oracle import neuronika_oracle as O
    rng = np.random.default_rng(0)
    k = 1.0 / np.sqrt(hidden)
    x, t = rng.random((sample_rows, hidden), dtype=np.float32), rng.random((sample_rows, hidden), dtype=np.float32)
    params = [((rng.random((hidden, hidden), dtype=np.float32) * 2 - 1) * k, (rng.random(hidden, dtype=np.float32) * 2 - 1) * k) for _ in range(3)]
    with threadpool_limits(limits=1):
        O.mlp_step(x[:64], t[:64], params)  # warm-up
        reps, t0 = 0, time.perf_counter()
        while True:                           # whole steps until >= 10 s of CPU work (bounded sample of the same workload)
            O.mlp_step(x, t, params)
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= 10.0 or reps >= 8:
                break
    return {"value": round(reps * sample_rows / dt, 2), "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"{reps} fwd+bwd step(s) of the same 3x Linear({hidden},{hidden}) MLP on {sample_rows} of the 4096 rows, "
                      f"NumPy/OpenBLAS oracle, 1 BLAS thread, {dt:.2f} s; host has {os.cpu_count()} cores"}


def run_mlp(a, dist):
    import neuronika_amd
    from neuronika_amd import capi
    t = neuronika_amd.tape
    tdev = t.Device(dist.local)
    cdev = capi.Device(handle=tdev.raw())
    H, B, world = a.hidden, a.batch, dist.world
    rng = np.random.default_rng(100 + dist.rank)
    x = rng.random((B, H), dtype=np.float32)
    tgt = np.random.default_rng(200 + dist.rank).random((B, H), dtype=np.float32)
    lins = [t.nn.Linear(tdev, H, H, seed) for seed in (1, 3, 5)]      # identical weights on every rank
    X, T = t.from_ndarray(tdev, x), t.from_ndarray(tdev, tgt)
    out = lins[2].forward(lins[1].forward(lins[0].forward(X).relu()).relu())
    loss = out.mse(T, t.Reduction.Mean)
    params = []
    for lin in lins:
        params += [lin.weight, lin.bias]
    opt = t.optim.SGD(1e-3)
    for p in params:
        opt.register(p)
    comm = sync = None
    if world > 1:
        uid = dist.bcast_bytes(t.dp.Communicator.unique_id() if dist.rank == 0 else None)
        comm = t.dp.Communicator(tdev, world, dist.rank, uid)
        sync = t.dp.GradientSync(comm, params)
    seed = 1.0 / world

    def step():
        loss.forward()
        loss.no_grad(); loss.with_grad()          # drop + re-create (zero) the intermediate gradients
        if sync is not None:
            loss.backward_sync(seed, sync)
            sync.join()
        else:
            loss.backward(seed)
        if not a.no_optimizer:
            opt.step()
        opt.zero_grad()

    dt, ev_ms, gemm, _ = timed_steps(dist, tdev, cdev, step, a.steps, a.warmup)
    loss_val = loss.item()
    n_launch, gemm_ms, gemm_flop = gemm
    res = None
    if dist.rank == 0:
        achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        res = {
            "metric": "training-step samples/sec (fwd+bwd+allreduce)", "value": round(B * world * a.steps / dt, 2),
            "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C4: 3-layer MLP Linear({H},{H})x3 + ReLU, MSE mean, batch {B}/GPU, data-parallel "
                                   f"gradient all-reduce (RCCL, side stream)", "global_batch": B * world,
                       "parallelism": f"dp{world}", "optimizer_step_in_timed_region": not a.no_optimizer,
                       "allreduce_bytes_per_step": 0 if world == 1 else 3 * (H * H + H) * 4},
            "roofline": {"bound": "mfma", "kernel": "sgemm_kernel (f32 MFMA 32x32x2, 128x128x32 tiles)",
                         "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved * 1e12 / MFMA_F32_PEAK, 4), "traffic": read_traffic("sgemm_kernel"),
                         "launches": n_launch, "avg_launch_ms": round(gemm_ms / max(1, n_launch), 4),
                         "algorithmic_flop_per_launch": gemm_flop / max(1, n_launch)},
            "device_ms_per_step": round(ev_ms / a.steps, 4), "loss": loss_val,
            "gemm_share_of_step": round(gemm_ms / ev_ms, 4) if ev_ms > 0 else None,
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_mlp(H, 2048 if H >= 2048 else B)
    if comm is not None:
        tdev.sync()
    return res


def run_matmul(a, dist):
    """C2: C = A.B forward + dA += G.B^T, dB += A^T.G, f32 N x N."""
    from neuronika_amd import capi as c
    dev = c.Device(dist.local)
    n = a.n
    mk = lambda s: dev.array(np.random.default_rng(s).random((n, n), dtype=np.float32))
    A, B, G = mk(0), mk(1), mk(2)
    Cm, dA, dB = dev.zeros((n, n)), dev.zeros((n, n)), dev.zeros((n, n))

    def step():
        c.mm_fwd(dev, A, B, Cm); c.mm_bwd_left(dev, dA, G, B); c.mm_bwd_right(dev, dB, A, G)

    class TD:  # adapt capi.Device to the sync interface
        def sync(self): dev.sync()
    dt, ev_ms, gemm, _ = timed_steps(dist, TD(), dev, step, a.steps, a.warmup)
    n_launch, gemm_ms, gemm_flop = gemm
    if dist.rank != 0:
        return None
    achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12
    return {"metric": "MatMul fwd+bwd TFLOP/s (MFMA %peak)", "value": round(6.0 * n ** 3 * a.steps / dt / 1e12 * dist.world, 2),
            "unit": "TFLOP/s", "n_gpus": dist.world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "config": {"workload": f"C2: matmul fwd+bwd, square N={n}", "n": n},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved * 1e12 / MFMA_F32_PEAK, 4), "traffic": read_traffic("sgemm_kernel"),
                         "launches": n_launch, "avg_launch_ms": round(gemm_ms / max(1, n_launch), 4)}}


def run_conv(a, dist):
    """C3: zero-pad(1) -> Conv2d 3x3 s1, NCHW 128x64x56x56 -> 128 channels, fwd + both backward passes."""
    import neuronika_amd
    from neuronika_amd import capi
    t = neuronika_amd.tape
    tdev = t.Device(dist.local)
    cdev = capi.Device(handle=tdev.raw())
    N = 128
    x = np.random.default_rng(0).random((N, 64, 56, 56), dtype=np.float32)
    conv = t.nn.Conv2d(tdev, 64, 128, [3, 3], [1, 1], [1, 1], [1, 1], 1, 1)
    X = t.from_ndarray(tdev, x).requires_grad()
    y = conv.forward(X)
    G = t.from_ndarray(tdev, np.random.default_rng(2).random((N, 128, 56, 56), dtype=np.float32))
    loss = (y * G).sum()

    def step():
        loss.forward()
        loss.no_grad(); loss.with_grad()
        loss.backward(1.0)
        X.zero_grad(); conv.weight.zero_grad(); conv.bias.zero_grad()

    dt, ev_ms, _, conv_stats = timed_steps(dist, tdev, cdev, step, a.steps, a.warmup)
    n_launch, ms, flop = conv_stats
    if dist.rank != 0:
        return None
    achieved = flop / (ms * 1e-3) / 1e12
    return {"metric": "Conv2d fwd+bwd samples/s", "value": round(N * a.steps / dt, 2), "unit": "samples/s",
            "n_gpus": dist.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3: pad(1) -> conv 3x3 s1 d1 g1, x 128x64x56x56 -> 128 ch, +bias, fwd+bwd-input+bwd-kernel"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved * 1e12 / MFMA_F32_PEAK, 4), "traffic": read_traffic("conv"),
                         "launches": n_launch, "avg_launch_ms": round(ms / max(1, n_launch), 4)},
            "conv_share_of_step": round(ms / ev_ms, 4)}


def run_mha(a, dist):
    """C5: composed multi-head attention d=1024 h=16 S=1024 B=32, dropout 0.1, fwd+bwd."""
    import neuronika_amd
    from neuronika_amd import capi
    t = neuronika_amd.tape
    tdev = t.Device(dist.local)
    cdev = capi.Device(handle=tdev.raw())
    B, S, d, H = 32, 1024, 1024, 16
    mha = t.nn.MultiheadAttention(tdev, d, H, 0.1, 1)
    if "NK_MHA_STRIDED" in os.environ:               # A/B aid: heads addressed in place (1) or split/merge copies (0)
        mha.strided_heads = os.environ["NK_MHA_STRIDED"] == "1"
    X = t.from_ndarray(tdev, np.random.default_rng(0).random((B * S, d), dtype=np.float32)).requires_grad()
    G = t.from_ndarray(tdev, np.random.default_rng(5).random((B * S, d), dtype=np.float32))
    loss = (mha.forward(X, B) * G).sum()
    leaves = [X] + [getattr(getattr(mha, n), w) for n in "qkvo" for w in ("weight", "bias")]

    def step():
        loss.forward()
        loss.no_grad(); loss.with_grad()
        loss.backward(1.0)
        for p in leaves:
            p.zero_grad()

    dt, ev_ms, gemm, _ = timed_steps(dist, tdev, cdev, step, a.steps, a.warmup)
    n_launch, ms, flop = gemm
    if dist.rank != 0:
        return None
    achieved = flop / (ms * 1e-3) / 1e12
    return {"metric": "MultiheadAttention fwd+bwd sequences/s", "value": round(B * a.steps / dt, 2), "unit": "sequences/s",
            "n_gpus": dist.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C5: MHA d_model=1024 heads=16 seq=1024 batch=32 dropout=0.1, composed from reference ops"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved * 1e12 / MFMA_F32_PEAK, 4), "traffic": None, "launches": n_launch,
                         "avg_launch_ms": round(ms / max(1, n_launch), 4)},
            "gemm_share_of_step": round(ms / ev_ms, 4)}


def main():
    a = parse()
    dist = Dist(a.gpus)
    try:
        res = {"mlp": run_mlp, "matmul": run_matmul, "conv": run_conv, "mha": run_mha}[a.workload](a, dist)
        if dist.rank == 0 and res is not None:
            print(json.dumps(res), flush=True)
    finally:
        dist.close()


if __name__ == "__main__":
    main()
