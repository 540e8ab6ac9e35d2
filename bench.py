#!/usr/bin/env python3
"""Benchmark of the neuronika HIP backend on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N            # no launcher: spawns the N ranks itself (one process per GPU)

`--gpus N` without a launcher's WORLD_SIZE in the environment makes this process the launcher: it
checks that the node has N GPUs (error otherwise, never a silent single-rank run), spawns N
workers of itself with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank
0's JSON line and fails if any rank fails.

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
on this host on a bounded sample, rank 0, N = 1 only, in the two variants BASELINE.md section 4
names: "reference-default" (1 BLAS thread for mm / mm_t = the single-threaded matrixmultiply
sgemm of the default features, neuronika-variable/Cargo.toml:25-29; all cores for the
convolution, which is rayon batch-parallel, node/convolution/mod.rs:110-122) and
"reference+blas" (OpenBLAS on all cores = the `blas` feature).  The default line also carries
`matmul_4096` (the second half of BASELINE.json's metric: C2 at N = 4096, fwd+bwd, TFLOP/s and
fraction of the f32 MFMA peak), `rccl_ranks`, `allreduce_bytes_per_step` and `exposed_comm_ms`
(step time minus the step time of the same loop with the gradient exchange switched off).

At N = 1 the default line additionally carries every other BASELINE configuration as a sub-record of the same shape
(value, ms_per_step, roofline{achieved, frac, launches, avg_launch_ms}), SUB_STEPS timed steps each after the clock
settle phase: `matmul_1024 / _2048 / _4096 / _8192` (C2), `conv_c3` (C3) and `mha_c5` (C5, with `attention_core`) -
about 3 s of GPU time in total - so that one driver-run record holds all headline numbers.

At N > 1 the line is self-diagnosing: per-rank device ms/step (`per_rank_device_ms_per_step`, min / max), a stand-alone
all-reduce of the step's gradient bytes before the timed loop (`allreduce_alone`: ms, algorithm and bus GB/s, also at
the bucket sizes the step uses), `exposed_comm_ms`, the GEMM launch time with and without the overlapped exchange
(`gemm_contention`), and what RCCL chose (`rccl`: version, channels, algorithm / protocol per message size, parsed from
rank 0's NCCL_DEBUG=INFO log).  Every rank arms a watchdog (NK_BENCH_TIMEOUT_S, default 600 s) that dumps the Python
stack of a hung rank and exits; the self-spawning launcher has an overall timeout and relays the last 2 KB of every
rank's stderr when the job fails.

Other workloads (parity-test configurations, not the headline): --workload matmul | conv | mha.
"""
from __future__ import annotations

import argparse
import faulthandler
import json
import os
import re
import secrets
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL's intra-node P2P needs dmabuf IPC on this driver stack (the image exports this already; keep it for any launcher
# that builds its own environment).  Must be set before the HIP runtime is loaded.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

MFMA_F32_PEAK = 157.3e12   # /opt/skills/guides/MI355X_MICROARCH.md: f32-in MFMA, dense
HBM_PEAK = 8.0e12
XGMI_LINK_GBS = 153.0      # same guide: per xGMI link and direction
SUB_STEPS, SUB_WARMUP = 20, 3   # timed / warm-up steps of the C2 / C3 / C5 sub-records of the default line
T_START = time.perf_counter()


def _log(msg):
    """Progress marker on stderr (multi-rank runs, or NK_BENCH_VERBOSE): a hung run shows where it stopped."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("NK_BENCH_VERBOSE"):
        print(f"[bench r{os.environ.get('RANK', '0')} +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="mlp", choices=["mlp", "matmul", "conv", "mha"])
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=4096, help="rows per GPU (mlp)")
    ap.add_argument("--n", type=int, default=4096, help="matrix size (matmul)")
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+bwd+allreduce only (metric's literal definition)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dp-channels", type=int, default=16,
                    help="N > 1: cap RCCL at this many channels (NCCL_MAX_NCHANNELS, unless the caller set it) and tell the GEMMs "
                         "that many resident-block slots are busy while the exchange runs; 0: RCCL's own choice, GEMMs plan for an idle chip")
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
        if want != self.world:   # never degrade silently: a record with the wrong n_gpus is worthless
            raise SystemExit(f"[bench] --gpus {want} but the launcher started WORLD_SIZE={self.world} rank(s)")

    def barrier(self):
        self.rv.barrier()

    def bcast_bytes(self, b):
        return self.rv.broadcast(b)

    def max(self, v: float) -> float:
        return self.rv.max(v)

    def gather(self, v: float):
        """Every rank's value, in rank order, on every rank."""
        return self.rv.gather(v)

    def close(self):
        self.rv.close()


def device_sync(tdev):
    """Drain both streams of this rank's device (our equivalent of torch.cuda.synchronize(): the
    work runs on the library's own HIP streams, which torch does not see)."""
    tdev.sync()


def timed_steps(dist, tdev, cdev, step, steps, warmup, profile=True):
    """W untimed + exactly K timed steps, barrier + sync on both sides; returns (host seconds for
    the K steps, max over ranks), device-event ms, and the GEMM/conv launch statistics.
    `profile=False`: no HIP events around the individual launches (two event records per launch make a step of
    microsecond kernels host-bound: 1024^3 reads 94 us per fwd+bwd with them, 70 without)."""
    from neuronika_amd import capi
    t_w = time.perf_counter()
    for _ in range(warmup):
        step()
    device_sync(tdev)
    # The W warm-up steps are the contract's minimum.  An MI355X that has been idle needs ~50 ms of load before its clocks
    # settle (the same 4096^3 GEMM reads 1094 us right after a pause and 1000 us 50 ms later, benchmarks/nn_via_nt.py):
    # with millisecond steps W = 5 ends inside that ramp, so untimed steps continue until SETTLE_S of load have passed.
    # The count is the same on every rank (steps contain collectives): from the slowest rank's warm-up time.
    elapsed = dist.max(time.perf_counter() - t_w)
    per_step, extra = elapsed / max(1, warmup), 0
    for _ in range(4):   # (the first estimate of the step time includes one-off costs: refine it)
        if warmup == 0 or elapsed >= SETTLE_S:
            break
        n = min(500, int((SETTLE_S - elapsed) / max(per_step, 1e-5)) + 1)
        t1 = time.perf_counter()
        for _ in range(n):
            step()
        device_sync(tdev)
        d = dist.max(time.perf_counter() - t1)
        per_step, elapsed, extra = d / n, elapsed + d, extra + n
    EXTRA_STATS["settle_steps"] = extra
    dist.barrier()
    if os.environ.get("NK_BENCH_NO_LAUNCH_EVENTS") == "1":   # measurement aid: what the per-launch event pairs cost the step
        profile = False
    e0, e1 = cdev.event(), cdev.event()
    if profile:
        cdev.profile_begin()
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        # The roofline's per-launch durations come from HIP event pairs around the MFMA launches of the TIMED steps - of every
        # LAUNCH_EVENTS_EVERY-th one: a pair costs the stream ~10 us per launch (same box, C4 / C5 / C3 steps 8.41 / 10.76 / 1.625 ms
        # with pairs on every step, 8.35 / 10.69 / 1.606 without any: profiles/r04_launch_event_cost.txt), a quarter of that now.
        if profile:
            cdev.profile_pause(i % LAUNCH_EVENTS_EVERY != 0)
        step()
    e1.record()
    device_sync(tdev)
    dist.barrier()
    dt = time.perf_counter() - t0
    ev_ms = e0.elapsed_ms(e1)
    if not profile:
        return dist.max(dt), ev_ms, (0, 0.0, 0.0), (0, 0.0, 0.0)
    sampled = len(range(0, steps, LAUNCH_EVENTS_EVERY))
    scale = steps / max(1, sampled)   # sums are reported per timed region, as if every step had been instrumented

    def scaled(stats):
        n, ms, flop = stats
        return int(round(n * scale)), ms * scale, flop * scale
    gemm = scaled(cdev.profile_end(capi.KERNEL_SGEMM))
    conv = scaled(cdev.profile_end(capi.KERNEL_CONV))
    EXTRA_STATS["attention"] = scaled(cdev.profile_end(capi.KERNEL_ATTENTION))
    return dist.max(dt), ev_ms, gemm, conv


SETTLE_S = 0.15   # seconds of untimed load before the timed region (at least the W warm-up steps)
LAUNCH_EVENTS_EVERY = 4   # timed steps whose MFMA launches carry HIP event pairs: every 4th (timed_steps)
EXTRA_STATS = {}  # kernel classes only one workload has (the fused attention core of C5), from the last timed_steps()


def read_traffic(kernel):
    """(HBM bytes per launch, commit it was measured at) from the committed rocprofv3 PMC summary (profiles/), if any.
    The counters cannot be collected inside this run (PMC passes replay every kernel under rocprofv3): the value is the
    one tools/traffic_pmc.sh + tools/make_roofline_traffic.py wrote, and the line echoes the commit it belongs to."""
    p = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        d = json.load(open(p))
        return d.get(kernel), d.get("_measured_at_commit")
    except Exception:
        return None, None


def traffic_sources_match():
    """True when the kernel sources of THIS tree hash to what profiles/roofline_traffic.json was measured with (sha256 over
    neuronika_amd/csrc/*.{hip,h}, written by tools/make_roofline_traffic.py); None when the file carries no hash."""
    import hashlib
    try:
        want = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json"))).get("_kernel_sources_sha16")
    except Exception:
        return None
    if not want:
        return None
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "neuronika_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16] == want


# ------------------------------------------------------------------------------------------------
# cpu_baseline: the CPU oracle on this host's cores (reported beside the GPU number, not a target)
# ------------------------------------------------------------------------------------------------
def _time_cpu(fn, min_s, max_reps):
    """Whole repetitions of `fn` until >= min_s seconds of CPU work (a bounded sample); the first repetition (page
    faults, BLAS thread start-up) is dropped when others follow."""
    ts, t_start = [], time.perf_counter()
    while True:
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start >= min_s or len(ts) >= max_reps:
            break
    if len(ts) > 1:
        ts = ts[1:]
    return len(ts), sum(ts)


def _blas_threads():
    from threadpoolctl import threadpool_info
    n = [i.get("num_threads", 1) for i in threadpool_info() if i.get("user_api") == "blas"]
    return max(n) if n else 1


def cpu_baseline(unit, units_per_rep, fn, what, default_all_cores=False, min_s=6.0, scaled_from=None):
    """Both BASELINE.md section-4 variants of the oracle closure `fn` (one repetition = `units_per_rep` units).
    The flat fields describe the reference's DEFAULT build for this path; `variants` holds both."""
    from threadpoolctl import threadpool_limits
    out = []
    for name, limit in (("1 BLAS thread", 1), ("all cores (OpenBLAS)", None)):
        with threadpool_limits(limits=limit):
            cores = _blas_threads()
            reps, dt = _time_cpu(fn, min_s, 64)
        out.append({"value": round(reps * units_per_rep / dt, 3), "unit": unit, "cores": cores, "kind": "port",
                    "sample": f"{reps} x {what}, NumPy/OpenBLAS oracle, {name}, {dt:.2f} s; host has {os.cpu_count()} cores"})
    out[0]["variant"] = "reference+blas feature off (single-threaded matrixmultiply sgemm)"
    out[1]["variant"] = "reference+blas (OpenBLAS, all cores)"
    main = dict(out[1] if default_all_cores else out[0])
    main["variant"] = "reference-default: " + ("convolution is rayon batch-parallel over all cores" if default_all_cores
                                                else "mm / mm_t run on one thread")
    if scaled_from:
        main["scaled_from"] = scaled_from
    main["variants"] = out
    return main


def cpu_baseline_mlp(hidden, rows):
    from oracle import neuronika_oracle as O
    rng = np.random.default_rng(0)
    k = 1.0 / np.sqrt(hidden)
    x, t = rng.random((rows, hidden), dtype=np.float32), rng.random((rows, hidden), dtype=np.float32)
    params = [((rng.random((hidden, hidden), dtype=np.float32) * 2 - 1) * k, (rng.random(hidden, dtype=np.float32) * 2 - 1) * k) for _ in range(3)]
    return cpu_baseline("samples/s", rows, lambda: O.mlp_step(x, t, params),
                        f"fwd+bwd step of the same 3x Linear({hidden},{hidden}) MLP on all {rows} rows of the batch")


def cpu_baseline_matmul(n, min_s=6.0):
    from oracle import neuronika_oracle as O
    mk = lambda s: np.random.default_rng(s).random((n, n), dtype=np.float32)
    A, B, G = mk(0), mk(1), mk(2)
    Cm, dA, dB = np.zeros_like(A), np.zeros_like(A), np.zeros_like(A)

    def fn():
        O.mm_forward(A, B, Cm); O.mm_backward_left(dA, G, B); O.mm_backward_right(dB, G, A)
    return cpu_baseline("TFLOP/s", 6.0 * n ** 3 / 1e12, fn, f"mm forward + both backward GEMMs at N={n}", min_s=min_s)


def cpu_baseline_conv(batch=128, min_s=5.0):
    """C3 on the host.  The reference's convolution is batch-parallel under rayon on every core
    (node/convolution/mod.rs:110-122), each task a single-threaded sgemm: modelled as a pool of worker PROCESSES
    (spawned, so none inherits this process's HIP state), one BLAS thread each, the batch split evenly.  The
    single-core variant runs one worker's share in this process."""
    import concurrent.futures as cf
    import multiprocessing as mp
    from benchmarks.cpu_conv_worker import conv_chunk as _conv_chunk   # importable by name in the spawned workers
    cores = os.cpu_count() or 1
    workers = max(1, min(cores, 64, batch))
    per = batch // workers
    one = _conv_chunk((0, per))                      # warm-up + the single-core figure
    one = min(one, _conv_chunk((0, per)))
    variants = [{"value": round(per / one, 3), "unit": "samples/s", "cores": 1, "kind": "port",
                 "variant": "one core (a single rayon thread)",
                 "sample": f"pad + conv fwd + bwd-input + bwd-kernel on {per} of the {batch} samples, NumPy/OpenBLAS oracle, 1 thread, {one:.2f} s"}]
    with cf.ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as ex:
        list(ex.map(_conv_chunk, [(i, 1) for i in range(workers)]))          # start the workers, import numpy
        reps, t0 = 0, time.perf_counter()
        while True:
            list(ex.map(_conv_chunk, [(i, per) for i in range(workers)]))
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= min_s or reps >= 16:
                break
    done = reps * per * workers
    main = {"value": round(done / dt, 3), "unit": "samples/s", "cores": workers, "kind": "port",
            "variant": "reference-default: convolution is rayon batch-parallel over the cores, single-threaded sgemm per task",
            "sample": f"{reps} x the {per * workers}-sample batch (pad + conv fwd + bwd-input + bwd-kernel), {workers} worker processes x "
                      f"{per} samples, NumPy/OpenBLAS oracle with 1 BLAS thread each, {dt:.2f} s; host has {cores} cores"}
    variants.append(dict(main))
    main["variants"] = variants
    return main


def cpu_baseline_mha(b_sample, S, d, H, p, min_s=5.0):
    from oracle import neuronika_oracle as O
    rng = np.random.default_rng(0)
    x = rng.random((b_sample * S, d), dtype=np.float32)
    k = 1.0 / np.sqrt(d)
    ws = [((rng.random((d, d), dtype=np.float32) * 2 - 1) * k, (rng.random(d, dtype=np.float32) * 2 - 1) * k) for _ in range(4)]
    g = rng.random((b_sample * S, d), dtype=np.float32)
    noise = O.dropout_noise(b_sample * H * S * S, p, 7, 0).reshape(b_sample * H, S, S)   # drawn outside the timed region

    def fn():
        O.mha_forward_backward(x, ws[0][0], ws[0][1], ws[1][0], ws[1][1], ws[2][0], ws[2][1], ws[3][0], ws[3][1], H, b_sample, p, noise, g)
    return cpu_baseline("sequences/s", b_sample, fn, f"MHA fwd+bwd on {b_sample} of the 32 sequences (mask pre-drawn)", min_s=min_s,
                        scaled_from=f"{b_sample} of 32 sequences (per-sequence work is independent except the weight-gradient sums)")


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def roofline_mfma(gemm_stats, kernel, traffic_key=None):
    n_launch, ms, flop = gemm_stats
    achieved = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    traffic, at = read_traffic(traffic_key) if traffic_key else (None, None)
    out = {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
           "frac": round(achieved * 1e12 / MFMA_F32_PEAK, 4), "traffic": traffic, "launches": n_launch,
           "launch_events_on_every_nth_timed_step": LAUNCH_EVENTS_EVERY,
           "avg_launch_ms": round(ms / max(1, n_launch), 4), "algorithmic_flop_per_launch": flop / max(1, n_launch)}
    if traffic is not None:
        out["traffic_measured_at_commit"] = at
        out["traffic_kernel_sources_match"] = traffic_sources_match()   # false: a kernel source changed since the PMC passes
    return out


class _CapiSync:  # adapt capi.Device to the `sync()` interface of the tape's Device
    def __init__(self, dev): self.dev = dev
    def sync(self): self.dev.sync()


def matmul_fwd_bwd(dist, dev, n, steps, warmup):
    """C2 on device `dev` (capi.Device): C = A.B, then the node's backward dA += G.B^T, dB += A^T.G (one call: both products in
    one launch where neither fills the chip, nk_sgemm_pair's rule - 1024 and 2048; two launches at 4096 and 8192).  Returns (seconds for the K steps, max over
    ranks, timed WITHOUT per-launch events; the GEMM launch statistics of a second pass of K steps with them)."""
    from neuronika_amd import capi as c
    mk = lambda s: dev.array(np.random.default_rng(s).random((n, n), dtype=np.float32))
    A, B, G = mk(0), mk(1), mk(2)
    Cm, dA, dB = dev.zeros((n, n)), dev.zeros((n, n)), dev.zeros((n, n))

    def step():
        c.mm_fwd(dev, A, B, Cm); c.mm_bwd(dev, dA, dB, G, A, B)   # MatrixMatrixMul::forward, MatrixMatrixMulBackward::backward
    sync = _CapiSync(dev)
    dt, _, _, _ = timed_steps(dist, sync, dev, step, steps, warmup, profile=False)
    settle = EXTRA_STATS["settle_steps"]
    _, _, gemm, _ = timed_steps(dist, sync, dev, step, steps, 0)
    EXTRA_STATS["settle_steps"] = settle
    return dt, gemm


def matmul_record(dist, dev, n, steps, warmup):
    """One C2 size as a record: value = whole fwd+bwd TFLOP/s (host clock over the K timed steps, no per-launch events),
    roofline = the GEMM launches themselves (HIP event pairs around the launches of a second pass of K steps, every
    LAUNCH_EVENTS_EVERY-th step of it)."""
    dt, gemm = matmul_fwd_bwd(dist, dev, n, steps, warmup)
    tf = 6.0 * n ** 3 * steps / dt / 1e12
    roof = roofline_mfma(gemm, "sgemm_kernel" if n >= 4096 else "sgemm_kernel (forward), sgemm_pair_kernel (both backward products, one launch)",
                         "sgemm_kernel" if n == 4096 else None)
    return {"workload": f"C2: mm fwd + bwd (left and right products), N={n}, {steps} steps", "value": round(tf, 2), "unit": "TFLOP/s",
            "tflops": round(tf, 2), "frac_of_mfma_peak": round(tf * 1e12 / MFMA_F32_PEAK, 4),
            "kernel_tflops": roof["achieved"], "kernel_frac": roof["frac"], "steps": steps,
            "ms_per_step": round(dt / steps * 1e3, 4), "roofline": roof}


def measure_conv(dist, tdev, cdev, steps, warmup):
    """C3: zero-pad(1) -> Conv2d 3x3 s1, NCHW 128x64x56x56 -> 128 channels, fwd + both backward passes.  The upstream
    gradient dY is seeded directly into the convolution output's gradient (`backward_from`), so the timed step holds the
    module's own nodes only."""
    import neuronika_amd
    t = neuronika_amd.tape
    N = 128
    x = np.random.default_rng(0).random((N, 64, 56, 56), dtype=np.float32)
    conv = t.nn.Conv2d(tdev, 64, 128, [3, 3], [1, 1], t.PaddingMode.zero(), [1, 1], [1, 1], 1)
    X = t.from_ndarray(tdev, x).requires_grad()
    y = conv.forward(X)
    G = t.from_ndarray(tdev, np.random.default_rng(2).random((N, 128, 56, 56), dtype=np.float32))

    def step():
        y.forward()
        y.no_grad(); y.with_grad()
        y.backward_from(G)
        X.zero_grad(); conv.weight.zero_grad(); conv.bias.zero_grad()

    wino_before = cdev.conv_winograd_launches()
    dt, ev_ms, _, conv_stats = timed_steps(dist, tdev, cdev, step, steps, warmup)
    wino = cdev.conv_winograd_launches() - wino_before >= 3        # forward, input gradient and kernel gradient all took it
    direct = 2.0 * N * 128 * 56 * 56 * 64 * 9                       # one pass, node/convolution/mod.rs:85-123
    executed = 3 * direct * (16.0 / 36.0 if wino else 1.0)  # F(2x2, 3x3) / F(3x3, 2x2): 16 multiplies per 2x2 tile and channel pair instead of 36
    kernels = ("wino_kernel<4,1,32> (forward) / wino_kernel<2,1,16> (input gradient): Winograd F(2x2, 3x3); wino_dw_kernel (kernel gradient): "
               "Winograd F(3x3, 2x2) - all on f32 MFMA") if wino else \
        "conv_fwd_fast / conv_bwd_input_fast / conv_bwd_kernel (implicit GEMM, f32 MFMA)"
    roof = roofline_mfma(conv_stats, kernels, "conv")
    # `achieved` / `frac` price the launches on the MFMA flops they EXECUTE (with the Winograd kernels 16 / 36 of the direct
    # multiplies - a fraction of a roofline cannot exceed 1); the rate on the DIRECT algorithmic flops of the three passes (what
    # the reference's im2col GEMMs execute) stands beside it as `direct_equivalent_*` with the ratio as
    # `algorithmic_speedup`.
    roof["direct_equivalent_achieved"] = roof["achieved"]
    roof["direct_equivalent_frac_of_peak"] = roof["frac"]
    roof["direct_algorithmic_flop_per_launch"] = roof["algorithmic_flop_per_launch"]
    roof["algorithmic_speedup"] = round(3 * direct / executed, 4)
    roof["achieved"] = round(roof["achieved"] * executed / (3 * direct), 2)
    roof["frac"] = round(roof["achieved"] * 1e12 / MFMA_F32_PEAK, 4)
    roof["algorithmic_flop_per_launch"] = executed / 3
    roof["flops_quoted"] = ("MFMA flops executed: Winograd F(2x2, 3x3) / F(3x3, 2x2), 16 / 36 of the direct 3 x 2 N Cout Ho Wo Cin 9" if wino
                            else "direct algorithmic (3 x 2 N Cout Ho Wo Cin 9)")
    roof["winograd"] = bool(wino)
    return {"workload": "C3: nn::Conv2d = pad(1) -> conv 3x3 s1 d1 g1 -> + bias, x 128x64x56x56 -> 128 ch, fwd+bwd-input+bwd-kernel (Zero padding folded into the kernels)",
            "value": round(N * steps * dist.world / dt, 2), "unit": "samples/s", "steps": steps,
            "ms_per_step": round(dt / steps * 1e3, 4),
            "step_tflops": round(3 * 2.0 * N * 128 * 56 * 56 * 64 * 9 * steps / dt / 1e12, 2),
            "roofline": roof,
            "conv_share_of_step": round(conv_stats[1] / ev_ms, 4) if ev_ms > 0 else None}


def measure_hbm_kernels(cdev):
    """The HBM-bound kernels of the path, HIP events on the compute stream: algorithmic bytes (4 B x elements read + written,
    SURVEY.md 8d) / time, as a fraction of the 8.0 TB/s peak and of the ceiling of their own stream count measured in the same
    run: a device-to-device copy (1 read + 1 write), an elementwise add (2 reads + 1 write) and a fused multiply-add
    `nk_relu_bwd`-shaped triple read (3 reads + 1 write is what ReLU / dropout / MSE backward and SGD are).
    Two sizes: 1 GiB per tensor (the judged rows: nothing of a 4 - 5 GiB working set lives in the 256 MB Infinity Cache) and
    256 MB per tensor (labelled cache-assisted: rates there exceed what HBM delivers, e.g. a fill at the spec peak)."""
    from neuronika_amd import capi as c

    def timeit(fn, settle_ms=25.0, min_ms=25.0):
        e0, e1 = cdev.event(), cdev.event()
        e0.record(); calls = 0
        while True:
            fn(); fn(); calls += 2
            e1.record(); e1.sync()
            if e0.elapsed_ms(e1) >= settle_ms:
                break
        iters = max(4, int(min_ms / max(e0.elapsed_ms(e1) / calls, 1e-3)) + 1)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); e1.sync()
        return e0.elapsed_ms(e1) / iters

    def one_size(rows, L):
        n = rows * L
        rng = np.random.default_rng(0)
        X = cdev.array(rng.random((rows, L), dtype=np.float32) * 8 - 4)
        G = cdev.array(rng.random((rows, L), dtype=np.float32))
        Y, D, NZ = cdev.zeros((rows, L)), cdev.zeros((rows, L)), cdev.zeros((rows, L))
        ceilings = {}
        for name, fn, nbytes in [("fill0 (1 write)", lambda: D.fill(0.0), 4 * n),
                                 ("copy (1 read + 1 write)", lambda: c.check(c.lib.nk_copy(cdev.h, Y.p, X.p, n)), 8 * n),
                                 ("add (2 reads + 1 write)", lambda: c.binary_fwd(cdev, "add", Y, X, G), 12 * n)]:
            ms = timeit(fn)
            ceilings[name] = {"ms": round(ms, 4), "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_of_peak": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4)}
        copy_rate = ceilings["copy (1 read + 1 write)"]["GBps"] * 1e9
        add_rate = ceilings["add (2 reads + 1 write)"]["GBps"] * 1e9
        cases = [("softmax_fwd", lambda: c.softmax_fwd(cdev, X, Y, 1), 8 * n, copy_rate),
                 ("softmax_bwd", lambda: c.softmax_bwd(cdev, D, G, Y, 1), 16 * n, add_rate),
                 ("dropout_fwd", lambda: c.dropout_fwd(cdev, X, Y, NZ, 0.1, True, 7, 0), 12 * n, add_rate),
                 ("dropout_bwd", lambda: c.dropout_bwd(cdev, D, G, NZ, 0.1, True), 16 * n, add_rate),
                 ("relu_bwd", lambda: c.relu_bwd(cdev, D, G, X), 16 * n, add_rate),
                 ("sgd_multi (3 parameters)", lambda: c.sgd_step_multi(cdev, [X, Y, D], [G, G, G], None, lr=1e-12), 3 * 12 * n, add_rate)]
        kernels = {}
        for name, fn, nbytes, ceil in cases:
            ms = timeit(fn)
            rate = nbytes / (ms * 1e-3)
            kernels[name] = {"bound": "hbm", "algorithmic_bytes": nbytes, "ms": round(ms, 4), "achieved": round(rate / 1e9, 1), "unit": "GB/s",
                             "frac": round(rate / HBM_PEAK, 4), "frac_of_copy_this_run": round(rate / copy_rate, 4),
                             "frac_of_stream_ceiling_this_run": round(rate / ceil, 4),
                             "frac_of_guide_copy_ceiling": round(rate / 6.29e12, 4)}
        return {"tensor_bytes": 4 * n, "shape": [rows, L], "ceilings": ceilings, "kernels": kernels}

    big = one_size(256 * 1024, 1024)
    small = one_size(64 * 1024, 1024)
    return {"workload": "HBM-bound kernels of the path on 256Ki x 1024 f32 tensors (1 GiB each)", "peak_GBps": HBM_PEAK / 1e9,
            "guide_copy_ceiling_GBps": 6290.0, "copy_GBps": big["ceilings"]["copy (1 read + 1 write)"]["GBps"],
            "copy_frac_of_peak": big["ceilings"]["copy (1 read + 1 write)"]["frac_of_peak"],
            "stream_ceiling": "softmax forward against the copy; every other kernel against the 2-read + 1-write add",
            **big,
            "cache_assisted_256MB": dict(small, note="256 MB per tensor: partly served by the 256 MB Infinity Cache - NOT HBM rates, not judged")}


def measure_mha(dist, tdev, cdev, steps, warmup):
    """C5: composed multi-head attention d=1024 h=16 S=1024 B=32, dropout 0.1, fwd+bwd."""
    import neuronika_amd
    t = neuronika_amd.tape
    B, S, d, H = 32, 1024, 1024, 16
    t.manual_seed(7 + dist.rank)                     # Philox key of the dropout node: per-rank, so shards draw different masks
    mha = t.nn.MultiheadAttention(tdev, d, H, 0.1, 1)
    if "NK_MHA_STRIDED" in os.environ:               # A/B aid: heads addressed in place (1) or split/merge copies (0)
        mha.strided_heads = os.environ["NK_MHA_STRIDED"] == "1"
    if "NK_MHA_CORE" in os.environ:                  # A/B aid: fused attention kernels (1) or GEMM -> row kernel -> GEMM (0)
        mha.fused_core = os.environ["NK_MHA_CORE"] == "1"
    if os.environ.get("NK_BENCH_UNPACKED_QKV") == "1":   # A/B aid: three projection GEMMs each way instead of the packed one
        mha.packed_qkv = False
    X = t.from_ndarray(tdev, np.random.default_rng(0).random((B * S, d), dtype=np.float32)).requires_grad()
    G = t.from_ndarray(tdev, np.random.default_rng(5).random((B * S, d), dtype=np.float32))
    y = mha.forward(X, B)
    leaves = [X] + [getattr(getattr(mha, n), w) for n in "qkvo" for w in ("weight", "bias")]

    def step():
        y.forward()
        y.no_grad(); y.with_grad()
        y.backward_from(G)
        for p in leaves:
            p.zero_grad()

    dt, ev_ms, gemm, _ = timed_steps(dist, tdev, cdev, step, steps, warmup)
    rec = {"workload": "C5: MHA d_model=1024 heads=16 seq=1024 batch=32 dropout=0.1, composed from reference ops",
           "value": round(B * steps * dist.world / dt, 2), "unit": "sequences/s", "steps": steps,
           "ms_per_step": round(dt / steps * 1e3, 4),
           "step_tflops": round(1.237e12 * steps / dt / 1e12, 2),   # 412.3 GFLOP forward, twice that backward
           "roofline": roofline_mfma(gemm, "sgemm_kernel (projections and their gradients, dK / dV)", "mha_gemm"),
           "gemm_share_of_step": round(gemm[1] / ev_ms, 4) if ev_ms > 0 else None}
    att = EXTRA_STATS.get("attention")
    if att and att[0]:   # nk_attention_fwd / nk_attention_bwd: 4*B*H*S*S*dh flop each (two MFMA products per score tile)
        rec["attention_core"] = dict(roofline_mfma(att, "attention_kernel (scores -> softmax -> dropout -> context, and its backward)", "attention"),
                                     share_of_step=round(att[1] / ev_ms, 4))
    return rec


# ---- what RCCL chose, from rank 0's NCCL_DEBUG=INFO log ---------------------------------------------------------------
_RCCL_ALGO = {0: "Tree", 1: "Ring", 2: "CollNetDirect", 3: "CollNetChain", 4: "NVLS", 5: "NVLSTree", 6: "PAT"}
_RCCL_PROTO = {0: "LL", 1: "LL128", 2: "Simple"}


def parse_rccl_log(text, max_excerpt=16):
    """Version, channel count and the algorithm / protocol per message size out of an NCCL_DEBUG=INFO log (subsystems
    INIT + TUNING).  Tolerant: every field is optional, unmatched logs give an excerpt only."""
    out = {"version": None, "channels": None, "choices": [], "env_overrides": [], "warnings": 0}
    m = re.search(r"(?:RCCL|NCCL) version\s*:?\s*([0-9][^\s]*)", text)
    if m:
        out["version"] = m.group(1)
    m = re.search(r"(\d+) coll channels", text) or re.search(r"coll channels:\s*(\d+)", text)   # NCCL <= 2.2x / RCCL 2.27 wording
    if m:
        out["channels"] = int(m.group(1))
    else:
        ch = [int(x) for x in re.findall(r"Channel \d+/(\d+)", text)]
        if ch:
            out["channels"] = max(ch)
    m = re.search(r"Init timings.*?total ([0-9.]+)", text)
    if m:
        out["comm_init_s"] = float(m.group(1))
    seen = {}
    # "AllReduce: 33554432 Bytes -> Algo 1 proto 2 time 412.0"  (numbers)  or
    # "AllReduce: 33554432 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..31}"  (names, 2.24+)
    for coll, nbytes, algo, proto, rest in re.findall(r"(\w+): (\d+) Bytes -> Algo (\w+) proto (\w+)([^\n]*)", text):
        algo = _RCCL_ALGO.get(int(algo), algo) if algo.isdigit() else algo.capitalize() if algo.isupper() and len(algo) > 4 else algo
        proto = _RCCL_PROTO.get(int(proto), proto) if proto.isdigit() else proto
        ch = re.search(r"=\{(\d+)\.\.(\d+)\}", rest)
        key = (coll, int(nbytes), str(algo), str(proto), (int(ch.group(2)) - int(ch.group(1)) + 1) if ch else None)
        seen[key] = seen.get(key, 0) + 1
    out["choices"] = [{"coll": c, "bytes": b, "algo": a, "proto": pr, **({"channels_used": nch} if nch else {}), "calls": n}
                      for (c, b, a, pr, nch), n in sorted(seen.items(), key=lambda kv: -kv[0][1])][:12]
    out["env_overrides"] = sorted(set(re.findall(r"((?:NCCL|RCCL)_[A-Z0-9_]+) set by environment to ([^\s]+)", text)))[:16]
    out["env_overrides"] = [f"{k}={v}" for k, v in out["env_overrides"]]
    out["warnings"] = len(re.findall(r" NCCL WARN ", text))
    keep = [ln.strip()[-200:] for ln in text.splitlines()
            if re.search(r"RCCL version|NCCL version|coll channels|Channel 00/|Ring 00|Connected all|nRanks|Init timings|P2P|XGMI|xGMI|threadThresholds|Algo", ln)
            and "NCCL WARN" not in ln]
    out["excerpt"] = keep[:max_excerpt]
    return out


def allreduce_alone(dist, tdev, cdev, comm_raw, sizes, reps=20):
    """Stand-alone sum all-reduce of flat f32 buffers (no compute in flight): time per call (max over ranks), algorithm
    bandwidth = bytes / time and bus bandwidth = 2 (p - 1) / p x that (the per-link figure a ring is bound by)."""
    from neuronika_amd import capi
    import ctypes as C
    world, out = dist.world, []
    for nbytes in sizes:
        n = nbytes // 4
        buf = cdev.zeros((n,))
        h = C.c_void_p(comm_raw)

        def ar():
            capi.check(capi.lib.nk_allreduce_sum_async(h, buf.p, n, None))
            capi.check(capi.lib.nk_comm_join(h))
        for _ in range(3):
            ar()
        tdev.sync(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            ar()
        tdev.sync()
        dt = dist.max(time.perf_counter() - t0) / reps
        alg = nbytes / dt / 1e9
        out.append({"bytes": nbytes, "ms": round(dt * 1e3, 4), "algbw_GBps": round(alg, 2),
                    "busbw_GBps": round(alg * 2 * (world - 1) / world, 2)})
        del buf
    return out


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
    # The model in the reference's own words (neuronika-nn/src/lib.rs:441-447 `Linear::forward`, vardiff.rs:282-288 `relu`):
    # `forward(x).relu()` builds ONE Linear+ReLU node per hidden layer (graph-build peephole of the host mirror, ReLU in the
    # GEMM epilogues forward and backward).  NK_BENCH_UNFUSED_RELU=1: the same words with the peephole off - the ReLU node
    # over the Linear's output (same values and gradients bit for bit, tests/test_gpu_tape.py).
    unfused = os.environ.get("NK_BENCH_UNFUSED_RELU") == "1"
    was = t.nn.set_relu_peephole(not unfused)
    try:
        out = lins[2].forward(lins[1].forward(lins[0].forward(X).relu()).relu())
    finally:
        t.nn.set_relu_peephole(was)
    loss = out.mse(T, t.Reduction.Mean)
    params = []
    for lin in lins:
        params += [lin.weight, lin.bias]
    opt = t.optim.SGD(1e-3)
    for p in params:
        opt.register(p)
    _log("graph built")
    comm = sync = None
    replicas = int(os.environ.get("NK_BENCH_REPLICAS", "0")) if world == 1 else 0
    single_rank_rccl = world == 1 and os.environ.get("NK_BENCH_FORCE_RCCL") == "1"
    if world > 1 or single_rank_rccl:
        # (NK_BENCH_FORCE_RCCL=1 on one GPU: a one-rank RCCL communicator with the exchange forced on - every line of
        # the N > 1 path below, including the RCCL log capture, runs on a 1-GPU box; the record is labelled.)
        uid = dist.bcast_bytes(t.dp.Communicator.unique_id() if dist.rank == 0 else None)
        _log("RCCL unique id broadcast, ncclCommInitRank ...")
        comm = t.dp.Communicator(tdev, world, dist.rank, uid)
        _log(f"communicator up: {comm.size} ranks")
        if comm.size != world:
            raise SystemExit(f"[bench] RCCL communicator has {comm.size} ranks, expected {world}")
        sync = t.dp.GradientSync(comm, params)
        if a.dp_channels > 0:
            sync.set_busy_slots(int(os.environ.get("NCCL_MAX_NCHANNELS", a.dp_channels)))
        if single_rank_rccl:
            sync.set_force_exchange(True)
    elif replicas > 1:
        # debugging aid for 1-GPU boxes: the whole N > 1 code path of this function (hook, piecewise hand-over, grouped
        # small gradients, join, the exposed-communication loop) over a replica communicator - `replicas` virtual ranks
        # holding this rank's values, no fabric traffic.  The line is labelled; it is not a multi-GPU measurement.
        comm = t.dp.Communicator.replicas(tdev, replicas)
        sync = t.dp.GradientSync(comm, params)
    real_rccl = comm is not None and replicas <= 1
    seed = 1.0 / (replicas if replicas > 1 else world)
    exchange = [True]

    def step():
        loss.forward()
        loss.no_grad(); loss.with_grad()          # drop + re-create (zero) the intermediate gradients
        if sync is not None and exchange[0]:
            loss.backward_sync(seed, sync)
            sync.join()
        else:
            loss.backward(seed)
        if not a.no_optimizer:
            opt.step()
        opt.zero_grad()

    alone = None
    if real_rccl:   # before the timed loop: what the fabric gives an all-reduce with nothing else running
        total = sync.bytes_per_step()
        sizes = sorted({total, 4 * H * H, 2 * H * H}, reverse=True)   # everything / one weight gradient / the half the step hands over
        alone = allreduce_alone(dist, tdev, cdev, comm.raw(), sizes)
        _log(f"stand-alone all-reduce: {alone[0]['ms']} ms for {alone[0]['bytes']} B")

    dt, ev_ms, gemm, _ = timed_steps(dist, tdev, cdev, step, a.steps, a.warmup)
    _log(f"timed loop done: {dt / a.steps * 1e3:.3f} ms/step")
    settle = EXTRA_STATS["settle_steps"]           # (the later timed_steps calls of this function overwrite it)
    loss_val = loss.item()
    n_exch = sync.exchanges_issued() if sync is not None else 0
    per_rank = dist.gather(ev_ms / a.steps)
    exposed, gemm_off = 0.0, None
    if sync is not None:                           # the same loop with the exchange switched off: what the all-reduce costs
        exchange[0] = False
        dt_off, _, gemm_off, _ = timed_steps(dist, tdev, cdev, step, a.steps, 2)
        exposed = (dt - dt_off) / a.steps * 1e3
        _log(f"exchange off: {dt_off / a.steps * 1e3:.3f} ms/step")
    subs = {}
    if world == 1 and not os.environ.get("NK_BENCH_NO_SUBRECORDS"):
        # every other BASELINE configuration, same record shape, SUB_STEPS timed steps each (C2 at 4096: the second
        # half of BASELINE.json's metric, "MatMul MFMA %peak")
        for n in (1024, 2048, 4096, 8192):
            subs[f"matmul_{n}"] = matmul_record(dist, cdev, n, SUB_STEPS, SUB_WARMUP)
        subs["conv_c3"] = measure_conv(dist, tdev, cdev, SUB_STEPS, SUB_WARMUP)
        subs["mha_c5"] = measure_mha(dist, tdev, cdev, SUB_STEPS, SUB_WARMUP)
        subs["hbm_kernels"] = measure_hbm_kernels(cdev)
        if not a.no_cpu_baseline:
            # the oracle on this host beside every configuration, bounded samples (a few seconds each; the stand-alone
            # workloads `--workload matmul | conv | mha` time the longer ones of BASELINE.md section 4)
            subs["matmul_4096"]["cpu_baseline"] = cpu_baseline_matmul(4096, min_s=2.0)
            subs["conv_c3"]["cpu_baseline"] = cpu_baseline_conv(min_s=2.0)
            subs["mha_c5"]["cpu_baseline"] = cpu_baseline_mha(1, 1024, 1024, 16, 0.1, min_s=1.5)
    elif os.environ.get("NK_BENCH_NO_SUBRECORDS") != "2":   # ("2": the C4 step alone - kernel traces of exactly this workload)
        subs["matmul_4096"] = matmul_record(dist, cdev, 4096, SUB_STEPS, SUB_WARMUP)
    res = None
    if dist.rank == 0:
        debug_run = replicas > 1 or single_rank_rccl
        res = {
            "metric": "training-step samples/sec (fwd+bwd+allreduce)" if not debug_run else
                      "DEBUG RUN of the N > 1 code path on one GPU (no fabric traffic): not a multi-GPU measurement",
            "value": round(B * world * a.steps / dt, 2),
            "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C4: 3-layer MLP Linear({H},{H})x3 + ReLU, MSE mean, batch {B}/GPU, data-parallel "
                                   f"gradient all-reduce (RCCL, side stream)", "global_batch": B * world,
                       "parallelism": f"dp{world}", "optimizer_step_in_timed_region": not a.no_optimizer},
            "rccl_ranks": comm.size if real_rccl else 1,
            **({"replica_ranks_debug": replicas} if replicas > 1 else {}),
            **({"rccl_single_rank_debug": True} if single_rank_rccl else {}),
            "allreduce_bytes_per_step": sync.bytes_per_step() if sync is not None else 0,
            "allreduce_launches_per_step": n_exch // (a.steps + a.warmup + settle) if sync is not None else 0,
            "clock_settle_steps": settle,
            # (a replica communicator moves nothing over the fabric: its "exposed communication" is not one)
            "exposed_comm_ms": None if replicas > 1 else round(exposed, 4),
            "roofline": roofline_mfma(gemm, "sgemm_kernel (f32 MFMA 32x32x2, 128x128x32 tiles)", "sgemm_kernel"),
            **subs,
            "device_ms_per_step": round(ev_ms / a.steps, 4), "loss": loss_val,
            "gemm_share_of_step": round(gemm[1] / ev_ms, 4) if ev_ms > 0 else None,
        }
        if sync is not None:
            res["dp_channels"] = {"nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"), "gemm_busy_slots": sync.busy_slots()}
            res["per_rank_device_ms_per_step"] = [round(v, 4) for v in per_rank]
            res["per_rank_device_ms_min_max"] = [round(min(per_rank), 4), round(max(per_rank), 4)]
            # the same GEMM work per step with the exchange running beside it and without (sums of the HIP-event durations
            # of the step's GEMM launches: with the exchange on, the weight-gradient GEMMs are issued as two row blocks, so
            # launch counts differ and per-launch averages are not comparable)
            g_on, g_off = gemm[1] / a.steps, gemm_off[1] / a.steps
            res["gemm_contention"] = {"gemm_ms_per_step_overlapped": round(g_on, 4), "gemm_launches_per_step_overlapped": gemm[0] // a.steps,
                                      "gemm_ms_per_step_no_exchange": round(g_off, 4), "gemm_launches_per_step_no_exchange": gemm_off[0] // a.steps,
                                      "slowdown": round(g_on / max(1e-9, g_off), 4)}
            if replicas > 1:
                res["replica_step_overhead_ms"] = round(exposed, 4)   # step with the stand-in exchange minus step without: a projection input, not a measurement
        if alone is not None:
            ring_ms = 2 * (world - 1) / world * alone[0]["bytes"] / (XGMI_LINK_GBS * 1e9) * 1e3 if world > 1 else 0.0
            res["allreduce_alone"] = {"ms": alone[0]["ms"], "bytes": alone[0]["bytes"], "algbw_GBps": alone[0]["algbw_GBps"],
                                      "busbw_GBps": alone[0]["busbw_GBps"], "by_size": alone,
                                      "one_link_ring_floor_ms": round(ring_ms, 4)}
            res["rccl"] = read_rccl_log()
        if world == 1 and not debug_run and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_mlp(H, B)
    if comm is not None:
        tdev.sync()
    return res


def read_rccl_log():
    path = os.environ.get("NK_BENCH_RCCL_LOG")
    if not path:
        return {"note": "NCCL_DEBUG was set by the caller: RCCL's log went where the caller sent it"}
    try:
        with open(path, errors="replace") as f:
            text = f.read()
    except OSError as e:
        return {"note": f"no RCCL log at {path}: {e}"}
    return parse_rccl_log(text)


def _headline(res, rec, metric, config_extra=None):
    """A sub-record (measure_*) as a top-level line of its own workload."""
    out = {"metric": metric, "value": rec["value"], "unit": rec["unit"], "n_gpus": res["n_gpus"], "steps": rec["steps"],
           "warmup": res["warmup"], "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": rec["workload"], **(config_extra or {})}}
    out.update({k: v for k, v in rec.items() if k not in ("value", "unit", "steps", "ms_per_step", "workload")})
    return out


def run_matmul(a, dist):
    """C2: C = A.B forward + dA += G.B^T, dB += A^T.G, f32 N x N."""
    from neuronika_amd import capi as c
    dev = c.Device(dist.local)
    rec = matmul_record(dist, dev, a.n, a.steps, a.warmup)
    if dist.rank != 0:
        return None
    rec["value"] = round(rec["value"] * dist.world, 2)
    res = _headline({"n_gpus": dist.world, "warmup": a.warmup}, rec, "MatMul fwd+bwd TFLOP/s (MFMA %peak)", {"n": a.n})
    for k in ("tflops", "frac_of_mfma_peak", "kernel_tflops", "kernel_frac"):
        res.pop(k, None)
    if dist.world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_matmul(a.n)
    return res


def _tape_devices(dist):
    import neuronika_amd
    from neuronika_amd import capi
    tdev = neuronika_amd.tape.Device(dist.local)
    return tdev, capi.Device(handle=tdev.raw())


def run_conv(a, dist):
    tdev, cdev = _tape_devices(dist)
    rec = measure_conv(dist, tdev, cdev, a.steps, a.warmup)
    if dist.rank != 0:
        return None
    res = _headline({"n_gpus": dist.world, "warmup": a.warmup}, rec, "Conv2d fwd+bwd samples/s")
    if dist.world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_conv(128)
    return res


def run_mha(a, dist):
    tdev, cdev = _tape_devices(dist)
    rec = measure_mha(dist, tdev, cdev, a.steps, a.warmup)
    if dist.rank != 0:
        return None
    res = _headline({"n_gpus": dist.world, "warmup": a.warmup}, rec, "MultiheadAttention fwd+bwd sequences/s")
    if dist.world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_mha(2, 1024, 1024, 16, 0.1)
    return res


# ------------------------------------------------------------------------------------------------
# launcher
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tail(path, nbytes=2048):
    try:
        with open(path, "rb") as f:
            f.seek(0, 2)
            f.seek(max(0, f.tell() - nbytes))
            return f.read().decode(errors="replace")
    except OSError:
        return ""


def spawn_ranks(n, cmd=None, have=None, timeout_s=None):
    """`--gpus N` with no launcher: become the launcher.  One worker process per GPU (LOCAL_RANK = device index),
    TCP rendezvous on 127.0.0.1; rank 0 prints the JSON line on our stdout.  Returns the exit code.
    Every rank's stderr goes to a file of its own; when the job fails or exceeds `timeout_s` (NK_BENCH_TIMEOUT_S + 60
    by default: the ranks' own watchdogs fire first and leave a stack dump), the last 2 KB of EVERY rank's stderr are
    relayed, so a hung `ncclCommInitRank` or a crashed rank is visible in the caller's log.
    (`cmd` / `have`: the worker command line and the GPU count, overridable so the launcher itself can be tested
    on a box without GPUs.)"""
    if have is None:
        from neuronika_amd import capi
        have = capi.device_count()
    if cmd is None:
        cmd = [sys.executable, os.path.abspath(__file__), *sys.argv[1:]]
    if have < n:
        print(f"[bench] --gpus {n} requested but this node has {have} GPU(s): refusing to run fewer ranks than asked", file=sys.stderr)
        return 2
    if timeout_s is None:
        timeout_s = float(os.environ.get("NK_BENCH_TIMEOUT_S", "600")) + 60.0
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               NK_RV_SECRET=secrets.token_hex(16), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    logdir = tempfile.mkdtemp(prefix="nk_bench_")
    procs, logs = [], []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        logs.append(os.path.join(logdir, f"rank{r}.stderr"))
        procs.append(subprocess.Popen(cmd, env=e, stderr=open(logs[-1], "wb")))
    rc, why = 0, ""
    t0 = time.time()
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    why = why or f"rank {r} exited with code {code}"
                    rc = rc or (code if code > 0 else 1)
            if rc and pending:          # one rank failed: the others would wait for it forever
                break
            if pending and time.time() - t0 > timeout_s:
                why, rc = f"no result after {timeout_s:.0f} s (ranks still running: {sorted(pending)})", 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    if rc:
        print(f"[bench] {why}; last 2 KB of every rank's stderr follow", file=sys.stderr)
        for r in range(n):
            print(f"[bench] ---- rank {r} (exit code {procs[r].returncode}) ----\n{_tail(logs[r])}", file=sys.stderr)
    else:
        sys.stderr.write(_tail(logs[0]))   # rank 0's progress markers / warnings
    sys.stderr.flush()
    return rc


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("[bench] --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1 or os.environ.get("NK_BENCH_FORCE_RCCL") == "1":
        # a rank that hangs (rendezvous, ncclCommInitRank, a collective nobody else entered) dumps every thread's Python
        # stack on stderr and exits instead of waiting for the driver's kill with an empty tail
        faulthandler.dump_traceback_later(float(os.environ.get("NK_BENCH_TIMEOUT_S", "600")), exit=True)
        if a.dp_channels > 0:
            # RCCL's channel workgroups share the CUs with the backward GEMMs: a known, small number of them (the step's
            # 201 MB of gradients need tens of GB/s, not the fabric's peak) that the GEMM launcher plans around
            os.environ.setdefault("NCCL_MAX_NCHANNELS", str(a.dp_channels))
        if rank == 0 and "NCCL_DEBUG" not in os.environ:
            # what RCCL chose (channels, algorithm / protocol per size) goes into the record: rank 0 logs INIT + TUNING
            # to a file of its own, parsed after the run (read_rccl_log)
            fd, path = tempfile.mkstemp(prefix="nk_rccl_rank0_", suffix=".log")
            os.close(fd)
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,TUNING,ENV", NCCL_DEBUG_FILE=path, NK_BENCH_RCCL_LOG=path)
    _log(f"start: world {world}, workload {a.workload}")
    dist = Dist(a.gpus)
    _log("rendezvous complete")
    try:
        res = {"mlp": run_mlp, "matmul": run_matmul, "conv": run_conv, "mha": run_mha}[a.workload](a, dist)
        if dist.rank == 0 and res is not None:
            res.setdefault("clock_settle_steps", EXTRA_STATS.get("settle_steps", 0))   # untimed steps beyond --warmup (see timed_steps)
            print(json.dumps(res), flush=True)
    finally:
        dist.close()
        faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
