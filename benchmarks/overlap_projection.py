"""PROJECTION (not a multi-GPU measurement) of what the overlapped gradient exchange costs the C4 step on one GPU.

RCCL cannot run two ranks on one GPU, but what the exchange takes away from the GEMMs can be measured: the replica
communicator's stand-in all-reduce (nk_comm.hip: replica_sum_paced_kernel) occupies K workgroups - RCCL runs one workgroup per
channel - and paces its pass over each gradient to an emulated algorithm bandwidth, launched from the SAME GradientSync
schedule (side stream, events from the last writer, two row blocks per weight gradient, biases as one group).

    python benchmarks/overlap_projection.py [steps]      -> one JSON line per (K, GB/s) + the baseline without exchange

ms/step vs K shows the CU contention; vs GB/s how much of the exchange stays hidden behind backward.  The projected 8-GPU
efficiency is  t(no exchange) / t(K, GB/s)  under the stated assumptions."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuronika_amd  # noqa: E402

t = neuronika_amd.tape


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    H = B = 4096
    dev = t.Device(0)
    x = np.random.default_rng(100).random((B, H), dtype=np.float32)
    tg = np.random.default_rng(200).random((B, H), dtype=np.float32)
    lins = [t.nn.Linear(dev, H, H, s) for s in (1, 3, 5)]
    loss = lins[2].forward(lins[1].forward(lins[0].forward(t.from_ndarray(dev, x)).relu()).relu()).mse(t.from_ndarray(dev, tg), t.Reduction.Mean)
    params = [p for l in lins for p in (l.weight, l.bias)]
    opt = t.optim.SGD(1e-3)
    for p in params:
        opt.register(p)
    ranks = 8

    def run(sync):
        def step():
            loss.forward(); loss.no_grad(); loss.with_grad()
            if sync is not None:
                loss.backward_sync(1.0 / ranks, sync); sync.join()
            else:
                loss.backward(1.0 / ranks)
            opt.step(); opt.zero_grad()
        t_end = time.perf_counter() + 0.25
        while time.perf_counter() < t_end:       # settle the clocks
            step()
        dev.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        dev.sync()
        return (time.perf_counter() - t0) / steps * 1e3

    base = run(None)
    print(json.dumps({"variant": "no exchange", "ms_per_step": round(base, 4)}), flush=True)
    if len(sys.argv) > 2 and sys.argv[2] == "defence":
        # round 5: the same step with the backward GEMMs told how many slots the exchange holds (GradientSync.set_busy_slots ->
        # nk_device_set_busy_slots -> sgemm_tail_kernel) against the plain launches, same box, alternating
        for gbps in (60.0, 120.0):
            for k in (8, 16, 32):
                comm = t.dp.Communicator.replicas(dev, ranks, int(k), float(gbps))
                sync = t.dp.GradientSync(comm, params)
                row = {"variant": "paced replica exchange", "channels": k, "algbw_GBps": gbps}
                for rep in range(2):
                    for busy in (0, k):
                        sync.set_busy_slots(busy)
                        row[f"ms_per_step_busy_slots_{'k' if busy else '0'}_run{rep}"] = round(run(sync), 4)
                row["projected_efficiency_plain"] = round(base / min(row["ms_per_step_busy_slots_0_run0"], row["ms_per_step_busy_slots_0_run1"]), 4)
                row["projected_efficiency_defended"] = round(base / min(row["ms_per_step_busy_slots_k_run0"], row["ms_per_step_busy_slots_k_run1"]), 4)
                print(json.dumps(row), flush=True)
                del sync, comm
        print(json.dumps({"variant": "no exchange (again)", "ms_per_step": round(run(None), 4)}), flush=True)
        return
    for gbps in (60.0, 120.0, 240.0):            # emulated algorithm bandwidth of the all-reduce (bytes of the buffer / time)
        for k in (8, 16, 32, 64, 128):
            comm = t.dp.Communicator.replicas(dev, ranks, int(k), float(gbps))
            sync = t.dp.GradientSync(comm, params)
            ms = run(sync)
            print(json.dumps({"variant": "paced replica exchange", "channels": k, "algbw_GBps": gbps,
                              "exchange_alone_ms": round(sync.bytes_per_step() / gbps / 1e6, 3), "ms_per_step": round(ms, 4),
                              "projected_efficiency": round(base / ms, 4)}), flush=True)
            del sync, comm
    base2 = run(None)
    print(json.dumps({"variant": "no exchange (again)", "ms_per_step": round(base2, 4)}), flush=True)


if __name__ == "__main__":
    main()
