"""4096^3 GEMMs (NT / NN / TN, the C4 step's three layouts) next to a background stream of fabric traffic, one GPU.

VERDICT r03 item 7: the GEMMs pull 3 - 4x their algorithmic bytes over the fabric (Infinity-Cache hits, harmless on an idle
chip); an 8-GPU step adds xGMI traffic through the same fabric.  Does that traffic slow the GEMMs?  The load generator is the
paced replica kernel of nk_comm.hip (K workgroups walking a buffer at a set algorithm bandwidth, read + write) on the side
stream, (a) over device memory (HBM) and (b) over page-locked HOST memory - every byte of (b) crosses the IO die and the
fabric the way peer traffic does.  Prints one JSON line per point: background GB/s asked / achieved, GEMM us per launch.

    python benchmarks/gemm_under_load.py            (about 30 s on the GPU box)
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from neuronika_amd import capi as c  # noqa: E402
from benchmarks.microbench import rand  # noqa: E402


class Raw:
    """A float buffer by address (device or page-locked host memory)."""
    def __init__(self, ptr, n):
        self.p, self.size, self.shape = C.c_void_p(ptr), n, (n,)


def main():
    dev = c.Device(0)
    n = 4096
    A, B, Cm = rand(dev, (n, n), 0, 0, 1), rand(dev, (n, n), 1, 0, 1), dev.zeros((n, n))
    layouts = (("NT", 0, 1), ("NN", 0, 0), ("TN", 1, 0))
    big = dev.zeros((1 << 29,))                       # 2 GB of HBM for the background stream
    host_n = 1 << 26                                  # 256 MB page-locked
    hp = C.c_void_p()
    c.check(c.lib.nk_host_alloc(host_n * 4, C.byref(hp)))
    C.memset(hp, 0, host_n * 4)
    host = Raw(hp.value, host_n)

    def gemms(reps):
        e0, e1 = dev.event(), dev.event()
        out = {}
        for name, ta, tb in layouts:
            for _ in range(2):
                c.sgemm(dev, ta, tb, n, n, n, 1.0, A, n, B, n, 0.0, Cm, n)
            e0.record()
            for _ in range(reps):
                c.sgemm(dev, ta, tb, n, n, n, 1.0, A, n, B, n, 0.0, Cm, n)
            e1.record()
            e1.sync()
            out[name] = round(e0.elapsed_ms(e1) / reps * 1e3, 1)
        return out

    for _ in range(2):
        base = gemms(8)                               # clock settle
    print(json.dumps({"background": "none", "gemm_us": base}), flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "defence":
        # round 5: the same launches with the device handle told how many slots the generator holds
        # (nk_device_set_busy_slots -> sgemm_tail_kernel: whole rounds of the free slots + the left-over tiles cut along K)
        # against the plain launches - INTERLEAVED under one background pass (plain, told, plain, told ... per layout, three
        # rounds, after ~12 ms of untimed GEMMs: the first measurement behind an idle gap reads high, r04's NT column) - and what
        # a wrong guess costs on an idle chip.  Median of the three rounds, us per launch.
        import statistics

        def one(ta, tb, reps=3):
            e0, e1 = dev.event(), dev.event()
            e0.record()
            for _ in range(reps):
                c.sgemm(dev, ta, tb, n, n, n, 1.0, A, n, B, n, 0.0, Cm, n)
            e1.record(); e1.sync()
            return e0.elapsed_ms(e1) / reps * 1e3

        def interleaved(busy):
            for _ in range(12):
                c.sgemm(dev, 0, 1, n, n, n, 1.0, A, n, B, n, 0.0, Cm, n)
            got = {(name, b): [] for name, _, _ in layouts for b in (0, busy)}
            for _ in range(3):
                for name, ta, tb in layouts:
                    for b in (0, busy):
                        dev.busy_slots(b)
                        got[(name, b)].append(one(ta, tb))
            dev.busy_slots(0)
            return {f"{name}_{'told' if b else 'plain'}": round(statistics.median(v), 1) for (name, b), v in got.items() if b or True}

        rate = 16.0                                   # GB/s each way: the loss does not follow the rate (r04); 1.9 GB lasts ~120 ms
        for channels in (8, 16, 32, 64):
            idle = interleaved(channels)
            comm = c.Comm(dev, 1, 0, None, channels=channels, gbps=rate)
            count = min(big.size, int(rate * 1e9 * 0.12 / 4))
            bg0, bg1 = dev.event(), dev.event()
            bg0.record(comm_stream=True)
            comm.allreduce_sum_async(Raw(big.p.value, count))
            bg1.record(comm_stream=True)
            e_start, e_end = dev.event(), dev.event()
            e_start.record()
            loaded = interleaved(channels)
            e_end.record(); e_end.sync()
            gemm_window_ms = e_start.elapsed_ms(e_end)
            comm.join(); dev.sync()
            pass_ms = bg0.elapsed_ms(bg1)
            comm.close()
            row = {"channels": channels, "busy_slots_told": channels, "background_pass_ms": round(pass_ms, 1), "gemm_window_ms": round(gemm_window_ms, 1),
                   "covered": pass_ms >= gemm_window_ms, "idle_chip_us": idle, "beside_generator_us": loaded}
            row["loss_plain"] = {k: round(loaded[f"{k}_plain"] / idle[f"{k}_plain"] - 1, 4) for k, _, _ in layouts}
            row["loss_told"] = {k: round(loaded[f"{k}_told"] / idle[f"{k}_plain"] - 1, 4) for k, _, _ in layouts}
            row["idle_cost_of_telling"] = {k: round(idle[f"{k}_told"] / idle[f"{k}_plain"] - 1, 4) for k, _, _ in layouts}
            print(json.dumps(row), flush=True)
        c.check(c.lib.nk_host_free(hp))
        return
    for where, buf, points in (("hbm", big, ((16, 50.0), (16, 100.0), (16, 200.0), (32, 400.0), (64, 800.0))),
                               ("host-pinned", host, ((8, 10.0), (16, 25.0), (32, 50.0)))):
        for channels, gbps in points:
            comm = c.Comm(dev, 1, 0, None, channels=channels, gbps=gbps)     # one virtual rank: x *= 1, read + write of every byte
            # size the pass so that it outlasts the GEMMs it runs beside: ~45 ms at the asked rate
            count = min(buf.size, int(gbps * 1e9 * 0.045 / 4))
            span = Raw(buf.p.value, count)
            bg0, bg1 = dev.event(), dev.event()
            bg0.record(comm_stream=True)
            comm.allreduce_sum_async(span)
            bg1.record(comm_stream=True)
            t = gemms(4)                              # 3 layouts x (2 + 4) launches ~ 18 ms of GEMMs beside the pass
            comm.join()
            dev.sync()
            achieved = count * 4 / (bg0.elapsed_ms(bg1) * 1e-3) / 1e9
            print(json.dumps({"background": where, "channels": channels, "asked_GBps": gbps, "achieved_GBps": round(achieved, 1),
                              "bytes_moved_each_way": count * 4, "gemm_us": t,
                              "slowdown": {k: round(t[k] / base[k], 4) for k in t}}), flush=True)
            comm.close()
    # (c) traffic that occupies NO compute unit: page-locked host -> HBM copies by the SDMA engines on the copy stream, queued
    # back to back underneath the GEMMs (PCIe Gen5 x16: the order of magnitude of one GPU's share of an 8-GPU all-reduce,
    # ~350 MB per 8 ms step in each direction).  What (a) and (b) show is mostly the price of SHARING CUs with the generator's
    # workgroups (it does not follow the rate); this row isolates the fabric.
    dst = dev.zeros((host_n,))
    for ncopies in (0, 6):
        e0, e1 = dev.event(), dev.event()
        dev.sync()
        for _ in range(ncopies):
            c.check(c.lib.nk_upload_async(dev.h, dst.p, hp, host_n))
        t = gemms(4)
        dev.sync()
        print(json.dumps({"background": "sdma host->device copies" if ncopies else "none (before sdma)", "copies_of_256MB": ncopies, "gemm_us": t,
                          "slowdown": {k: round(t[k] / base[k], 4) for k in t}}), flush=True)
    # how long do those copies take alone (rate of the background stream)
    e0, e1 = dev.event(), dev.event()
    dev.sync()
    import time
    t0 = time.perf_counter()
    for _ in range(6):
        c.check(c.lib.nk_upload_async(dev.h, dst.p, hp, host_n))
    dev.sync()
    dt = time.perf_counter() - t0
    print(json.dumps({"background": "sdma copies alone", "GBps": round(6 * host_n * 4 / dt / 1e9, 1), "ms": round(dt * 1e3, 2)}), flush=True)
    base2 = gemms(8)
    print(json.dumps({"background": "none (again)", "gemm_us": base2}), flush=True)
    c.check(c.lib.nk_host_free(hp))


if __name__ == "__main__":
    main()
