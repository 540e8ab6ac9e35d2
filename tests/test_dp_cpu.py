"""CPU coverage of the N > 1 path (no GPU needed): two processes over gloo / plain TCP.

* the data-parallel algebra the GPU path relies on — per-rank `backward(seed = 1/p)` on a
  batch shard followed by a SUM all-reduce of the parameter gradients equals the full-batch
  gradient — with the CPU oracle as the compute and gloo as the transport;
* the TCP control plane used by bench.py (rendezvous, barrier, broadcast, max)."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _dp_worker(rank, world, port, q):
    import torch
    import torch.distributed as td
    from oracle import neuronika_oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)                       # identical parameters on every rank
    n, d = 32, 16
    x, t = rng.random((n * world, d), dtype=np.float32), rng.random((n * world, d), dtype=np.float32)
    params = [((rng.random((d, d), dtype=np.float32) - .5), (rng.random(d, dtype=np.float32) - .5)) for _ in range(3)]
    sl = slice(rank * n, (rank + 1) * n)                 # this rank's batch shard
    _, grads = O.mlp_step(x[sl], t[sl], params, seed=1.0 / world)
    flat = [g for pair in grads for g in pair]
    for g in flat:                                       # the exchange step: SUM all-reduce
        tt = torch.from_numpy(g)
        td.all_reduce(tt, op=td.ReduceOp.SUM)
    _, full = O.mlp_step(x, t, params, seed=1.0)         # full-batch reference
    err = max(float(np.abs(a - b).max()) for a, b in zip(flat, [g for pair in full for g in pair]))
    q.put((rank, err))
    td.destroy_process_group()


def test_dp_shard_seed_allreduce_equals_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(30) for p in ps]
    assert [r for r, _ in res] == [0, 1]
    assert all(err < 1e-6 for _, err in res), res


def _rv_worker(rank, world, port, q):
    from neuronika_amd.rendezvous import Rendezvous
    rv = Rendezvous(rank, world, "127.0.0.1", port)
    rv.barrier()
    out = (rank, rv.broadcast(b"\x01" * 128 if rank == 0 else None), rv.max(10.0 + rank), rv.sum(1.0))
    rv.barrier()
    rv.close()
    q.put(out)


@pytest.mark.parametrize("world", [2, 3])
def test_tcp_rendezvous(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rv_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=60) for _ in range(world))
    [p.join(30) for p in ps]
    for r, (rank, uid, mx, sm) in enumerate(res):
        assert rank == r and uid == b"\x01" * 128 and mx == 10.0 + world - 1 and sm == float(world)


def test_bench_is_torch_free():
    """bench.py must not import torch in-process (second HIP runtime; see rendezvous.py)."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    assert "import torch" not in src
