"""Multi-GPU execution of the PRODUCT path (one process per GPU, RCCL over xGMI), skipped on a box with fewer than
two GPUs: `bench.py --gpus 2` self-spawned, and the data-parallel algebra of SURVEY.md section 8e with the tape,
`dp::GradientSync` and RCCL doing the work - per-rank `backward(1/p)` on a batch shard + overlapped sum all-reduce
== the full-batch gradient of a single rank (what tests/test_dp_cpu.py checks with the oracle over gloo).

Insertion point in the reference: between `VarDiff::backward` (neuronika-variable/src/vardiff.rs:125-141) and
`Optimizer::step` (neuronika-optim/src/optimizer.rs:81-86)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    from neuronika_amd import capi
    return capi.device_count()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_bench_refuses_more_ranks_than_gpus():
    """`--gpus N` on a node with fewer GPUs must fail, never report a smaller job under the requested label."""
    n = _ngpu() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing" in r.stderr and r.stdout.strip() == ""


@pytest.mark.skipif(_ngpu() < 2, reason="needs two GPUs")
def test_bench_self_spawns_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--hidden", "1024", "--batch", "512", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2
    assert res["config"]["global_batch"] == 1024
    assert res["allreduce_bytes_per_step"] == 3 * (1024 * 1024 + 1024) * 4
    assert np.isfinite(res["loss"]) and res["value"] > 0


_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import neuronika_amd
from neuronika_amd.rendezvous import Rendezvous
t = neuronika_amd.tape
rv = Rendezvous()
rank, world = rv.rank, rv.world
dev = t.Device(rv.local)
H, B = {hidden}, {batch}                       # B rows per rank
rng = np.random.default_rng(7)
x = rng.random((B * world, H), dtype=np.float32)
tg = rng.random((B * world, H), dtype=np.float32)

def build(xs, ts):
    lins = [t.nn.Linear(dev, H, H, s) for s in (1, 3, 5)]          # identical weights on every rank
    out = lins[2].forward(lins[1].forward(lins[0].forward(t.from_ndarray(dev, xs)).relu()).relu())
    loss = out.mse(t.from_ndarray(dev, ts), t.Reduction.Mean)
    return loss, [p for l in lins for p in (l.weight, l.bias)]

uid = rv.broadcast(t.dp.Communicator.unique_id() if rank == 0 else None)
comm = t.dp.Communicator(dev, world, rank, uid)
assert comm.size == world
sl = slice(rank * B, (rank + 1) * B)
loss, params = build(x[sl], tg[sl])
sync = t.dp.GradientSync(comm, params)
worst = 0.0
for rep in range(2):                                  # second pass: re-armed events, recycled buffers
    for p in params: p.zero_grad()
    loss.forward(); loss.no_grad(); loss.with_grad()
    loss.backward_sync(1.0 / world, sync); sync.join()
    got = [p.grad().copy() for p in params]
    full, fparams = build(x, tg)                      # single-rank full-batch reference on this GPU
    full.forward(); full.backward(1.0)
    for g, fp in zip(got, fparams):
        w = fp.grad()
        err = float(np.abs(g - w).max() / (np.abs(w).max() + 1e-30))
        worst = max(worst, err)
assert sync.elements_exchanged() == 2 * sum(int(np.prod(p.shape)) for p in params)
mx = rv.max(worst)
rv.barrier()
dev.sync()
if rank == 0:
    print("WORST", mx, flush=True)
rv.close()
"""


@pytest.mark.skipif(_ngpu() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("hidden,batch", [(4096, 256), (640, 96)])   # piecewise weight gradients / small unsplit ones
def test_two_rank_gradient_sync_equals_full_batch(tmp_path, hidden, batch):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, hidden=hidden, batch=batch))
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), NK_RV_SECRET="t",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    worst = float([l for l in outs[0][0].splitlines() if l.startswith("WORST")][-1].split()[1])
    assert worst < 2e-5, worst       # f32 sums in a different order (two shards of B rows vs one pass over 2B rows)
