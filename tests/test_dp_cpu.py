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
    assert rv.gather(0.5 + rank) == [0.5 + r for r in range(world)]
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


def _hostile(port, n_tries=6):
    """A peer that does not know the secret: garbage instead of the HMAC answer, a huge length prefix, a pickle."""
    import pickle, struct, time
    payloads = [b"\x00" * 32, struct.pack("!I", 0xFFFFFFFF) + b"x" * 28, pickle.dumps(("boom",))[:32].ljust(32, b"."),
                b"", b"A" * 100000]
    sent = 0
    deadline = time.time() + 20
    while sent < n_tries and time.time() < deadline:
        for off in range(1, 33):
            try:
                c = socket.create_connection(("127.0.0.1", port + off), timeout=0.5)
            except OSError:
                continue
            try:
                c.settimeout(2.0)
                c.recv(32)                                   # the nonce
                c.sendall(payloads[sent % len(payloads)])
                c.recv(64)
            except OSError:
                pass
            finally:
                c.close()
            sent += 1
            break
        else:
            time.sleep(0.05)


def test_tcp_rendezvous_ignores_unauthenticated_peers():
    """Connections that fail the HMAC challenge are dropped before anything they send is parsed (nothing on this
    channel is ever unpickled) and do not disturb the rendezvous of the real ranks."""
    import threading
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    th = threading.Thread(target=_hostile, args=(port,), daemon=True)
    ps = [ctx.Process(target=_rv_worker, args=(0, world, port, q))]
    ps[0].start()
    th.start()
    import time; time.sleep(1.0)                            # the hostile peer talks to rank 0 first
    ps.append(ctx.Process(target=_rv_worker, args=(1, world, port, q)))
    ps[1].start()
    res = sorted(q.get(timeout=60) for _ in range(world))
    [p.join(30) for p in ps]
    th.join(25)
    assert [r[0] for r in res] == [0, 1] and all(r[1] == b"\x01" * 128 for r in res)


def test_control_plane_wire_format():
    from neuronika_amd import rendezvous as rz
    for v in (None, 0, -5, 2 ** 40, 1.5, float("inf"), b"", b"\x00\xff" * 64):
        assert rz._decode(rz._encode(v)) == v
    for bad in ("text", [1], {"a": 1}, True, (1, 2)):
        with pytest.raises(TypeError):
            rz._encode(bad)
    for body in (b"", b"Z", b"I123", b"F1", b"Nx"):
        with pytest.raises(ConnectionError):
            rz._decode(body)
    a, b = socket.socketpair()
    with pytest.raises(ValueError):
        rz._send(a, b"x" * (rz.MAX_FRAME + 1))
    import struct
    a.sendall(struct.pack("!I", rz.MAX_FRAME + 1))          # an oversized length prefix is refused before any allocation
    with pytest.raises(ConnectionError):
        rz._recv(b)
    src = open(rz.__file__).read()
    assert "import pickle" not in src and "pickle.loads" not in src
    a.close(); b.close()


def test_bench_refuses_ranks_without_gpus():
    """`python bench.py --gpus 2` with no launcher must spawn the ranks itself or fail loudly; on a box with fewer
    than two GPUs that means a non-zero exit and no JSON line (never a silent single-rank run)."""
    import subprocess, sys
    from neuronika_amd import capi
    if capi.device_count() >= 2:
        pytest.skip("two GPUs present: covered by tests/test_gpu_multi.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode != 0 and r.stdout.strip() == "" and "refusing" in r.stderr
    # a launcher that started a different number of ranks than --gpus asks for is an error as well
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True,
                       text=True, timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE=1" in r.stderr


_SPAWN_WORKER = """
import os, sys
sys.path.insert(0, {root!r})
from neuronika_amd.rendezvous import Rendezvous
rv = Rendezvous()
assert rv.world == int(os.environ["WORLD_SIZE"]) and rv.local == rv.rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
uid = rv.broadcast(b"u" * 128 if rv.rank == 0 else None)
tot = rv.sum(float(rv.rank + 1))
rv.barrier()
if rv.rank == {fail_rank}:
    sys.exit(3)
if rv.rank == 0:
    print("OK", rv.world, tot, len(uid), flush=True)
rv.close()
"""


def test_bench_launcher_spawns_and_relays(tmp_path, capfd):
    """The launcher half of `bench.py --gpus N` (environment, rendezvous, rank 0's line on our stdout, failure of any
    rank = failure of the job), driven with a stand-in worker so it runs without GPUs."""
    import importlib.util, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("nk_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    ok = tmp_path / "ok.py"; ok.write_text(_SPAWN_WORKER.format(root=root, fail_rank=-1))
    assert bench.spawn_ranks(3, cmd=[sys.executable, str(ok)], have=3) == 0
    assert "OK 3 6.0 128" in capfd.readouterr().out
    bad = tmp_path / "bad.py"; bad.write_text(_SPAWN_WORKER.format(root=root, fail_rank=1))
    assert bench.spawn_ranks(2, cmd=[sys.executable, str(bad)], have=2) == 3
    assert bench.spawn_ranks(4, cmd=[sys.executable, str(ok)], have=2) == 2      # fewer GPUs than ranks: refused


_HANG_WORKER = """
import os, sys, time
print("rank", os.environ["RANK"], "alive", file=sys.stderr, flush=True)
if os.environ["RANK"] == "1":
    print("rank 1 is stuck in ncclCommInitRank", file=sys.stderr, flush=True)
    time.sleep(600)
time.sleep(600)
"""


def test_bench_launcher_times_out_and_relays_every_ranks_stderr(tmp_path, capfd):
    """A job that hangs (e.g. in ncclCommInitRank on its first multi-GPU run) ends at the launcher's timeout with exit
    code 124 and the tail of EVERY rank's stderr, not as a silent kill; a rank that dies reports its own tail too."""
    import importlib.util, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("nk_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    hang = tmp_path / "hang.py"; hang.write_text(_HANG_WORKER)
    t0 = time.time()
    assert bench.spawn_ranks(2, cmd=[sys.executable, str(hang)], have=2, timeout_s=2.0) == 124
    assert time.time() - t0 < 30
    err = capfd.readouterr().err
    assert "no result after 2 s" in err and "rank 0 alive" in err and "rank 1 is stuck in ncclCommInitRank" in err
    bad = tmp_path / "bad.py"; bad.write_text(_SPAWN_WORKER.format(root=root, fail_rank=1).replace("sys.exit(3)", "sys.stderr.write('boom on rank 1\\n'); sys.exit(3)"))
    assert bench.spawn_ranks(2, cmd=[sys.executable, str(bad)], have=2) == 3
    err = capfd.readouterr().err
    assert "rank 1 exited with code 3" in err and "boom on rank 1" in err


_RCCL_LOG = """
host:101:101 [0] NCCL INFO RCCL version : 2.26.6-HEAD:1234
host:101:101 [0] NCCL INFO NCCL_MAX_NCHANNELS set by environment to 32.
host:101:120 [0] NCCL INFO Channel 00/32 : 0 1 2 3 4 5 6 7
host:101:120 [0] NCCL INFO Channel 31/32 : 0 7 6 5 4 3 2 1
host:101:120 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1 [1] 1/-1/-1->0->-1
host:101:120 [0] NCCL INFO Connected all rings
host:101:120 [0] NCCL INFO 32 coll channels, 0 collnet channels, 0 nvls channels, 32 p2p channels, 2 p2p channels per peer
host:101:101 [0] NCCL INFO AllReduce: 33554432 Bytes -> Algo 1 proto 2 time 412.0
host:101:101 [0] NCCL INFO AllReduce: 33554432 Bytes -> Algo 1 proto 2 time 412.0
host:101:101 [0] NCCL INFO AllReduce: 49152 Bytes -> Algo 0 proto 0 time 21.5
host:101:101 [0] NCCL WARN something odd
"""


def test_bench_parses_what_rccl_chose():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("nk_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    r = bench.parse_rccl_log(_RCCL_LOG)
    assert r["version"].startswith("2.26.6") and r["channels"] == 32 and r["warnings"] == 1
    assert r["choices"][0] == {"coll": "AllReduce", "bytes": 33554432, "algo": "Ring", "proto": "Simple", "calls": 2}
    assert r["choices"][1]["algo"] == "Tree" and r["choices"][1]["proto"] == "LL"
    assert r["env_overrides"] == ["NCCL_MAX_NCHANNELS=32."] or r["env_overrides"] == ["NCCL_MAX_NCHANNELS=32"]
    assert any("coll channels" in ln for ln in r["excerpt"])
    # RCCL 2.27 wording (gpurun_out of round 3, one rank): "coll channels:128", names instead of numbers in TUNING lines
    r27 = bench.parse_rccl_log("x NCCL INFO RCCL version : 2.27.7-HEAD:0d2c4fd\n"
                               "x NCCL INFO comm:0x1, nRanks:8, nNodes:1, coll channels:128 collnet channels:128, nvls channels:0, p2p channels:64\n"
                               "x NCCL INFO AllReduce: 33554432 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..31}\n"
                               "x NCCL INFO Init timings - ncclCommInitRank_impl: rank 0 nranks 8 total 1.68 (kernels 1.60)\n")
    assert r27["version"].startswith("2.27.7") and r27["channels"] == 128 and r27["comm_init_s"] == 1.68
    assert r27["choices"] == [{"coll": "AllReduce", "bytes": 33554432, "algo": "RING", "proto": "SIMPLE", "channels_used": 32, "calls": 1}]
    empty = bench.parse_rccl_log("nothing useful")
    assert empty["channels"] is None and empty["choices"] == [] and empty["excerpt"] == []


def test_rendezvous_fallback_secret_is_loopback_only(monkeypatch):
    """the derived secret (run id + world) is guessable; a routable MASTER_ADDR must bring NK_RV_SECRET."""
    from neuronika_amd import rendezvous as rz
    monkeypatch.delenv("NK_RV_SECRET", raising=False)
    assert len(rz._secret(2, "127.0.0.1")) == 32
    with pytest.raises(PermissionError):
        rz._secret(2, "10.1.2.3")
    for name in ("127.example.com", "127.0.0.1.evil.org", "localhost.example.com", "2130706433", ""):   # not IP literals in 127/8
        with pytest.raises(PermissionError):
            rz._secret(2, name)
    assert len(rz._secret(2, "127.8.9.10")) == 32 and len(rz._secret(2, "::1")) == 32 and len(rz._secret(2, "localhost")) == 32
    monkeypatch.setenv("NK_RV_SECRET", "s3cret")
    assert len(rz._secret(2, "10.1.2.3")) == 32


def test_rendezvous_under_torch_distributed_run(tmp_path):
    """The driver launches the multi-GPU bench as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py ...`: the control plane must come up from THAT environment (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*, secret derived from the run id on loopback, ports next to the agent's store)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    w = tmp_path / "w.py"
    w.write_text(_SPAWN_WORKER.format(root=root, fail_rank=-1))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NK_RV_SECRET")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(w)], capture_output=True, text=True, timeout=180, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "OK 3 6.0 128" in r.stdout


def test_bench_is_torch_free():
    """bench.py must not import torch in-process (second HIP runtime; see rendezvous.py)."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    assert "import torch" not in src
