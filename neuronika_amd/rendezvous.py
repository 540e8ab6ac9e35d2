"""Control plane for one-process-per-GPU launches (rendezvous, barrier, small broadcasts and
max-reductions) over plain TCP, using the RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
environment that `python -m torch.distributed.run` provides.

Why not torch.distributed here: the PyTorch wheel bundles its OWN libamdhip64 / libhsa-runtime64
/ librccl; importing torch into the process that also loads libneuronika_hip.so (linked against
/opt/rocm) puts two HIP+HSA runtimes in one address space, which corrupts the heap at exit
(observed: "double free or corruption" and a hung rocprofv3).  The data path (gradient
all-reduce) is RCCL inside the HIP library; only a few bytes of control traffic flow here.
"""
from __future__ import annotations

import os
import pickle
import socket
import struct
import time


def _send(sock, obj):
    b = pickle.dumps(obj)
    sock.sendall(struct.pack("!I", len(b)) + b)


def _recv(sock):
    hdr = b""
    while len(hdr) < 4:
        chunk = sock.recv(4 - len(hdr))
        if not chunk:
            raise ConnectionError("peer closed")
        hdr += chunk
    n = struct.unpack("!I", hdr)[0]
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed")
        buf += chunk
    return pickle.loads(buf)


class Rendezvous:
    """Star topology: rank 0 serves, every collective is gather-to-0 + scatter."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.local = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.peers = []
        self.sock = None
        if self.world == 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        if addr == "localhost":
            addr = "127.0.0.1"
        base = int(port or os.environ.get("MASTER_PORT", "29512"))
        # torchrun's agent already owns MASTER_PORT (its TCPStore); use the next free port of a
        # short, deterministic range and authenticate with a token so that every rank finds the
        # same server.
        ports = [base + 1 + i for i in range(32)]
        token = ("NKRV:" + os.environ.get("TORCHELASTIC_RUN_ID", "static") + f":{self.world}").encode()
        if self.rank == 0:
            srv = None
            for p in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, p))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise OSError(f"no free rendezvous port in {ports[0]}..{ports[-1]}")
            srv.listen(self.world + 8)
            srv.settimeout(timeout)
            peers = {}
            while len(peers) < self.world - 1:
                c, _ = srv.accept()
                c.settimeout(10.0)
                try:
                    if _recv(c) != token:
                        c.close()
                        continue
                    _send(c, token)
                    r = _recv(c)
                except (OSError, ConnectionError, pickle.UnpicklingError):
                    c.close()
                    continue
                c.settimeout(None)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                peers[r] = c
            self.peers = [peers[r] for r in range(1, self.world)]
            srv.close()
        else:
            deadline = time.time() + timeout
            s = None
            while s is None:
                for p in ports:
                    try:
                        c = socket.create_connection((addr, p), timeout=2.0)
                        c.settimeout(5.0)
                        _send(c, token)
                        if _recv(c) == token:
                            s = c
                            break
                        c.close()
                    except (OSError, ConnectionError, pickle.UnpicklingError, struct.error, EOFError):
                        continue
                if s is None:
                    if time.time() > deadline:
                        raise TimeoutError("rendezvous: could not reach rank 0")
                    time.sleep(0.1)
            s.settimeout(None)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send(s, self.rank)
            self.sock = s

    def _collective(self, value, reduce_fn):
        if self.world == 1:
            return reduce_fn([value])
        if self.rank == 0:
            vals = [value] + [_recv(p) for p in self.peers]
            out = reduce_fn(vals)
            for p in self.peers:
                _send(p, out)
            return out
        _send(self.sock, value)
        return _recv(self.sock)

    def barrier(self):
        self._collective(None, lambda v: None)

    def broadcast(self, obj):
        """Value of rank 0 on every rank."""
        return self._collective(obj if self.rank == 0 else None, lambda v: v[0])

    def max(self, x: float) -> float:
        return self._collective(float(x), max)

    def sum(self, x: float) -> float:
        return self._collective(float(x), sum)

    def close(self):
        for p in self.peers:
            p.close()
        if self.sock:
            self.sock.close()
        self.peers, self.sock = [], None
