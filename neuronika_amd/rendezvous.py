"""Control plane for one-process-per-GPU launches (rendezvous, barrier, small broadcasts and
max-reductions) over plain TCP, using the RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
environment that `python -m torch.distributed.run` (or bench.py's own spawner) provides.

Why not torch.distributed here: the PyTorch wheel bundles its OWN libamdhip64 / libhsa-runtime64
/ librccl; importing torch into the process that also loads libneuronika_hip.so (linked against
/opt/rocm) puts two HIP+HSA runtimes in one address space, which corrupts the heap at exit
(observed: "double free or corruption" and a hung rocprofv3).  The data path (gradient
all-reduce) is RCCL inside the HIP library; only a few bytes of control traffic flow here.

Wire format (nothing received from the network is ever unpickled or evaluated):

    frame   = u32 length (big endian, <= MAX_FRAME) | u8 tag | payload
    tag 'N' = None (empty payload)      tag 'F' = one IEEE double (8 bytes, big endian)
    tag 'I' = one signed 64-bit int     tag 'B' = raw bytes

Authentication is a mutual HMAC-SHA256 challenge on fixed-length raw bytes BEFORE any frame is
parsed: the server sends a 32-byte nonce, the client answers HMAC(secret, nonce | "client"), the
server verifies with `hmac.compare_digest` and answers HMAC(secret, nonce | "server"), which the
client verifies.  The secret is NK_RV_SECRET (bench.py's spawner draws it from os.urandom) or,
under torchrun ON LOOPBACK ONLY, derived from TORCHELASTIC_RUN_ID and the world size; a routable
MASTER_ADDR without NK_RV_SECRET is refused.
"""
from __future__ import annotations

import hashlib
import ipaddress
import hmac
import os
import socket
import struct
import time

MAX_FRAME = 64 * 1024
_NONCE = 32
_MAC = 32


def _recv_exact(sock, n):
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed")
        buf += chunk
    return buf


def _encode(obj) -> bytes:
    if obj is None:
        return b"N"
    if isinstance(obj, bool):
        raise TypeError("bool is not a control-plane value")
    if isinstance(obj, int):
        return b"I" + struct.pack("!q", obj)
    if isinstance(obj, float):
        return b"F" + struct.pack("!d", obj)
    if isinstance(obj, (bytes, bytearray)):
        return b"B" + bytes(obj)
    raise TypeError(f"control plane carries None / int / float / bytes only, not {type(obj).__name__}")


def _decode(body: bytes):
    if not body:
        raise ConnectionError("empty frame")
    tag, payload = body[:1], body[1:]
    if tag == b"N" and not payload:
        return None
    if tag == b"I" and len(payload) == 8:
        return struct.unpack("!q", payload)[0]
    if tag == b"F" and len(payload) == 8:
        return struct.unpack("!d", payload)[0]
    if tag == b"B":
        return payload
    raise ConnectionError("malformed frame")


def _send(sock, obj):
    b = _encode(obj)
    if len(b) > MAX_FRAME:
        raise ValueError(f"control-plane message of {len(b)} bytes exceeds {MAX_FRAME}")
    sock.sendall(struct.pack("!I", len(b)) + b)


def _recv(sock):
    n = struct.unpack("!I", _recv_exact(sock, 4))[0]
    if n == 0 or n > MAX_FRAME:
        raise ConnectionError(f"frame length {n} out of range")
    return _decode(_recv_exact(sock, n))


def _is_loopback(addr: str) -> bool:
    """The literal name `localhost` or an IP LITERAL inside 127.0.0.0/8 / ::1.  A host name is never loopback here (it
    may resolve anywhere: "127.example.com"): anything that is not an IP literal needs NK_RV_SECRET."""
    if addr == "localhost":
        return True
    try:
        return ipaddress.ip_address(addr).is_loopback
    except ValueError:
        return False


def _secret(world: int, addr: str = "127.0.0.1") -> bytes:
    """NK_RV_SECRET when given.  The fallback (run id + world size) is guessable by anyone who knows the run id, so it
    is accepted only when the rendezvous address is loopback (single-node launches: torchrun --master-addr 127.0.0.1,
    bench.py's own spawner); a routable MASTER_ADDR without NK_RV_SECRET fails closed."""
    s = os.environ.get("NK_RV_SECRET")
    if s:
        return hashlib.sha256(s.encode()).digest()
    if not _is_loopback(addr):
        raise PermissionError(f"rendezvous on the routable address {addr} needs NK_RV_SECRET (a shared random string) in every "
                              "rank's environment: the derived fallback secret is only accepted on loopback")
    return hashlib.sha256(("NKRV:" + os.environ.get("TORCHELASTIC_RUN_ID", "static") + f":{world}").encode()).digest()


def _mac(secret: bytes, nonce: bytes, role: bytes) -> bytes:
    return hmac.new(secret, nonce + role, hashlib.sha256).digest()


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
        # short, deterministic range; the HMAC challenge makes every rank find the same server.
        ports = [base + 1 + i for i in range(32)]
        secret = _secret(self.world, addr)
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
                    nonce = os.urandom(_NONCE)
                    c.sendall(nonce)
                    if not hmac.compare_digest(_recv_exact(c, _MAC), _mac(secret, nonce, b"client")):
                        c.close()
                        continue
                    c.sendall(_mac(secret, nonce, b"server"))
                    r = _recv(c)
                    if not isinstance(r, int) or not 1 <= r < self.world or r in peers:
                        c.close()
                        continue
                except (OSError, ConnectionError, struct.error):
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
                    c = None
                    try:
                        c = socket.create_connection((addr, p), timeout=2.0)
                        c.settimeout(5.0)
                        nonce = _recv_exact(c, _NONCE)
                        c.sendall(_mac(secret, nonce, b"client"))
                        if hmac.compare_digest(_recv_exact(c, _MAC), _mac(secret, nonce, b"server")):
                            s = c
                            break
                        c.close()
                    except (OSError, ConnectionError, struct.error):
                        if c is not None:
                            c.close()
                        continue
                if s is None:
                    if time.time() > deadline:
                        raise TimeoutError("rendezvous: could not reach rank 0")
                    time.sleep(0.1)
            s.settimeout(None)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send(s, int(self.rank))
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
        """Value of rank 0 (None / int / float / bytes up to 64 KiB) on every rank."""
        return self._collective(obj if self.rank == 0 else None, lambda v: v[0])

    def max(self, x: float) -> float:
        return self._collective(float(x), max)

    def sum(self, x: float) -> float:
        return self._collective(float(x), sum)

    def gather(self, x: float):
        """Every rank's float, in rank order, on every rank (at most MAX_FRAME / 8 ranks)."""
        b = self._collective(float(x), lambda v: struct.pack(f"!{len(v)}d", *v))
        return list(struct.unpack(f"!{len(b) // 8}d", b))

    def close(self):
        for p in self.peers:
            p.close()
        if self.sock:
            self.sock.close()
        self.peers, self.sock = [], None
