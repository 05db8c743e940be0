"""Launcher glue for one-process-per-GPU jobs WITHOUT torch.distributed: a star over plain TCP sockets (rank 0 listens, the
others connect) that offers what bench.py needs around the library's own RCCL communicator -- broadcast of a few bytes (the
communicator id, the codebooks), a barrier, max / min over ranks of one number, an all-gather of byte strings, and a
"is everybody still alive and well" check that is run BEFORE any rank enters a blocking collective (cvtmi_comm_create has
no time-out: a rank that failed earlier would leave the others inside ncclCommInitRank for ever).

Addressing follows the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT as set by torch.distributed.run);
MASTER_PORT itself belongs to the launcher's own store, so rank 0 listens on the first free port of MASTER_PORT + 1009 + i
and answers a hello that carries the job's token -- a foreign listener on one of those ports is skipped.
Every wait has a time-out (default 600 s): a dead peer turns into an exception, not a hang.

Nothing received is ever unpickled: the hello is a fixed 48-byte record (magic, SHA-256 of the job token, rank) compared with
hmac.compare_digest before anything else is read from that connection (5 s to deliver it), and the payloads are a closed set of
plain values (None, bool, int, float, str, bytes, list, tuple) in a tagged binary form with a length cap -- a stranger who can
reach the port can at worst make rank 0 drop his connection."""
import hashlib
import hmac
import os
import socket
import struct
import time

_PORT_SHIFT, _PORT_TRIES = 1009, 16
_MAGIC = b"cvtmi-rv"
_MAX_MSG = 1 << 30       # bytes: codebooks are the largest thing sent (128 KB)
_HELLO_TIMEOUT = 5.0


class RendezvousError(RuntimeError):
    pass


def _enc(obj, out):
    if obj is None:
        out.append(b"N")
    elif obj is True:
        out.append(b"T")
    elif obj is False:
        out.append(b"F")
    elif isinstance(obj, int):
        out.append(b"i" + struct.pack("<q", obj))
    elif isinstance(obj, float):
        out.append(b"d" + struct.pack("<d", obj))
    elif isinstance(obj, str):
        b = obj.encode("utf-8")
        out.append(b"s" + struct.pack("<Q", len(b)) + b)
    elif isinstance(obj, (bytes, bytearray, memoryview)):
        b = bytes(obj)
        out.append(b"b" + struct.pack("<Q", len(b)) + b)
    elif isinstance(obj, (list, tuple)):
        out.append((b"l" if isinstance(obj, list) else b"t") + struct.pack("<Q", len(obj)))
        for x in obj:
            _enc(x, out)
    else:
        import numbers
        if isinstance(obj, numbers.Integral):
            out.append(b"i" + struct.pack("<q", int(obj)))
        elif isinstance(obj, numbers.Real):
            out.append(b"d" + struct.pack("<d", float(obj)))
        else:
            raise TypeError("rendezvous: cannot send a %s" % type(obj).__name__)


def dumps(obj):
    out = []
    _enc(obj, out)
    return b"".join(out)


def _dec(buf, p, depth=0):
    if depth > 32 or p >= len(buf):
        raise RendezvousError("malformed message")
    t = buf[p:p + 1]; p += 1
    if t == b"N":
        return None, p
    if t == b"T":
        return True, p
    if t == b"F":
        return False, p
    if t == b"i":
        return struct.unpack_from("<q", buf, p)[0], p + 8
    if t == b"d":
        return struct.unpack_from("<d", buf, p)[0], p + 8
    if t in (b"s", b"b"):
        (n,) = struct.unpack_from("<Q", buf, p); p += 8
        if n > len(buf) - p:
            raise RendezvousError("malformed message")
        raw = bytes(buf[p:p + n])
        return (raw.decode("utf-8") if t == b"s" else raw), p + n
    if t in (b"l", b"t"):
        (n,) = struct.unpack_from("<Q", buf, p); p += 8
        if n > len(buf) - p:      # every element takes at least one byte
            raise RendezvousError("malformed message")
        items = []
        for _ in range(n):
            x, p = _dec(buf, p, depth + 1)
            items.append(x)
        return (items if t == b"l" else tuple(items)), p
    raise RendezvousError("malformed message")


def loads(buf):
    try:
        obj, p = _dec(buf, 0)
    except struct.error:
        raise RendezvousError("malformed message")
    if p != len(buf):
        raise RendezvousError("malformed message")
    return obj


def _send(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(min(1 << 20, n - len(buf)))
        if not part:
            raise RendezvousError("peer closed the connection")
        buf += part
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > _MAX_MSG:
        raise RendezvousError("peer announced a %d-byte message" % n)
    return _recv_exact(sock, n)


class Rendezvous:
    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=600.0, token=None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        self.port = int(os.environ.get("MASTER_PORT", "29500")) if port is None else port
        self.timeout = timeout
        self.token = (token or "cvtmi:%s:%d:%d" % (os.environ.get("TORCHELASTIC_RUN_ID", "-"), self.port, self.world)).encode()
        self._digest = hashlib.sha256(self.token).digest()
        self.peers = {}    # rank 0: rank -> socket
        self.up = None     # other ranks: socket to rank 0
        if self.world > 1:
            self._connect()

    # ---- set-up ----
    def _connect(self):
        deadline = time.time() + self.timeout
        if self.rank == 0:
            srv = None
            for i in range(_PORT_TRIES):
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((self.addr, self.port + _PORT_SHIFT + i))
                    break
                except OSError:
                    srv.close(); srv = None
            if srv is None:
                raise RendezvousError("no free port near %d" % (self.port + _PORT_SHIFT))
            srv.listen(self.world)
            while len(self.peers) < self.world - 1:
                srv.settimeout(max(0.1, deadline - time.time()))
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    raise RendezvousError("only %d of %d ranks showed up" % (len(self.peers) + 1, self.world))
                c.settimeout(_HELLO_TIMEOUT)   # a connection that sends nothing costs five seconds, not the job's time-out
                try:
                    hello = _recv_exact(c, len(_MAGIC) + 32 + 8)
                    peer = struct.unpack("<q", hello[-8:])[0]
                    ok = (hello[:len(_MAGIC)] == _MAGIC and hmac.compare_digest(hello[len(_MAGIC):-8], self._digest)
                          and 0 < peer < self.world and peer not in self.peers)
                except Exception:
                    ok = False
                if not ok:
                    c.close(); continue
                c.settimeout(self.timeout)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                _send(c, b"ok")
                self.peers[peer] = c
            srv.close()
        else:
            while True:
                for i in range(_PORT_TRIES):
                    try:
                        s = socket.create_connection((self.addr, self.port + _PORT_SHIFT + i), timeout=2.0)
                        s.settimeout(self.timeout)
                        s.sendall(_MAGIC + self._digest + struct.pack("<q", self.rank))
                        if _recv(s) == b"ok":
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self.up = s
                            return
                        s.close()
                    except (OSError, RendezvousError, struct.error):
                        pass
                if time.time() > deadline:
                    raise RendezvousError("rank %d: rank 0 did not answer on %s:%d+" % (self.rank, self.addr, self.port + _PORT_SHIFT))
                time.sleep(0.2)

    # ---- primitives (rank 0 is the hub) ----
    def gather(self, obj):
        """rank 0 gets [obj of rank 0, ..., obj of rank world-1]; the others get None"""
        if self.world == 1:
            return [obj]
        try:
            if self.rank == 0:
                out = [obj] + [None] * (self.world - 1)
                for r, c in self.peers.items():
                    out[r] = loads(_recv(c))
                return out
            _send(self.up, dumps(obj))
            return None
        except (OSError, socket.timeout) as e:
            raise RendezvousError("rendezvous gather failed: %s" % e)

    def bcast(self, obj=None):
        """every rank gets rank 0's obj"""
        if self.world == 1:
            return obj
        try:
            if self.rank == 0:
                blob = dumps(obj)
                for c in self.peers.values():
                    _send(c, blob)
                return obj
            return loads(_recv(self.up))
        except (OSError, socket.timeout) as e:
            raise RendezvousError("rendezvous broadcast failed: %s" % e)

    def allgather(self, obj):
        return self.bcast(self.gather(obj))

    def barrier(self):
        self.allgather(None)

    def max(self, x):
        return max(self.allgather(x))

    def min(self, x):
        return min(self.allgather(x))

    def all_ok(self, ok, what=""):
        """True on every rank iff every rank passed ok=True; the ranks that are fine learn who is not"""
        flags = self.allgather((bool(ok), what if not ok else ""))
        bad = [(r, w) for r, (f, w) in enumerate(flags) if not f]
        return len(bad) == 0, bad

    def allgather_bytes(self, mine):
        """concatenation of every rank's byte string in rank order (all the same length)"""
        return b"".join(self.allgather(bytes(mine)))

    def close(self):
        for c in self.peers.values():
            try:
                c.close()
            except OSError:
                pass
        if self.up is not None:
            try:
                self.up.close()
            except OSError:
                pass
        self.peers, self.up = {}, None
