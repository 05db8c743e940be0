"""cvt_amd/rendezvous.py: the torch-free launcher glue of bench.py --gpus N (broadcast, barrier, max / min, all-gather, the
readiness check) between real processes over 127.0.0.1."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["CVT_ROOT"])
from cvt_amd.rendezvous import Rendezvous
rv = Rendezvous(timeout=60)
r, w = rv.rank, rv.world
assert rv.bcast(b"id-%d" % r if r == 0 else None) == b"id-0"
assert rv.allgather(r * r) == [i * i for i in range(w)]
assert rv.max(float(r)) == float(w - 1) and rv.min(r + 5) == 5
assert rv.allgather_bytes(bytes([r]) * 3) == b"".join(bytes([i]) * 3 for i in range(w))
rv.barrier()
ok, bad = rv.all_ok(r != 1, "rank 1 says no")
assert not ok and bad == [(1, "rank 1 says no")], bad
ok, bad = rv.all_ok(True)
assert ok and bad == []
rv.barrier(); rv.close()
print("rank %d fine" % r)
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_rendezvous_three_processes():
    port = _free_port()
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CVT_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_rendezvous_missing_rank_times_out():
    import pytest
    sys.path.insert(0, ROOT)
    from cvt_amd.rendezvous import Rendezvous, RendezvousError
    with pytest.raises(RendezvousError):
        Rendezvous(rank=0, world=2, addr="127.0.0.1", port=_free_port(), timeout=1.0)


def test_codec_round_trip_and_garbage():
    import pytest
    sys.path.insert(0, ROOT)
    from cvt_amd import rendezvous as rz
    for obj in (None, True, False, 0, -7, 1 << 40, 2.5, "text \u00e9", b"\x00\xff" * 9, [1, (2.0, None, [b"x"])], (True, "why"), []):
        assert rz.loads(rz.dumps(obj)) == obj
    import numpy as np
    assert rz.loads(rz.dumps((np.int64(3), np.float32(0.5)))) == (3, 0.5)
    for junk in (b"", b"x", b"i123", b"l" + b"\xff" * 8, b"s" + b"\x09" + b"\x00" * 7 + b"ab", b"N" + b"N", b"\x80\x04\x95"):
        with pytest.raises(rz.RendezvousError):
            rz.loads(junk)
    with pytest.raises(TypeError):
        rz.dumps({"a": 1})


def test_stranger_on_the_port_is_dropped():
    """a connection that sends a pickle (the old hello), nothing at all, or the wrong token never becomes a peer -- and the real
    rank still gets in"""
    import pickle
    import struct
    import threading
    import time
    sys.path.insert(0, ROOT)
    from cvt_amd import rendezvous as rz
    port = _free_port()
    got = {}

    def hub():
        got["rv"] = rz.Rendezvous(rank=0, world=2, addr="127.0.0.1", port=port, timeout=30.0, token="job-1")

    old = rz._HELLO_TIMEOUT
    rz._HELLO_TIMEOUT = 0.5
    try:
        t = threading.Thread(target=hub); t.start()
        time.sleep(0.3)
        lp = port + rz._PORT_SHIFT
        evil = pickle.dumps(("job-1".encode(), 1))
        strangers = []
        for payload in (struct.pack("<Q", len(evil)) + evil, b"", rz._MAGIC + b"\x00" * 32 + struct.pack("<q", 1),
                        struct.pack("<Q", 1 << 62)):
            c = socket.create_connection(("127.0.0.1", lp), timeout=5.0)
            if payload:
                c.sendall(payload)
            strangers.append(c)
        peer = rz.Rendezvous(rank=1, world=2, addr="127.0.0.1", port=port, timeout=30.0, token="job-1")
        t.join(timeout=30)
        assert not t.is_alive() and list(got["rv"].peers) == [1]
    finally:
        rz._HELLO_TIMEOUT = old
        for c in strangers:
            c.close()
    th = threading.Thread(target=lambda: got.__setitem__("x", peer.bcast()))
    th.start()
    got["rv"].bcast((True, b"payload"))
    th.join(timeout=10)
    assert got["x"] == (True, b"payload")
    peer.close(); got["rv"].close()
