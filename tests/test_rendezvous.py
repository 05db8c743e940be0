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
