"""nsparse_amd/rendezvous.py: the torch-free host-side rendezvous of bench.py's ranks (world 2 and 3 as threads of
this process -- the sockets do not care), its deadlines, and the static guarantees the bench makes about itself."""
import os
import re
import threading
import time

import pytest

from nsparse_amd import rendezvous
from nsparse_amd.rendezvous import Rendezvous, RendezvousError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(world, body, tmp_path, timeout=20.0):
    out, errs = [None] * world, []

    def one(r):
        try:
            rdv = Rendezvous(r, world, directory=str(tmp_path / "rdv"), timeout=timeout)
            out[r] = body(r, rdv)
            rdv.close()
        except Exception as e:  # surfaced below
            errs.append((r, e))
    th = [threading.Thread(target=one, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout + 10)
    assert not errs, errs
    return out


@pytest.mark.parametrize("world", [1, 2, 3])
def test_collectives(world, tmp_path):
    def body(r, rdv):
        ident = rdv.bcast(b"\x01" * 128 if r == 0 else None, "id")
        s = rdv.allreduce([r + 1.0, 10.0 * r], "sum")
        m = rdv.allreduce([r + 1.0], "max")
        rdv.barrier()
        ok_all = rdv.all_ok(True)
        ok_some = rdv.all_ok(r != world - 1 or world == 1)
        return ident, s, m, ok_all, ok_some, rdv.gather(r * r)
    for r, (ident, s, m, ok_all, ok_some, g) in enumerate(_run_ranks(world, body, tmp_path)):
        assert ident == b"\x01" * 128
        assert s == [sum(q + 1.0 for q in range(world)), sum(10.0 * q for q in range(world))]
        assert m == [float(world)]
        assert ok_all is True and ok_some is (world == 1)
        assert g == [q * q for q in range(world)]


def test_a_missing_rank_is_an_error_not_a_hang(tmp_path):
    t0 = time.time()
    with pytest.raises(RendezvousError, match="did not join"):
        Rendezvous(0, 2, directory=str(tmp_path / "a"), timeout=1.0)
    with pytest.raises(RendezvousError, match="did not publish"):
        Rendezvous(1, 2, directory=str(tmp_path / "b"), timeout=1.0)
    assert time.time() - t0 < 10.0


def test_a_rank_that_stops_answering_times_out(tmp_path):
    errs = []

    def rank1():
        rdv = Rendezvous(1, 2, directory=str(tmp_path / "rdv"), timeout=2.0)
        rdv.barrier()
        time.sleep(4.0)  # never enters the second barrier in time
        rdv.close()
    t = threading.Thread(target=rank1)
    t.start()
    rdv = Rendezvous(0, 2, directory=str(tmp_path / "rdv"), timeout=2.0)
    rdv.barrier()
    try:
        rdv.barrier("second barrier")
    except RendezvousError as e:
        errs.append(str(e))
    t.join()
    rdv.close()
    assert errs and "second barrier" in errs[0]


def test_strangers_are_dropped_at_the_door(tmp_path):
    """Advisor r04: the socket is on localhost and anyone can connect.  A connection without the job's token, with a
    rank outside 1..world-1, with a rank that has already joined, or with bytes that are not a message must be
    dropped without disturbing the rendezvous; nothing received is unpickled (a pickle is just malformed JSON)."""
    import json
    import pickle
    import socket
    import struct
    d = str(tmp_path / "rdv")
    res = {}

    def rank0():
        rdv = Rendezvous(0, 2, directory=d, timeout=20.0)
        res["sum"] = rdv.allreduce([1.0], "sum")
        res["rejected"] = rdv.rejected
        rdv.close()
    t = threading.Thread(target=rank0)
    t.start()
    port_file = os.path.join(d, "port")
    for _ in range(2000):
        if os.path.exists(port_file):
            break
        time.sleep(0.005)
    port, token = open(port_file).read().split()
    assert os.stat(d).st_mode & 0o077 == 0 and os.stat(port_file).st_mode & 0o077 == 0

    def knock(payload):
        c = socket.create_connection(("127.0.0.1", int(port)), timeout=5)
        c.sendall(struct.pack("<I", len(payload)) + payload)
        try:
            assert c.recv(1) == b""  # closed by rank 0, nothing sent back
        except (ConnectionResetError, socket.timeout):
            pass
        c.close()

    class Boom:
        def __reduce__(self):
            return (os.system, ("touch %s" % (tmp_path / "pwned"),))
    knock(pickle.dumps(Boom()))
    knock(json.dumps({"rank": 1, "token": "0" * 32}).encode())
    knock(json.dumps({"rank": 7, "token": token}).encode())
    knock(json.dumps({"rank": 0, "token": token}).encode())
    knock(json.dumps({"rank": True, "token": token}).encode())
    knock(json.dumps([1, token]).encode())
    # advisor r05: a token that is not ASCII (hmac.compare_digest raises TypeError on such a str) and a payload nested
    # deeper than the JSON parser recurses (RecursionError) used to take rank 0 down with one packet each
    knock(json.dumps({"rank": 1, "token": "\u00e9" * 32}).encode())
    knock(json.dumps({"rank": 1, "token": "\ud800"}).encode())
    deep = b'{"rank":1,"token":' + b"[" * 1900 + b"]" * 1900 + b"}"  # fits a hello (4096 B), too deep for json.loads
    with pytest.raises(RendezvousError, match="RecursionError"):
        rendezvous._parse(deep)
    knock(deep)
    c = socket.create_connection(("127.0.0.1", int(port)), timeout=5)
    c.sendall(struct.pack("<I", 1 << 30))  # an absurd length is refused before anything is buffered
    c.close()
    # silent strangers: connected, saying nothing.  They used to be served one at a time, 5 s each, with the real rank
    # queued behind them; now they wait in their own slots and the real rank joins at once
    mute = [socket.create_connection(("127.0.0.1", int(port)), timeout=5) for _ in range(6)]
    t0 = time.time()
    rdv1 = Rendezvous(1, 2, directory=d, timeout=20.0)
    assert rdv1.allreduce([2.0], "sum") == [3.0]
    assert time.time() - t0 < 3.0, "the real rank waited behind connections that said nothing"
    rdv1.close()
    t.join(30)
    for c in mute:
        c.close()
    assert res["sum"] == [3.0] and res["rejected"] == 10  # (the mute ones were still waiting when the job was complete)
    assert not (tmp_path / "pwned").exists()


def test_a_duplicate_rank_is_dropped(tmp_path):
    """Two processes claiming rank 1: the first joins, the second is dropped and times out instead of replacing it."""
    d = str(tmp_path / "rdv")
    got = {}

    def rank0():
        try:
            Rendezvous(0, 3, directory=d, timeout=2.0)
        except RendezvousError as e:
            got["e"] = str(e)
    t = threading.Thread(target=rank0)
    t.start()
    a = Rendezvous(1, 3, directory=d, timeout=5.0)
    b = Rendezvous(1, 3, directory=d, timeout=5.0)  # connects; rank 0 closes it
    t.join(10)
    assert "ranks [2]" in got["e"]
    a.close()
    b.close()


def test_a_directory_of_somebody_else_is_refused(tmp_path):
    """A rendezvous directory that others can enter (pre-created 0777, or a symlink) is an error on every rank."""
    loose = tmp_path / "loose"
    loose.mkdir()
    os.chmod(loose, 0o777)
    with pytest.raises(RendezvousError, match="not a private directory"):
        Rendezvous(0, 2, directory=str(loose), timeout=1.0)
    (loose / "port").write_text("1 x")
    with pytest.raises(RendezvousError, match="not a private directory"):
        Rendezvous(1, 2, directory=str(loose), timeout=1.0)
    real = tmp_path / "real"
    real.mkdir(mode=0o700)
    link = tmp_path / "link"
    link.symlink_to(real)
    with pytest.raises(RendezvousError, match="not a private directory"):
        Rendezvous(0, 2, directory=str(link), timeout=1.0)


def test_only_plain_data_travels():
    from nsparse_amd import rendezvous as R
    src = open(R.__file__).read()
    assert "pickle" not in src.replace("unpickled", "").replace("(a pickle", "")
    assert R._dec(R._enc([b"\x00\xff", 1.5, True, None, "s", [1, 2]])) == [b"\x00\xff", 1.5, True, None, "s", [1, 2]]
    with pytest.raises(TypeError):
        R._enc(object())


def test_bench_is_torch_free():
    src = open(os.path.join(ROOT, "bench.py")).read()
    code = "\n".join(ln for ln in src.splitlines() if not ln.lstrip().startswith("#"))
    code = re.sub(r'""".*?"""', "", code, flags=re.S)
    assert not re.search(r"^\s*(import|from)\s+torch", code, flags=re.M), "bench.py must not import torch"
    for mod in ("tools/bench_config.py", "tools/pmc_one.py", "nsparse_amd/rendezvous.py", "nsparse_amd/capi.py"):
        s = open(os.path.join(ROOT, mod)).read()
        assert not re.search(r"^\s*(import|from)\s+torch", s, flags=re.M), mod


def test_bench_refuses_more_ranks_than_gpus_without_a_gpu():
    """No GPU here: the rank process must say so and leave with a non-zero code at once (nothing waits for a peer)."""
    import subprocess
    import sys
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu", "--no-pmc", "--no-vendor", "--no-configs", "--no-irregular", "--no-large"],
                       capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
    assert r.returncode != 0
    assert "GPU" in r.stderr
    assert time.time() - t0 < 90
