"""nsparse_amd/rendezvous.py: the torch-free host-side rendezvous of bench.py's ranks (world 2 and 3 as threads of
this process -- the sockets do not care), its deadlines, and the static guarantees the bench makes about itself."""
import os
import re
import threading
import time

import pytest

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
