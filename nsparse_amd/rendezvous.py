"""Host-side rendezvous of the ranks of ONE node: no torch, no MPI -- a directory and a localhost socket.

What a launcher of the row-sharded path needs outside the device library (include/nsparse_dist.h):
  * hand rank 0's ncclUniqueId to the other ranks (nsparse_dist_init wants it on every rank),
  * agree on "did every rank get that far" before the first collective,
  * in the one-GPU EMULATION of a multi-rank run (tests: ranks share a device, no communicator, which RCCL
    refuses) also the barriers and reductions that the real run takes from the device library
    (nsparse_dist_barrier / nsparse_dist_allreduce_f64).

Rank 0 listens on 127.0.0.1 (an ephemeral port, published as `<dir>/port` by an atomic rename); the others poll for
that file and connect.  Every collective is "everybody sends to rank 0, rank 0 answers everybody": tens of
microseconds on localhost, nothing to tune for 8 ranks.  Every wait has a deadline and fails with a message that
names the rank and the step -- a missing rank is an error, never a hang.

The directory: $NSPARSE_RDV when the launcher made one (bench.py spawning its own ranks), else
/tmp/nsparse_rdv_<MASTER_PORT>_<parent pid> -- the ranks of `python -m torch.distributed.run` share both.
"""
import os
import pickle
import socket
import struct
import tempfile
import time


class RendezvousError(RuntimeError):
    pass


def default_dir():
    d = os.environ.get("NSPARSE_RDV")
    if d:
        return d
    return os.path.join(tempfile.gettempdir(), f"nsparse_rdv_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=4)
    sock.sendall(struct.pack("<I", len(data)) + data)


def _recv(sock):
    hdr = b""
    while len(hdr) < 4:
        chunk = sock.recv(4 - len(hdr))
        if not chunk:
            raise RendezvousError("peer closed the connection")
        hdr += chunk
    (n,) = struct.unpack("<I", hdr)
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise RendezvousError("peer closed the connection")
        buf += chunk
    return pickle.loads(bytes(buf))


class Rendezvous:
    def __init__(self, rank, world, directory=None, timeout=120.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.dir = directory or default_dir()
        self.peers = {}   # rank 0: rank -> socket
        self.sock = None  # other ranks: socket to rank 0
        self.step = 0
        if self.world == 1:
            return
        deadline = time.time() + self.timeout
        port_file = os.path.join(self.dir, "port")
        if self.rank == 0:
            os.makedirs(self.dir, exist_ok=True)
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(self.world)
            tmp = port_file + f".tmp{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(str(srv.getsockname()[1]))
            os.replace(tmp, port_file)  # atomic: a reader sees the whole number or no file
            self._srv = srv
            while len(self.peers) < self.world - 1:
                left = deadline - time.time()
                if left <= 0:
                    missing = sorted(set(range(1, self.world)) - set(self.peers))
                    raise RendezvousError(f"rank 0: ranks {missing} of {self.world} did not join within "
                                          f"{self.timeout:.0f} s ({self.dir})")
                srv.settimeout(left)
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(self.timeout)
                self.peers[int(_recv(conn))] = conn
        else:
            while not os.path.exists(port_file):
                if time.time() > deadline:
                    raise RendezvousError(f"rank {self.rank}: rank 0 did not publish {port_file} within "
                                          f"{self.timeout:.0f} s")
                time.sleep(0.01)
            port = int(open(port_file).read())
            self.sock = socket.create_connection(("127.0.0.1", port), timeout=self.timeout)
            self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self.sock.settimeout(self.timeout)
            _send(self.sock, self.rank)

    # ---- the one primitive: everybody's object to rank 0, f(list) back to everybody --------------------------
    def _exchange(self, obj, combine, what):
        self.step += 1
        if self.world == 1:
            return combine([obj])
        try:
            if self.rank == 0:
                got = {0: obj}
                for r, s in self.peers.items():
                    step, val = _recv(s)
                    if step != self.step:
                        raise RendezvousError(f"rank {r} is at step {step}, rank 0 at {self.step} ({what})")
                    got[r] = val
                res = combine([got[r] for r in range(self.world)])
                for s in self.peers.values():
                    _send(s, res)
                return res
            _send(self.sock, (self.step, obj))
            return _recv(self.sock)
        except (socket.timeout, OSError) as e:
            raise RendezvousError(f"rank {self.rank} of {self.world}: '{what}' (step {self.step}) did not complete "
                                  f"within {self.timeout:.0f} s: {e!r}") from e

    def barrier(self, what="barrier"):
        self._exchange(None, lambda xs: None, what)

    def bcast(self, obj, what="broadcast"):
        """rank 0's object on every rank."""
        return self._exchange(obj if self.rank == 0 else None, lambda xs: xs[0], what)

    def allreduce(self, vals, op="sum", what="allreduce"):
        """element-wise sum / max of equally long float lists."""
        f = sum if op == "sum" else max
        return self._exchange([float(v) for v in vals], lambda xs: [f(col) for col in zip(*xs)], what)

    def gather(self, obj, what="gather"):
        """every rank's object, as a list, on every rank."""
        return self._exchange(obj, lambda xs: list(xs), what)

    def all_ok(self, ok, what="agree"):
        """True when EVERY rank says ok (the ranks then take the same branch)."""
        return bool(self._exchange(bool(ok), lambda xs: all(xs), what))

    def close(self):
        for s in list(self.peers.values()) + ([self.sock] if self.sock else []):
            try:
                s.close()
            except OSError:
                pass
        if self.rank == 0 and self.world > 1:
            try:
                self._srv.close()
                os.remove(os.path.join(self.dir, "port"))
                os.rmdir(self.dir)
            except OSError:
                pass
