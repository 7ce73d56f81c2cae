"""Host-side rendezvous of the ranks of ONE node: no torch, no MPI -- a directory and a localhost socket.

What a launcher of the row-sharded path needs outside the device library (include/nsparse_dist.h):
  * hand rank 0's ncclUniqueId to the other ranks (nsparse_dist_init wants it on every rank),
  * agree on "did every rank get that far" before the first collective,
  * in the one-GPU EMULATION of a multi-rank run (tests: ranks share a device, no communicator, which RCCL
    refuses) also the barriers and reductions that the real run takes from the device library
    (nsparse_dist_barrier / nsparse_dist_allreduce_f64).

Rank 0 listens on 127.0.0.1 (an ephemeral port, published together with a random token as `<dir>/port` by an atomic
rename); the others poll for that file and connect, and open with {rank, token}.  The directory belongs to the user and
is closed to everybody else (mode 0700, checked on every rank, never a symlink), so the token is known to this job's
ranks only: a connection without it, with a rank outside 1..world-1 or with a rank that has already joined is dropped.
Messages are length-prefixed JSON (floats, bools, small ints, lists of them; the 128-byte ncclUniqueId travels as hex)
-- nothing received from the socket is ever executed or unpickled.  Every collective is "everybody sends to rank 0,
rank 0 answers everybody": tens of microseconds on localhost, nothing to tune for 8 ranks.  Every wait has a deadline
and fails with a message that names the rank and the step -- a missing rank is an error, never a hang.

The directory: $NSPARSE_RDV when the launcher made one (bench.py spawning its own ranks), else
/tmp/nsparse_rdv_<uid>_<MASTER_PORT>_<parent pid> -- the ranks of `python -m torch.distributed.run` share all three.
"""
import hmac
import json
import os
import socket
import stat
import struct
import tempfile
import time


class RendezvousError(RuntimeError):
    pass


def default_dir():
    d = os.environ.get("NSPARSE_RDV")
    if d:
        return d
    return os.path.join(tempfile.gettempdir(),
                        f"nsparse_rdv_{os.geteuid()}_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")


MAX_MESSAGE = 1 << 20  # bytes; the largest real message is a gathered list of 8 small records


def _private_dir(path, create):
    """The rendezvous directory must be a real directory of THIS user that nobody else can enter (the token in it
    is what authenticates a rank).  Rank 0 creates it (0700); a directory that is already there is accepted only
    when it passes the same check -- a pre-created or symlinked one from another user is an error, not a default."""
    if create:
        try:
            os.mkdir(path, 0o700)
        except FileExistsError:
            pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.geteuid() or (st.st_mode & 0o077):
        raise RendezvousError(f"{path}: not a private directory of uid {os.geteuid()} "
                              f"(mode {stat.S_IMODE(st.st_mode):o}, owner {st.st_uid}, dir {stat.S_ISDIR(st.st_mode)})")


def _enc(o):
    if isinstance(o, (bytes, bytearray)):
        return {"__hex__": bytes(o).hex()}
    if isinstance(o, (list, tuple)):
        return [_enc(x) for x in o]
    if isinstance(o, dict):
        return {str(k): _enc(v) for k, v in o.items()}
    if o is None or isinstance(o, (bool, int, float, str)):
        return o
    raise TypeError(f"rendezvous messages carry numbers, strings, bytes and lists of them, not {type(o).__name__}")


def _dec(o):
    if isinstance(o, dict):
        if set(o) == {"__hex__"}:
            return bytes.fromhex(o["__hex__"])
        return {k: _dec(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_dec(x) for x in o]
    return o


def _send(sock, obj):
    data = json.dumps(_enc(obj), allow_nan=True).encode()
    sock.sendall(struct.pack("<I", len(data)) + data)


def _recv(sock):
    hdr = b""
    while len(hdr) < 4:
        chunk = sock.recv(4 - len(hdr))
        if not chunk:
            raise RendezvousError("peer closed the connection")
        hdr += chunk
    (n,) = struct.unpack("<I", hdr)
    if n > MAX_MESSAGE:
        raise RendezvousError(f"message of {n} bytes refused (limit {MAX_MESSAGE})")
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise RendezvousError("peer closed the connection")
        buf += chunk
    try:
        return _dec(json.loads(bytes(buf).decode()))
    except (ValueError, UnicodeDecodeError) as e:
        raise RendezvousError(f"malformed message: {e!r}") from e


class Rendezvous:
    def __init__(self, rank, world, directory=None, timeout=120.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.dir = directory or default_dir()
        self.peers = {}   # rank 0: rank -> socket
        self.sock = None  # other ranks: socket to rank 0
        self.step = 0
        self.rejected = 0  # rank 0: connections dropped at the door (wrong token / rank)
        if self.world == 1:
            return
        deadline = time.time() + self.timeout
        port_file = os.path.join(self.dir, "port")
        if self.rank == 0:
            _private_dir(self.dir, create=True)
            token = os.urandom(16).hex()
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(self.world)
            tmp = port_file + f".tmp{os.getpid()}"
            with open(os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600), "w") as f:
                f.write(f"{srv.getsockname()[1]} {token}")
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
                conn.settimeout(min(self.timeout, 5.0))  # a stranger that connects and says nothing costs 5 s, not the job
                try:
                    hello = _recv(conn)
                    r = hello.get("rank") if isinstance(hello, dict) else None
                    ok = (isinstance(r, int) and not isinstance(r, bool) and 1 <= r < self.world and r not in self.peers
                          and isinstance(hello.get("token"), str) and hmac.compare_digest(hello["token"], token))
                except (RendezvousError, OSError):
                    ok = False
                if not ok:
                    self.rejected += 1
                    conn.close()
                    continue
                conn.settimeout(self.timeout)
                self.peers[r] = conn
        else:
            while not os.path.exists(port_file):
                if time.time() > deadline:
                    raise RendezvousError(f"rank {self.rank}: rank 0 did not publish {port_file} within "
                                          f"{self.timeout:.0f} s")
                time.sleep(0.01)
            _private_dir(self.dir, create=False)
            port, token = open(port_file).read().split()
            port = int(port)
            self.sock = socket.create_connection(("127.0.0.1", port), timeout=self.timeout)
            self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self.sock.settimeout(self.timeout)
            _send(self.sock, {"rank": self.rank, "token": token})

    # ---- the one primitive: everybody's object to rank 0, f(list) back to everybody --------------------------
    def _exchange(self, obj, combine, what):
        self.step += 1
        if self.world == 1:
            return combine([obj])
        try:
            if self.rank == 0:
                got = {0: obj}
                for r, s in self.peers.items():
                    msg = _recv(s)
                    if not (isinstance(msg, list) and len(msg) == 2):
                        raise RendezvousError(f"rank {r}: malformed message in '{what}'")
                    step, val = msg
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
