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
import select
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
    return _parse(bytes(buf))


def _parse(data):
    try:
        return _dec(json.loads(data.decode()))
    except (ValueError, UnicodeDecodeError, RecursionError) as e:  # (RecursionError: a deeply nested payload)
        raise RendezvousError(f"malformed message: {type(e).__name__}") from e


MAX_HELLO = 4096      # bytes: {"rank": r, "token": 32 hex digits}
HELLO_TIMEOUT = 5.0   # a connection that has not said hello by then is closed; it delays nobody meanwhile


def _hello_ok(hello, world, joined, token):
    """True when `hello` is {rank: an int in 1..world-1 that has not joined, token: this job's}.  Tokens are compared
    as BYTES (hmac.compare_digest refuses non-ASCII str), and nothing a stranger can put into the message may raise."""
    try:
        r = hello.get("rank") if isinstance(hello, dict) else None
        tok = hello.get("token") if isinstance(hello, dict) else None
        return (isinstance(r, int) and not isinstance(r, bool) and 1 <= r < world and r not in joined
                and isinstance(tok, str)
                and hmac.compare_digest(tok.encode("utf-8", "surrogatepass"), token.encode()))
    except Exception:  # noqa: BLE001 -- whatever it was, it was not a rank of this job
        return False


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
            # Hellos are read concurrently: every accepted connection waits in `pending` with its own deadline and
            # buffer, so a stranger that connects and says nothing (or dribbles bytes) holds up no real rank.
            pending = {}  # socket -> [bytes so far, drop-dead time]
            try:
                while len(self.peers) < self.world - 1:
                    now = time.time()
                    if now >= deadline:
                        missing = sorted(set(range(1, self.world)) - set(self.peers))
                        raise RendezvousError(f"rank 0: ranks {missing} of {self.world} did not join within "
                                              f"{self.timeout:.0f} s ({self.dir})")
                    for c in [c for c, (_, t) in pending.items() if t <= now]:
                        self._reject(pending, c)
                    wake = min([deadline] + [t for _, t in pending.values()])
                    ready, _, _ = select.select([srv] + list(pending), [], [], max(0.0, min(wake - now, 1.0)))
                    for c in ready:
                        if c is srv:
                            try:
                                conn, _ = srv.accept()
                            except OSError:
                                continue
                            if len(pending) >= 4 * self.world + 16:  # a flood: the oldest waiting stranger goes
                                self._reject(pending, min(pending, key=lambda k: pending[k][1]))
                            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            conn.setblocking(False)
                            pending[conn] = [b"", time.time() + min(self.timeout, HELLO_TIMEOUT)]
                            continue
                        # exactly the bytes of the hello and no more: what a rank sends next (its first collective)
                        # stays in the socket for _exchange
                        buf = pending[c][0]
                        want = 4 - len(buf) if len(buf) < 4 else 4 + struct.unpack("<I", buf[:4])[0] - len(buf)
                        try:
                            chunk = c.recv(want)
                        except (BlockingIOError, InterruptedError):
                            continue
                        except OSError:
                            chunk = b""
                        if not chunk:
                            self._reject(pending, c)
                            continue
                        buf += chunk
                        pending[c][0] = buf
                        if len(buf) < 4:
                            continue
                        n = struct.unpack("<I", buf[:4])[0]
                        if n > MAX_HELLO:
                            self._reject(pending, c)  # an absurd length is refused before anything is buffered
                        elif len(buf) == 4 + n:
                            try:
                                hello = _parse(buf[4:])
                            except Exception:  # noqa: BLE001
                                hello = None
                            if _hello_ok(hello, self.world, self.peers, token):
                                del pending[c]
                                c.setblocking(True)
                                c.settimeout(self.timeout)
                                self.peers[hello["rank"]] = c
                            else:
                                self._reject(pending, c)
            finally:
                for c in list(pending):
                    c.close()
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

    def _reject(self, pending, conn):
        self.rejected += 1
        pending.pop(conn, None)
        conn.close()

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
