"""Document sharding and the sufficient-statistic exchange for multi-GPU runs.

The reference has no distributed code.  Documents are
independent given (beta, mu_d, siginv) -- reference src/modules/stm.py:519-588
has no cross-document dependence except the two accumulators and the bound --
so every rank owns a contiguous, nnz-balanced range of documents and ONE
all-reduce per EM iteration sums

    [ bound | sigma_ss | regression moments incl. eta^T eta | beta_ss ]

after which each rank finishes the (tiny) M-step redundantly: gamma from the
centred moments, and (eta - mu)^T (eta - mu) expanded in the same moments
(strutopy_amd.stm.STM._covariance_from_moments).  A second, (K-1)^2 all-reduce
of the locally computed covariance is only taken when that expansion would
cancel too many digits.

Layers
  * host group  -- rendezvous + small host collectives between the ranks of one
    node.  ``TcpGroup`` is the product's: Python stdlib sockets only (no
    PyTorch, no pickle), a star through rank 0, found through ``STM_RDZV_PORT`` or a
    private rendezvous file keyed by MASTER_ADDR / MASTER_PORT, authenticated by the
    run's shared secret.  (tests/_gloo_rig.py has the same interface over
    torch.distributed/gloo for the CPU test rig.)
  * communicators used by STM
      ``SingleComm``  world size 1
      ``RcclComm``    the product path: RCCL all-reduce on the device-resident
                      packed buffer through the C-ABI (stm_comm_init /
                      stm_allreduce_*); the host group only ships the 128-byte
                      ncclUniqueId, barriers and a few scalars
      ``HostComm``    pull / reduce on the host / push back; for engines without
                      device collectives (CPU tests) and as a collective
                      fallback when RCCL cannot initialise
"""
import hashlib
import hmac
import os
import socket
import stat
import struct
import tempfile
import time

import numpy as np


def shard_bounds(indptr, world):
    """Contiguous document ranges with (nearly) equal nnz per rank: list of (lo, hi)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    N = len(indptr) - 1
    nnz = int(indptr[-1])
    cuts = [0]
    for r in range(1, world):
        target = nnz * r // world
        c = int(np.searchsorted(indptr, target, side="left"))
        c = min(max(c, cuts[-1]), N)
        cuts.append(c)
    cuts.append(N)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


# ------------------------------------------------------------------------------ host groups
# Wire format of the host group.  Nothing received is ever unpickled: a frame is an 8-byte length (capped) and a body in
# the small tagged encoding below (None, bool, int, float, str, bytes, list, tuple, dict, ndarray; object arrays element by element), and a connection is
# only used after both ends have proved, with an HMAC over a fresh nonce, that they hold the run's shared secret
# (STM_RDZV_SECRET from the launcher, or a 0600 file in a 0700 directory owned by this user).
_MAGIC = b"STMRDZV2"
_MAX_FRAME = 1 << 31
_SCALAR_TAGS = (b"N", b"T", b"F", b"I", b"D", b"S", b"B")


def _enc(obj, out):
    if obj is None:
        out.append(b"N")
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b"T" if obj else b"F")
    elif isinstance(obj, (int, np.integer)):
        out.append(b"I" + struct.pack("<q", int(obj)))
    elif isinstance(obj, (float, np.floating)):
        out.append(b"D" + struct.pack("<d", float(obj)))
    elif isinstance(obj, str):
        raw = obj.encode("utf-8")
        out.append(b"S" + struct.pack("<Q", len(raw)) + raw)
    elif isinstance(obj, (bytes, bytearray)):
        out.append(b"B" + struct.pack("<Q", len(obj)) + bytes(obj))
    elif isinstance(obj, (list, tuple)):
        out.append((b"L" if isinstance(obj, list) else b"U") + struct.pack("<Q", len(obj)))
        for x in obj:
            _enc(x, out)
    elif isinstance(obj, dict):
        out.append(b"M" + struct.pack("<Q", len(obj)))
        for k, v in obj.items():
            if not (k is None or isinstance(k, (bool, np.bool_, int, np.integer, float, np.floating, str, bytes))):
                raise TypeError(f"host group: a dict key of type {type(k).__name__} cannot be sent")
            _enc(k, out)    # keys keep their type ({1: 2} does not come back as {'1': 2})
            _enc(v, out)
    elif isinstance(obj, np.ndarray) and obj.dtype.hasobject:
        # what pandas hands over for string / categorical columns: the elements go one by one through this same tagged
        # encoding (scalars only -- None, bool, int, float, str, bytes), never as a serialised Python object
        if obj.dtype != object:
            raise TypeError("host group: structured arrays with object fields are not sent")
        out.append(b"O" + struct.pack("<B", obj.ndim) + struct.pack(f"<{obj.ndim}q", *obj.shape))
        for x in obj.ravel():
            if not (x is None or isinstance(x, (bool, np.bool_, int, np.integer, float, np.floating, str, bytes, bytearray))):
                raise TypeError(f"host group: an object array holding {type(x).__name__} cannot be sent")
            _enc(x, out)
    elif isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        dt = a.dtype.str.encode("ascii")
        out.append(b"A" + struct.pack("<BB", len(dt), a.ndim) + dt + struct.pack(f"<{a.ndim}q", *a.shape))
        out.append(a.tobytes())
    else:
        raise TypeError(f"host group: cannot send {type(obj).__name__}")


def _encode(obj):
    out = []
    _enc(obj, out)
    return b"".join(out)


def sendable(obj):
    """"" when the host group can carry `obj`, else why not (what _enc would raise) -- asked before a collective, so that a rank
    does not find out alone, inside it."""
    try:
        _enc(obj, [])
        return ""
    except (TypeError, ValueError) as e:
        return f"{type(e).__name__}: {e}"


def _dec(buf, pos):
    tag = buf[pos:pos + 1]
    pos += 1
    if tag == b"N":
        return None, pos
    if tag in (b"T", b"F"):
        return tag == b"T", pos
    if tag == b"I":
        return struct.unpack_from("<q", buf, pos)[0], pos + 8
    if tag == b"D":
        return struct.unpack_from("<d", buf, pos)[0], pos + 8
    if tag in (b"S", b"B"):
        (n,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        if n > len(buf) - pos:
            raise ValueError("host group: truncated frame")
        raw = bytes(buf[pos:pos + n])
        return (raw.decode("utf-8") if tag == b"S" else raw), pos + n
    if tag in (b"L", b"U", b"M"):
        (n,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        if n > len(buf) - pos:
            raise ValueError("host group: truncated frame")
        items = []
        for q in range(n if tag != b"M" else 2 * n):
            if tag == b"M" and q % 2 == 0 and buf[pos:pos + 1] not in _SCALAR_TAGS:
                raise ValueError("host group: a dict key must be a scalar")     # (what _enc sends; a list key would not hash)
            x, pos = _dec(buf, pos)
            items.append(x)
        if tag == b"L":
            return items, pos
        if tag == b"U":
            return tuple(items), pos
        return dict(zip(items[0::2], items[1::2])), pos
    if tag == b"O":
        (nd,) = struct.unpack_from("<B", buf, pos)
        pos += 1
        if nd > 8:
            raise ValueError("host group: bad array header")
        shape = struct.unpack_from(f"<{nd}q", buf, pos)
        pos += 8 * nd
        cnt = 1
        for d in shape:
            if d < 0 or d > _MAX_FRAME:
                raise ValueError("host group: bad array shape")
            cnt *= d
        # every element takes at least its tag byte; an empty array may not carry huge sibling dimensions into reshape
        if cnt > len(buf) - pos or (cnt == 0 and any(d > (1 << 20) for d in shape)):
            raise ValueError("host group: truncated frame")
        arr = np.empty(cnt, dtype=object)
        for q in range(cnt):
            if buf[pos:pos + 1] not in _SCALAR_TAGS:
                raise ValueError("host group: an object array holds scalars only")
            arr[q], pos = _dec(buf, pos)
        return arr.reshape(shape), pos
    if tag == b"A":
        ld, nd = struct.unpack_from("<BB", buf, pos)
        pos += 2
        dt = np.dtype(bytes(buf[pos:pos + ld]).decode("ascii"))
        pos += ld
        if dt.hasobject or nd > 8:
            raise ValueError("host group: bad array header")
        shape = struct.unpack_from(f"<{nd}q", buf, pos)
        pos += 8 * nd
        cnt = 1
        for d in shape:
            if d < 0:
                raise ValueError("host group: bad array shape")
            cnt *= d
        nbytes = cnt * dt.itemsize
        if nbytes > len(buf) - pos:
            raise ValueError("host group: truncated frame")
        return np.frombuffer(buf, dtype=dt, count=cnt, offset=pos).reshape(shape).copy(), pos + nbytes
    raise ValueError("host group: unknown tag")


def _decode(buf):
    obj, pos = _dec(buf, 0)
    if pos != len(buf):
        raise ValueError("host group: trailing bytes")
    return obj


def _send_msg(sock, obj):
    data = _encode(obj)
    if len(data) > _MAX_FRAME:
        raise ValueError("host group: frame too large")
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > _MAX_FRAME:
        raise ConnectionError("host group: frame length out of range")
    return _decode(_recv_exact(sock, n))


def _private_dir():
    """A directory only this user can enter (created 0700; refused when it is a link, someone else's or wider open)."""
    d = os.path.join(tempfile.gettempdir(), f"stm_rdzv_{os.getuid()}")
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (stat.S_IMODE(st.st_mode) & 0o077):
        raise PermissionError(f"rendezvous directory {d} is not private to this user")
    return d


def rendezvous_file(addr, port, run_id=""):
    """Where rank 0 publishes its listening port and the run's secret when the launcher exported neither
    STM_RDZV_PORT nor STM_RDZV_SECRET (the ranks share one node and one user)."""
    tag = f"{addr}_{port}_{run_id}".replace("/", "_").replace(":", "_")
    return os.path.join(_private_dir(), tag)


def _publish(path, text):
    tmp = f"{path}.{os.getpid()}.tmp"
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    with os.fdopen(fd, "w") as fh:
        fh.write(text)
    os.replace(tmp, path)      # atomic


def _read_private(path):
    st = os.lstat(path)
    if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or (stat.S_IMODE(st.st_mode) & 0o077):
        raise PermissionError(f"rendezvous file {path} is not private to this user")
    with open(path) as fh:
        return fh.read().split()


def _mac(secret, *parts):
    return hmac.new(secret, b"".join(parts), hashlib.sha256).digest()


class TcpGroup:
    """Host-side group over stdlib TCP sockets: every collective is a gather to rank 0 and a reply.

    Payloads are small (a 128-byte ncclUniqueId, scalars, at most the packed sufficient statistics
    in the HostComm fallback), so a star is enough; the data path between GPUs is RCCL.
    """
    kind = "tcp"

    def __init__(self, rank, size, addr="127.0.0.1", port=None, rdzv_file=None, timeout=600.0, secret=None):
        self.rank, self.size = int(rank), int(size)
        self._peers = []       # rank 0: sockets of ranks 1..size-1 (index r-1)
        self._root = None      # other ranks: socket to rank 0
        self._file = None
        if self.size <= 1:
            return
        if port is None and rdzv_file is None:
            raise ValueError("TcpGroup needs a port or a rendezvous file")
        if secret is None and os.environ.get("STM_RDZV_SECRET"):
            secret = bytes.fromhex(os.environ["STM_RDZV_SECRET"])
        if secret is None and rdzv_file is None:
            rdzv_file = rendezvous_file(addr, port, "secret")     # a fixed port without a secret: the file carries the secret
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, int(port) if port is not None else 0))
            srv.listen(self.size)
            if rdzv_file is not None:
                if secret is None:
                    secret = os.urandom(32)
                _publish(rdzv_file, f"{srv.getsockname()[1]} {secret.hex()}\n")
                self._file = rdzv_file
            peers = {}
            srv.settimeout(1.0)
            while len(peers) < self.size - 1:
                if time.time() > deadline:
                    raise TimeoutError(f"rendezvous: {len(peers) + 1} of {self.size} ranks arrived")
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                try:       # a stray, stale or hostile connection must not take the rendezvous down
                    c.settimeout(5.0)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    hello = _recv_exact(c, 64)
                    magic, (r, sz), nonce, mac = hello[:8], struct.unpack("<II", hello[8:16]), hello[16:32], hello[32:]
                    if (magic != _MAGIC or sz != self.size or not 0 < r < self.size or r in peers
                            or not hmac.compare_digest(mac, _mac(secret, hello[:32]))):
                        c.close()
                        continue
                    c.sendall(_MAGIC + _mac(secret, b"welcome", nonce))
                    c.settimeout(timeout)
                    peers[r] = c
                except (OSError, ConnectionError, struct.error, ValueError):
                    c.close()
            srv.close()
            self._peers = [peers[r] for r in range(1, self.size)]
            if self._file:
                try:
                    os.unlink(self._file)
                except OSError:
                    pass
        else:
            while True:
                if time.time() > deadline:
                    raise TimeoutError("rendezvous: rank 0 did not appear")
                p, sec = port, secret
                if rdzv_file is not None:
                    try:
                        fp, fs = _read_private(rdzv_file)
                        p, sec = (fp if port is None else port), (bytes.fromhex(fs) if secret is None else secret)
                    except (OSError, ValueError):
                        time.sleep(0.05)
                        continue
                try:
                    s = socket.create_connection((addr, int(p)), timeout=5.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    nonce = os.urandom(16)
                    head = _MAGIC + struct.pack("<II", self.rank, self.size) + nonce
                    s.sendall(head + _mac(sec, head))
                    reply = _recv_exact(s, 40)
                    if reply[:8] == _MAGIC and hmac.compare_digest(reply[8:], _mac(sec, b"welcome", nonce)):
                        s.settimeout(timeout)
                        self._root = s
                        break
                    s.close()
                except (OSError, ConnectionError, struct.error):
                    pass
                time.sleep(0.05)       # stale file / rank 0 not listening yet: read again

    # every collective: gather to the root, combine there, reply to everyone
    def allgather(self, obj):
        if self.size <= 1:
            return [obj]
        if self.rank == 0:
            objs = [obj] + [_recv_msg(c) for c in self._peers]
            for c in self._peers:
                _send_msg(c, objs)
            return objs
        _send_msg(self._root, obj)
        return _recv_msg(self._root)

    def _reduce(self, arr, fn):
        if self.size <= 1:
            return arr
        if self.rank == 0:
            acc = arr
            for c in self._peers:      # fixed rank order: every run sums in the same order
                acc = fn(acc, _recv_msg(c))
            for c in self._peers:
                _send_msg(c, acc)
            return acc
        _send_msg(self._root, arr)
        return _recv_msg(self._root)

    def allreduce(self, buf, op="sum"):
        a = np.array(buf, dtype=np.float64, copy=True)
        return self._reduce(a, np.add if op == "sum" else np.maximum)

    def gather(self, obj, dst=0):
        """Every rank's object at rank `dst` (a list in rank order), None elsewhere: the shards of theta / eta / mu on their
        way into save_model's files travel once, to the rank that writes them."""
        if self.size <= 1:
            return [obj]
        if dst != 0:
            objs = self.allgather(obj)
            return objs if self.rank == dst else None
        if self.rank == 0:
            return [obj] + [_recv_msg(c) for c in self._peers]
        _send_msg(self._root, obj)
        return None

    def broadcast(self, obj, src=0):
        return self.allgather(obj if self.rank == src else None)[src]

    def barrier(self):
        self.allgather(None)

    def close(self):
        for c in self._peers:
            c.close()
        if self._root is not None:
            self._root.close()
        self._peers, self._root = [], None


def init_from_env(timeout=600.0):
    """TcpGroup from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (what torch.distributed.run and
    bench.py's own launcher export).  MASTER_PORT itself belongs to the launcher's store, so rank 0
    listens on STM_RDZV_PORT when given, else on an ephemeral port published through a rendezvous file."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world <= 1:
        return TcpGroup(0, 1)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = os.environ.get("STM_RDZV_PORT")
    if port:
        return TcpGroup(rank, world, addr, port=int(port), timeout=timeout)
    run_id = os.environ.get("TORCHELASTIC_RUN_ID", "") + "_" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    return TcpGroup(rank, world, addr, rdzv_file=rendezvous_file(addr, os.environ.get("MASTER_PORT", "29500"), run_id),
                    timeout=timeout)


# ------------------------------------------------------------------------------ communicators
class SingleComm:
    """world_size 1: nothing to exchange."""
    rank, size, kind = 0, 1, "single"
    device_collective = True     # nothing to exchange: the engine's fused iteration applies

    def attach(self, engine):
        pass

    def allreduce_suffstats(self, engine, moments):
        return engine.allreduce_suffstats(moments)

    def allreduce_small(self, engine, buf):
        return np.array(buf, dtype=np.float64, copy=True)

    def allreduce_host(self, buf, op="sum"):
        return np.array(buf, dtype=np.float64, copy=True)

    def spectral_reduce(self, engine):
        pass

    def allgather(self, obj):
        return [obj]

    def gather(self, obj, dst=0):
        return [obj]

    def barrier(self):
        pass


class _GroupComm:
    def __init__(self, group):
        self.group = group
        self.rank, self.size = group.rank, group.size

    def allreduce_host(self, buf, op="sum"):
        return self.group.allreduce(buf, op)

    def allgather(self, obj):
        return self.group.allgather(obj)

    def gather(self, obj, dst=0):
        """List of every rank's object (rank order) on rank `dst`, None on the others."""
        if hasattr(self.group, "gather"):
            return self.group.gather(obj, dst)
        objs = self.group.allgather(obj)
        return objs if self.rank == dst else None

    def barrier(self):
        self.group.barrier()


class HostComm(_GroupComm):
    """Engines without device collectives: pull, reduce on the host, push back."""

    def __init__(self, group):
        super().__init__(group)
        self.kind = group.kind + "-host"

    def attach(self, engine):
        pass

    def allreduce_suffstats(self, engine, moments):
        sig = engine.get_sigma_ss()
        mom = np.asarray(moments, dtype=np.float64).ravel()
        packed = np.concatenate([[engine.get_bound_total()], sig.ravel(), mom, engine.get_beta_ss().ravel()])
        red = self.group.allreduce(packed)
        n2, ne = sig.size, len(mom)
        engine.put_sigma_ss(red[1:1 + n2])
        engine.put_beta_ss(red[1 + n2 + ne:])
        return float(red[0]), red[1 + n2:1 + n2 + ne].copy()

    def allreduce_small(self, engine, buf):
        return self.group.allreduce(buf)

    def spectral_reduce(self, engine):
        """Sum of the shards' gram matrices (stm.py:122-157 is a sum over documents): pulled, reduced on the host, pushed back."""
        Q = engine.spectral_q_rows(np.arange(engine.spectral_terms(), dtype=np.int32))
        engine.spectral_put_q(self.group.allreduce(Q.ravel()).reshape(Q.shape))


class RcclComm(_GroupComm):
    """RCCL all-reduce of the device-resident sufficient statistics (one per EM iteration)."""
    kind = "rccl"
    device_collective = True     # stm_em_begin all-reduces the packed buffer on the device

    def attach(self, engine):
        uid = self.group.broadcast(engine.comm_unique_id() if self.rank == 0 else None, src=0)
        engine.comm_init(uid, self.rank, self.size)

    def allreduce_suffstats(self, engine, moments):
        return engine.allreduce_suffstats(moments)

    def allreduce_small(self, engine, buf):
        return engine.allreduce_small(buf)

    def spectral_reduce(self, engine):
        # stm_spectral_allreduce is the identity on a handle without a communicator: an engine that was never attached would
        # silently initialise from its own shard, with a different beta on every rank
        info = engine.comm_info()
        if info["nranks"] != self.size:
            raise RuntimeError(f"RcclComm.spectral_reduce: the engine's communicator has {info['nranks']} ranks, the group "
                               f"{self.size} (attach() the engine first)")
        engine.spectral_allreduce()     # the Vk x Vk matrix in place on the device (200 MB at maxV = 5000)
