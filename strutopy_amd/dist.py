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
    PyTorch), a star through rank 0, found through ``STM_RDZV_PORT`` or a
    rendezvous file keyed by MASTER_ADDR / MASTER_PORT.  ``GlooGroup`` wraps an
    initialised torch.distributed process group (CPU test rig).
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
import os
import pickle
import socket
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
def _send_msg(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
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
    return pickle.loads(_recv_exact(sock, n))


def rendezvous_file(addr, port, run_id=""):
    """Where rank 0 publishes its listening port when STM_RDZV_PORT is not given (ranks share one node)."""
    tag = f"{addr}_{port}_{run_id}".replace("/", "_").replace(":", "_")
    return os.path.join(tempfile.gettempdir(), f"stm_rdzv_{tag}")


class TcpGroup:
    """Host-side group over stdlib TCP sockets: every collective is a gather to rank 0 and a reply.

    Payloads are small (a 128-byte ncclUniqueId, scalars, at most the packed sufficient statistics
    in the HostComm fallback), so a star is enough; the data path between GPUs is RCCL.
    """
    kind = "tcp"

    def __init__(self, rank, size, addr="127.0.0.1", port=None, rdzv_file=None, timeout=600.0):
        self.rank, self.size = int(rank), int(size)
        self._peers = []       # rank 0: sockets of ranks 1..size-1 (index r-1)
        self._root = None      # other ranks: socket to rank 0
        self._file = None
        if self.size <= 1:
            return
        if port is None and rdzv_file is None:
            raise ValueError("TcpGroup needs a port or a rendezvous file")
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, int(port) if port is not None else 0))
            srv.listen(self.size)
            token = os.urandom(8).hex()
            if port is None:
                tmp = rdzv_file + f".{os.getpid()}.tmp"
                with open(tmp, "w") as fh:
                    fh.write(f"{srv.getsockname()[1]} {token}\n")
                os.replace(tmp, rdzv_file)      # atomic publish
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
                c.settimeout(timeout)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                hello = _recv_msg(c)
                if not (isinstance(hello, tuple) and len(hello) == 3 and hello[0] == "stm-hello" and hello[2] == self.size
                        and 0 < hello[1] < self.size and hello[1] not in peers):
                    c.close()
                    continue
                _send_msg(c, ("stm-welcome", token))
                peers[hello[1]] = c
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
                p, token = port, None
                if p is None:
                    try:
                        with open(rdzv_file) as fh:
                            p, token = fh.read().split()
                    except (OSError, ValueError):
                        time.sleep(0.05)
                        continue
                try:
                    s = socket.create_connection((addr, int(p)), timeout=5.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _send_msg(s, ("stm-hello", self.rank, self.size))
                    reply = _recv_msg(s)
                    if isinstance(reply, tuple) and reply[0] == "stm-welcome" and (token is None or reply[1] == token):
                        s.settimeout(timeout)
                        self._root = s
                        break
                    s.close()
                except (OSError, ConnectionError, pickle.UnpicklingError, struct.error):
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


class GlooGroup:
    """The same interface over an initialised torch.distributed (gloo) process group (CPU test rig)."""
    kind = "gloo"

    def __init__(self):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist = torch, dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self._group = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else None

    def allgather(self, obj):
        out = [None] * self.size
        self._dist.all_gather_object(out, obj, group=self._group)
        return out

    def allreduce(self, buf, op="sum"):
        t = self._torch.from_numpy(np.array(buf, dtype=np.float64, copy=True))
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM if op == "sum" else self._dist.ReduceOp.MAX, group=self._group)
        return t.numpy()

    def broadcast(self, obj, src=0):
        box = [obj if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src, group=self._group)
        return box[0]

    def barrier(self):
        self._dist.barrier(group=self._group)

    def close(self):
        pass


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

    def allgather(self, obj):
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


def GlooComm():
    """HostComm over torch.distributed/gloo (tests/test_dist_gloo.py)."""
    return HostComm(GlooGroup())
