"""Document sharding and the sufficient-statistic exchange for multi-GPU runs.

The reference has no distributed code.  Documents are
independent given (beta, mu_d, siginv) -- reference src/modules/stm.py:519-588
has no cross-document dependence except the two accumulators and the bound --
so every rank owns a contiguous, nnz-balanced range of documents and ONE
all-reduce per EM iteration sums

    [ bound | sigma_ss | regression moments | beta_ss ]

after which each rank finishes the (tiny) M-step redundantly.  A second, (K-1)^2
all-reduce carries the covariance of (eta - mu), which needs the reduced gamma.

Two interchangeable communicators:
  * ``RcclComm`` -- the product path: RCCL all-reduce on device buffers through the
    C-ABI (stm_comm_init / stm_allreduce_*), bootstrapped over torch.distributed
    (gloo) which is used ONLY to ship the 128-byte ncclUniqueId.
  * ``GlooComm`` -- host numpy all-reduce over torch.distributed/gloo; used by the
    CPU tests (world_size 2) and as a fallback when RCCL cannot initialise.
"""
import os

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


class SingleComm:
    """world_size 1: nothing to exchange."""
    rank, size, kind = 0, 1, "single"

    def attach(self, engine):
        pass

    def allreduce_suffstats(self, engine, extra):
        return engine.allreduce_suffstats(extra)

    def allreduce_small(self, engine, buf):
        return np.array(buf, dtype=np.float64, copy=True)

    def allreduce_host(self, buf):
        return np.array(buf, dtype=np.float64, copy=True)

    def barrier(self):
        pass


class GlooComm:
    """Host all-reduce over an initialised torch.distributed (gloo) process group."""
    kind = "gloo-host"

    def __init__(self):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist = torch, dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self._group = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else None

    def attach(self, engine):
        pass

    def allreduce_host(self, buf):
        t = self._torch.from_numpy(np.array(buf, dtype=np.float64, copy=True))
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
        return t.numpy()

    def allreduce_suffstats(self, engine, extra):
        """Engines without device collectives: pull, reduce on the host, push back."""
        packed = np.concatenate([[engine.get_bound_total()], engine.get_sigma_ss().ravel(),
                                 np.asarray(extra, dtype=np.float64).ravel(), engine.get_beta_ss().ravel()])
        red = self.allreduce_host(packed)
        n2 = engine.get_sigma_ss().size
        ne = len(np.asarray(extra).ravel())
        engine.put_sigma_ss(red[1:1 + n2])
        engine.put_beta_ss(red[1 + n2 + ne:])
        return float(red[0]), red[1 + n2:1 + n2 + ne].copy()

    def allreduce_small(self, engine, buf):
        return self.allreduce_host(buf)

    def barrier(self):
        self._dist.barrier(group=self._group)


class RcclComm:
    """RCCL all-reduce of the device-resident sufficient statistics (one per EM iteration)."""
    kind = "rccl"

    def __init__(self):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (it ships the ncclUniqueId)")
        self._dist = dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self._host = GlooComm()

    def attach(self, engine):
        uid = [engine.comm_unique_id() if self.rank == 0 else None]
        self._dist.broadcast_object_list(uid, src=0, group=self._host._group)
        engine.comm_init(uid[0], self.rank, self.size)

    def allreduce_suffstats(self, engine, extra):
        return engine.allreduce_suffstats(extra)

    def allreduce_small(self, engine, buf):
        return engine.allreduce_small(buf)

    def allreduce_host(self, buf):
        return self._host.allreduce_host(buf)

    def barrier(self):
        self._host.barrier()


def init_from_env(backend="gloo"):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import datetime

    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=600))
    return dist.get_rank(), dist.get_world_size()
