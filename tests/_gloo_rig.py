"""The host-group interface of strutopy_amd.dist over an initialised torch.distributed (gloo) process group.

TEST INFRASTRUCTURE (tests/test_dist_gloo.py): the product's own group is strutopy_amd.dist.TcpGroup and never imports
torch."""
import numpy as np


class GlooGroup:
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

    def gather(self, obj, dst=0):
        out = [None] * self.size if self.rank == dst else None
        self._dist.gather_object(obj, out, dst=dst, group=self._group)
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


def GlooComm():
    """HostComm over torch.distributed/gloo."""
    from strutopy_amd.dist import HostComm
    return HostComm(GlooGroup())
