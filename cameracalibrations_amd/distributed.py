"""Multi-GPU plumbing: images shard one-extrinsic-block-per-GPU; the only
exchange is an all-reduce (sum, FP64) of the 112-double reduced intrinsic system per LM
trial, plus one 8-double all-reduce for the initial focal lengths.  One process per GPU,
`torch.distributed` (backend "nccl" == RCCL over xGMI; "gloo" for the CPU tests).
"""
import numpy as np


def shard_range(n_images, world_size, rank):
    """Contiguous image shard [lo, hi) of rank `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_images), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _DevView:
    """Zero-copy torch view of a raw device pointer via the CUDA array interface."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "version": 2,
                                         "data": (int(ptr), False), "strides": None}


class TorchAllReduce:
    """ccal all-reduce hook backed by torch.distributed.  The device buffer handed to the
    hook is wrapped once as a tensor; RCCL orders itself after the kernels already queued on
    the current stream and the next kernels after it."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self._views = {}
        self.calls = 0

    def __call__(self, ptr, n, stream):
        t = self._views.get((ptr, n))
        if t is None:
            t = self.torch.as_tensor(_DevView(ptr, n), device="cuda")
            self._views[(ptr, n)] = t
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        self.calls += 1
        return 0


def host_allreduce_sum(arr, group=None):
    """All-reduce a host numpy array in place (gloo; CPU tests of the sharding logic)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(arr))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    arr[...] = t.numpy()
    return arr
