"""Multi-GPU plumbing: images shard one-extrinsic-block-per-GPU; the only
exchange is an all-reduce (sum, FP64) of the 68-double reduced intrinsic system per LM
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
    hook is wrapped once as a tensor.  The collective is issued on the stream the library passes
    (the context's own non-blocking stream unless `Solver(stream=...)` chose another): that is the
    stream `k_lm_row` has just written the buffer on and the next `k_lm_trial` will read it on, so
    the all-reduce is ordered between the two by the stream itself.  (Issued on torch's current
    stream instead, nothing would order it against a non-blocking library stream.)"""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self._views = {}
        self._streams = {}
        self.calls = 0

    def __call__(self, ptr, n, stream):
        t = self._views.get((ptr, n))
        if t is None:
            t = self.torch.as_tensor(_DevView(ptr, n), device="cuda")
            self._views[(ptr, n)] = t
        if t.is_cuda and stream:
            ext = self._streams.get(stream)
            if ext is None:
                ext = self._streams[stream] = self.torch.cuda.ExternalStream(int(stream))
            with self.torch.cuda.stream(ext):
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        else:      # legacy default stream (handle 0): torch's current stream is the one to use
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


def broadcast_id_via_torch(group=None):
    """-> broadcast_bytes callable for Solver.init_rccl: ships rank 0's 128-byte RCCL id."""
    import torch
    import torch.distributed as dist

    def bcast(b):
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.zeros(128, dtype=torch.uint8, device=dev)
        if b is not None:
            t.copy_(torch.frombuffer(bytearray(b), dtype=torch.uint8))
        dist.broadcast(t, src=0, group=group)
        return bytes(t.cpu().numpy().tobytes())
    return bcast


def allgather_bytes_via_torch(group=None):
    """-> allgather_bytes callable for Solver.init_p2p: every rank's inbox handle, in rank order."""
    import torch
    import torch.distributed as dist

    def gather(b):
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
        dist.all_gather(out, t, group=group)
        return [bytes(o.cpu().numpy().tobytes()) for o in out]
    return gather


def reduced_system_from_gram(G, lam, mask=None):
    """numpy mirror of one image's contribution to the 68-double reduced system of the device
    kernels: A, g, SC = B C_l^-1 B^T, sg = B C_l^-1 h, err.  Used by the CPU
    tests of the sharding logic; G is the 16x16 Gram [Je|Ji|e]^T[Je|Ji|e]."""
    C = G[:6, :6].copy()
    C[np.diag_indices(6)] *= 1.0 + lam
    B = G[6:15, :6]
    h = G[:6, 15]
    CiBt = np.linalg.solve(C, B.T)
    Cih = np.linalg.solve(C, h)
    return dict(A=G[6:15, 6:15].copy(), g=G[6:15, 15].copy(), SC=B @ CiBt, sg=B @ Cih, err=float(G[15, 15]))


def pack_reduced(parts):
    """sum of per-image contributions -> flat vector (what is all-reduced)."""
    A = sum(p["A"] for p in parts)
    g = sum(p["g"] for p in parts)
    SC = sum(p["SC"] for p in parts)
    sg = sum(p["sg"] for p in parts)
    err = sum(p["err"] for p in parts)
    return np.concatenate([A.ravel(), g, SC.ravel(), sg, [err]])


def solve_reduced(vec, lam, mask):
    """damped Schur system of the device's LM advance (k_lm_trial prologue): -> delta_int (9), err."""
    A, g = vec[:81].reshape(9, 9), vec[81:90]
    SC, sg, err = vec[90:171].reshape(9, 9), vec[171:180], vec[180]
    S = A - SC
    S[np.diag_indices(9)] = np.diag(A) * (1.0 + lam) - np.diag(SC)
    rhs = g - sg
    free = np.array(mask, dtype=bool) & (np.diag(A) > 0)
    d = np.zeros(9)
    d[free] = np.linalg.solve(S[np.ix_(free, free)], rhs[free])
    return d, err
