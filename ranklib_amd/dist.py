"""Multi-GPU helpers: query sharding and the two transports behind rl_dist_init / rl_dist_init_callback.

One process per GPU.  Queries are sharded CONTIGUOUSLY (rank r owns a run of consecutive ranked lists), balanced by
document count -- the same shape as MyThreadPool.partition over queries (utilities/MyThreadPool.java:77-87,
learning/tree/LambdaMART.java:341-354), and it keeps "rank order == global document order", which the exact float
running sums rely on.
"""
import numpy as np


def partition_queries(qoff, n_ranks):
    """-> list of (q_begin, q_end) per rank: contiguous, every rank non-empty, balanced by documents"""
    qoff = np.asarray(qoff, dtype=np.int64)
    Q = len(qoff) - 1
    if n_ranks > Q:
        raise ValueError("more ranks than ranked lists")
    n = int(qoff[-1])
    cuts = [0]
    for r in range(1, n_ranks):
        target = n * r / n_ranks
        q = int(np.searchsorted(qoff, target, side="left"))
        q = max(q, cuts[-1] + 1)
        q = min(q, Q - (n_ranks - r))
        cuts.append(q)
    cuts.append(Q)
    return [(cuts[r], cuts[r + 1]) for r in range(n_ranks)]


def shard(X, labels, qoff, rank, n_ranks):
    """the (X, labels, qoff) slice of `rank`"""
    qb, qe = partition_queries(qoff, n_ranks)[rank]
    qoff = np.asarray(qoff)
    d0, d1 = int(qoff[qb]), int(qoff[qe])
    return X[d0:d1], labels[d0:d1], (qoff[qb:qe + 1] - qoff[qb]).astype(np.int32)


class TorchHostTransport:
    """all-reduce / all-gather of host numpy buffers over a torch.distributed process group (e.g. gloo).
    Plugs into Trainer.dist_init_callback; also usable on its own (tests/test_dist_cpu.py)."""
    SUM, MAX, MIN = 0, 1, 2

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.n = dist.get_world_size(group)

    def allreduce(self, arr, op):
        torch, dist = self.torch, self.dist
        rop = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}[op]
        if arr.dtype in (np.uint32, np.uint64):
            if arr.dtype == np.uint64 and (arr >> np.uint64(63)).any():
                raise ValueError("uint64 value above 2^63 cannot travel as int64")
            t = torch.from_numpy(arr.astype(np.int64))          # order-preserving widening
            dist.all_reduce(t, op=rop, group=self.group)
            arr[:] = t.numpy().astype(arr.dtype)
        else:
            t = torch.from_numpy(arr)                           # shares memory: reduced in place
            dist.all_reduce(t, op=rop, group=self.group)

    def allgather(self, src_u8):
        torch, dist = self.torch, self.dist
        t = torch.from_numpy(np.ascontiguousarray(src_u8))
        outs = [torch.empty_like(t) for _ in range(self.n)]
        dist.all_gather(outs, t, group=self.group)
        return np.concatenate([o.numpy() for o in outs])


# ---- exact 128-bit sums over int64 limbs (what the library does on the device; restated for the CPU tests) ----
LIMB_SHIFT = 44


def to_limbs(v):
    """python int (|v| < 2^107) -> (a, b) int64 limbs with v == (a << 44) + b, 0 <= b < 2^44"""
    return v >> LIMB_SHIFT, v & ((1 << LIMB_SHIFT) - 1)


def from_limbs(a, b):
    return (int(a) << LIMB_SHIFT) + int(b)
