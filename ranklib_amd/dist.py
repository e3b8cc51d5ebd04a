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


    def alltoallv(self, send, recv_bytes):
        """send[p]: uint8 array for rank p; recv_bytes[p]: bytes rank p sends here -> list of received uint8 arrays (own part included).
        Point-to-point over the group (gloo has no all_to_all for CPU tensors): posted in one batch, so there is no ordering to deadlock on."""
        torch, dist = self.torch, self.dist
        me = dist.get_rank(self.group)
        out = [np.empty(int(recv_bytes[p]), np.uint8) for p in range(self.n)]
        out[me][:] = send[me]
        keep, ops = [], []
        for p in range(self.n):
            if p == me:
                continue
            if len(send[p]):
                ts = torch.from_numpy(np.ascontiguousarray(send[p])); keep.append(ts)
                ops.append(dist.P2POp(dist.isend, ts, dist.get_global_rank(self.group, p) if self.group is not None else p, group=self.group))
            if recv_bytes[p]:
                tr = torch.from_numpy(out[p])
                ops.append(dist.P2POp(dist.irecv, tr, dist.get_global_rank(self.group, p) if self.group is not None else p, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return out


# ---- the leaf-owner exchange, restated on the host (what rl_trainer.hip enqueue_round + rl_dist.inc do per round; CPU tests) ----
def leaf_owners(glen, n_ranks):
    """owner rank of every leaf: largest leaf (documents over all ranks) first onto the least loaded rank, lowest rank on a tie;
    empty leaf slots go to l % n_ranks.  Every rank computes this from the same gathered leaf tables."""
    glen = [int(v) for v in glen]
    own, load = [0] * len(glen), [0] * n_ranks
    for l in sorted(range(len(glen)), key=lambda l: -glen[l]):          # stable: ties keep ascending leaf order
        if glen[l] == 0:
            own[l] = l % n_ranks
            continue
        o = min(range(n_ranks), key=lambda r: (load[r], r))
        own[l] = o
        load[o] += glen[l]
    return own


def leaf_exchange_plan(lens, me):
    """lens[r][l] = documents of leaf l on rank r.  -> (own, send blocks per destination, receive blocks per source) for rank `me`:
    a block is (leaf, documents); rank r's block of leaf l holds its lambda segment then its weight segment (2 * 8 * documents bytes)."""
    R, L = len(lens), len(lens[0])
    own = leaf_owners([sum(lens[r][l] for r in range(R)) for l in range(L)], R)
    send = [[(l, lens[me][l]) for l in range(L) if own[l] == d] for d in range(R)]
    recv = [[(l, lens[r][l]) for l in range(L) if own[l] == me] for r in range(R)]
    return own, send, recv


# ---- exact 128-bit sums over int64 limbs (what the library does on the device; restated for the CPU tests) ----
LIMB_SHIFT = 44


def to_limbs(v):
    """python int (|v| < 2^107) -> (a, b) int64 limbs with v == (a << 44) + b, 0 <= b < 2^44"""
    return v >> LIMB_SHIFT, v & ((1 << LIMB_SHIFT) - 1)


def from_limbs(a, b):
    return (int(a) << LIMB_SHIFT) + int(b)
