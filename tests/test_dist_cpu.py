"""CPU (gloo, world_size 2) coverage of the multi-rank host logic: query sharding, the torch host transport that
plugs into rl_dist_init_callback, and the int64-limb representation that makes the histogram all-reduce exact."""
import os
import random
import subprocess
import sys

import numpy as np

from ranklib_amd import dist as D
from ranklib_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_contiguous_balanced_and_complete():
    _, _, qoff = synth.make_dataset(50000, 2, "mslr")
    for R in (1, 2, 3, 8):
        parts = D.partition_queries(qoff, R)
        assert parts[0][0] == 0 and parts[-1][1] == len(qoff) - 1
        assert all(parts[i][1] == parts[i + 1][0] for i in range(R - 1))
        sizes = [int(qoff[e] - qoff[b]) for b, e in parts]
        assert min(sizes) > 0 and max(sizes) - min(sizes) < 2 * 1251
    X = np.arange(20, dtype=np.float32).reshape(10, 2)
    lab = np.arange(10, dtype=np.float32)
    q = np.array([0, 3, 4, 8, 10], np.int32)
    got = [D.shard(X, lab, q, r, 2) for r in range(2)]
    assert np.array_equal(np.concatenate([g[0] for g in got]), X)
    assert list(got[0][2]) == [0, 3, 4, 8] and list(got[1][2]) == [0, 2]


def test_limbs_are_exact():
    rnd = random.Random(5)
    for _ in range(2000):
        v = rnd.randrange(-(1 << 100), 1 << 100)
        a, b = D.to_limbs(v)
        assert 0 <= b < (1 << 44) and -(1 << 63) <= a < (1 << 63)
        assert D.from_limbs(a, b) == v


WORKER = r'''
import os, sys, random
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from ranklib_amd import dist as D
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
tr = D.TorchHostTransport()
# 1. exact histogram exchange: every rank holds 128-bit partial totals; limbs are summed as plain int64
rnd = random.Random(100 + rank)
mine = [rnd.randrange(-(1 << 75), 1 << 75) for _ in range(500)]
limbs = np.array([x for v in mine for x in D.to_limbs(v)], dtype=np.int64)
tr.allreduce(limbs, tr.SUM)
total = [D.from_limbs(limbs[2 * i], limbs[2 * i + 1]) for i in range(500)]
ref = [sum(random.Random(100 + r).randrange(-(1 << 75), 1 << 75) for _ in range(1)) for r in range(world)]
allv = [[random.Random(100 + r).randrange(-(1 << 75), 1 << 75) for _ in range(500)] for r in range(world)]
for r in range(world):
    rr = random.Random(100 + r); allv[r] = [rr.randrange(-(1 << 75), 1 << 75) for _ in range(500)]
assert total == [sum(allv[r][i] for r in range(world)) for i in range(500)]
# 2. unsigned keys: order-preserving max / min (float keys are >= 2^31 for positive floats)
k = np.array([0x80000001 + rank, 0x3fffffff - rank, 0xfffffff0 + rank], dtype=np.uint32)
kmax = k.copy(); tr.allreduce(kmax, tr.MAX)
kmin = k.copy(); tr.allreduce(kmin, tr.MIN)
assert list(kmax) == [0x80000001 + world - 1, 0x3fffffff, 0xfffffff0 + world - 1]
assert list(kmin) == [0x80000001, 0x3fffffff - (world - 1), 0xfffffff0]
m = np.array([np.float64(1.5 + rank).view(np.uint64)], dtype=np.uint64); tr.allreduce(m, tr.MAX)
assert m.view(np.float64)[0] == 1.5 + world - 1
# 3. all-gather keeps rank order
g = tr.allgather(np.full(5, rank, np.uint8))
assert list(g) == [r for r in range(world) for _ in range(5)]
# 4. the leaf-owner exchange (ranklib_amd/dist.py leaf_exchange_plan = what rl_trainer.hip does per round): documents of a global data set
#    live on the ranks in contiguous ascending ranges; every leaf's values must reach its owner in global document order
rg = np.random.RandomState(7)
Ndoc, L = 5000, 9
leaf_of = rg.randint(0, L - 2, Ndoc)                 # two leaf slots stay empty
val = rg.rand(Ndoc, 2)                                # (lambda, weight) per document
cut = [0] + sorted(rg.choice(np.arange(1, Ndoc), world - 1, replace=False).tolist()) + [Ndoc]
lens = [[int(np.sum(leaf_of[cut[r]:cut[r + 1]] == l)) for l in range(L)] for r in range(world)]
own, send, recv = D.leaf_exchange_plan(lens, rank)
loads = [sum(sum(lens[r][l] for r in range(world)) for l in range(L) if own[l] == o) for o in range(world)]
big = max(sum(lens[r][l] for r in range(world)) for l in range(L))
assert max(loads) <= max(big, -(-Ndoc // world) + big), (loads, big)          # LPT bound: no worse than the mean plus the largest leaf
mine = np.arange(cut[rank], cut[rank + 1])
def block(l):                                         # this rank's block of leaf l: lambda segment, then weight segment
    d = mine[leaf_of[mine] == l]
    return np.concatenate([val[d, 0], val[d, 1]])
sbuf = [np.concatenate([block(l) for l, n in send[d]] + [np.zeros(0)]).view(np.uint8) for d in range(world)]
got = tr.alltoallv(sbuf, [16 * sum(n for _, n in recv[r]) for r in range(world)])
for l in range(L):
    if own[l] != rank:
        continue
    lam, w = [], []
    for r in range(world):
        buf = got[r].view(np.float64); off = 0
        for l2, n in recv[r]:
            if l2 == l:
                lam.append(buf[off:off + n]); w.append(buf[off + n:off + 2 * n])
            off += 2 * n
    d = np.nonzero(leaf_of == l)[0]                   # global document order inside the leaf
    assert np.array_equal(np.concatenate(lam), val[d, 0]) and np.array_equal(np.concatenate(w), val[d, 1])
recv_bytes = sum(len(got[r]) for r in range(world) if r != rank)
assert recv_bytes <= 16 * max(loads)                  # never the 16 * Ndoc * (world - 1) / world an all-gather of every document would deliver ... unless one leaf is that big
dist.barrier()
sys.stdout.write("ok%d\n" % rank); sys.stdout.flush()
'''


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_host_transport_over_gloo(world, tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29505 + world), str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert all(("ok%d" % k) in r.stdout for k in range(world))


def test_leaf_owners_are_balanced_and_rank_invariant():
    rg = np.random.RandomState(3)
    for R in (1, 2, 3, 8):
        for _ in range(50):
            glen = (rg.pareto(1.2, 31) * 1000).astype(np.int64)
            glen[rg.randint(0, 31, 4)] = 0
            own = D.leaf_owners(glen, R)
            assert own == D.leaf_owners(list(glen), R) and all(0 <= o < R for o in own)
            load = [int(sum(glen[l] for l in range(31) if own[l] == o)) for o in range(R)]
            assert max(load) <= max(int(glen.max()), -(-int(glen.sum()) // R) + int(glen.max()))
