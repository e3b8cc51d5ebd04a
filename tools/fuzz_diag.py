import os, sys
import numpy as np
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.init()
import oracle_ffi as O
from ranklib_amd import _native as N
from tree_equiv import node_members

target = int(sys.argv[1])
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
for case in range(target + 1):
    F = int(rng.choice([3, 8, 17, 40]))
    kind = rng.choice(["tiny", "mixed", "long"])
    if kind == "tiny":
        sizes = rng.integers(1, 17, int(rng.integers(20, 400)))
    elif kind == "mixed":
        sizes = np.concatenate([rng.integers(1, 17, 100), rng.integers(17, 200, 30), rng.integers(200, 500, 3)])
    else:
        sizes = rng.integers(100, 700, int(rng.integers(3, 12)))
    rng.shuffle(sizes)
    qoff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    n = int(qoff[-1])
    X = rng.random((n, F)).astype(np.float32)
    X[:, ::3] = np.floor(X[:, ::3] * rng.integers(2, 30))
    if F > 8:
        X[:, 5] = 0.0
    z = X[:, 0] * 0.3 + X[:, 1 % F] * X[:, 2 % F] + 0.5 * rng.random(n)
    lab = np.floor(np.clip(z / z.max() * 5, 0, 4)).astype(np.float32)
    ranker = rng.choice(["LAMBDAMART", "LAMBDAMART", "MART"])
    metric = rng.choice(["NDCG", "NDCG", "DCG", "MAP", "ERR"])
    k = int(rng.choice([1, 3, 10, 16, 25])) if metric != "MAP" else int(rng.choice([0, 5]))
    leaves = int(rng.choice([2, 3, 7, 10, 31, 64]))
    mls = int(rng.choice([1, 1, 5, 50]))
    tc = int(rng.choice([256, 256, 10, -1]))
    frate = float(rng.choice([1.0, 1.0, 0.5, 0.3]))
    lr = float(rng.choice([0.1, 0.05, 1.0]))
    rounds = int(rng.integers(2, 6))
    seed = int(rng.integers(0, 2 ** 31))
print(dict(n=n, F=F, ranker=str(ranker), metric=str(metric), k=k, leaves=leaves, mls=mls, tc=tc, frate=frate, lr=lr, rounds=rounds))
o = O.Oracle(X, lab, qoff, n_trees=rounds, n_leaves=leaves, lr=lr, n_threshold=tc, mls=mls, k=k, ranker=str(ranker), metric=str(metric), n_threads=3, frate=frate, seed=seed)
g = N.Trainer(n_trees=rounds, n_leaves=leaves, learning_rate=lr, n_threshold=tc, min_leaf_support=mls, metric_k=k, metric=str(metric), ranker=str(ranker), feature_sampling_rate=frate, seed=seed)
g.set_train(X, lab, qoff); o.init(); g.init()
for m in range(rounds):
    to, tmo, _, _ = o.round(); tg, tmg, _, _ = g.boost_round()
    lam = o.lambdas().copy()
    assert np.array_equal(g.array("LAMBDA"), lam)
    a, b = to.trimmed(), tg.trimmed()
    ma, mb = node_members(a, X, None), node_members(b, X, None)
    paths = {(0, 0): ([], [])}
    stack = [(0, 0)]
    if to.n_nodes != tg.n_nodes: print("round", m, "node counts", to.n_nodes, tg.n_nodes)
    while stack:
        na, nb = stack.pop()
        if (a["feature"][na] == -1) != (b["feature"][nb] == -1):
            docs = ma[na]
            tot = sum(Fraction(float(lam[d])) for d in docs); sq = sum(Fraction(float(lam[d])) ** 2 for d in docs)
            dev = sq - tot * tot / len(docs)
            print("round", m, "node", na, nb, "LEAF vs SPLIT: oracle leaf" if a["feature"][na] == -1 else "LEAF vs SPLIT: gpu leaf", "docs", len(docs),
                  "same docs", np.array_equal(ma[na], mb[nb]), "exact deviance %.3e" % float(dev), "distinct lambdas", len(set(lam[docs].tolist())),
                  "oracle dev", a["deviance"][na], "gpu dev", b["deviance"][nb])
            def devs(tr, mem):
                out = []
                for i, d in mem.items():
                    if len(d) == 0: continue
                    tt = sum(Fraction(float(lam[x])) for x in d); qq = sum(Fraction(float(lam[x])) ** 2 for x in d)
                    out.append((float(qq - tt * tt / len(d)), len(d), "leaf" if tr["feature"][i] == -1 else "split", i))
                return sorted(out, reverse=True)
            da, db = devs(a, ma), devs(b, mb)
            print("   oracle nodes by exact deviance:", [(round(x[0], 6), x[1], x[2], x[3]) for x in da if x[2] == "leaf"][:6], "... splits with the smallest deviance:", [(round(x[0], 6), x[1], x[3]) for x in da if x[2] == "split"][-4:])
            print("   gpu    nodes by exact deviance:", [(round(x[0], 6), x[1], x[2], x[3]) for x in db if x[2] == "leaf"][:6], "... splits with the smallest deviance:", [(round(x[0], 6), x[1], x[3]) for x in db if x[2] == "split"][-4:])
            sys.exit(0)
        if a["feature"][na] == -1 or b["feature"][nb] == -1:
            continue
        al, ar, bl, br = int(a["left"][na]), int(a["right"][na]), int(b["left"][nb]), int(b["right"][nb])
        pa, pb = paths[(na, nb)]
        if np.array_equal(ma[al], mb[bl]):
            stack += [(al, bl), (ar, br)]; paths[(al, bl)] = (pa + [0], pb + [0]); paths[(ar, br)] = (pa + [1], pb + [1])
        elif np.array_equal(ma[al], mb[br]):
            stack += [(al, br), (ar, bl)]; paths[(al, br)] = (pa + [0], pb + [1]); paths[(ar, bl)] = (pa + [1], pb + [0])
        else:
            docs = ma[na]
            def exactS(left):
                sl = sum(Fraction(float(lam[d])) for d in left); st = sum(Fraction(float(lam[d])) for d in docs)
                cl, cr = len(left), len(docs) - len(left)
                return sl * sl / cl + (st - sl) * (st - sl) / cr
            Sa, Sb = exactS(ma[al]), exactS(mb[bl])
            print("round", m, "node", na, nb, "docs", len(docs), "oracle split f", a["feature"][na], a["threshold"][na], "left", len(ma[al]),
                  "| gpu split f", b["feature"][nb], b["threshold"][nb], "left", len(mb[bl]))
            print("   exact S oracle-choice %.17g  gpu-choice %.17g  rel diff %.3e" % (float(Sa), float(Sb), float(abs(Sa - Sb) / max(abs(Sa), abs(Sb)))))
            L = O.lib()
            for nm, pth in (("oracle", pa), ("gpu", pb)):
                h = L.ro_root_hash(seed, m)
                for sd in pth: h = L.ro_child_hash(h, sd)
                order = O.feature_order(h, F, frate).tolist() if frate < 1 else list(range(F))
                fo = [i for i in range(F)]
                print("   path of the %s tree %s: drawn %s ; oracle's feature idx %d in draw: %s, gpu's feature idx %d in draw: %s" % (
                      nm, pth, order, a["feature"][na] - 1, (a["feature"][na] - 1) in order, b["feature"][nb] - 1, (b["feature"][nb] - 1) in order))
            print("   same node docs:", np.array_equal(ma[na], mb[nb]), " max|lambda| %.3e  node max|lambda| %.3e" % (np.abs(lam).max(), np.abs(lam[docs]).max()))
            sys.exit(0)
print("no mismatch")
