#!/usr/bin/env python
"""Companion of fuzz_parity.py for a 'lambda' mismatch: python tools/fuzz_lambda_diag.py <case> <seed> -- which documents differ and how"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.init()
import oracle_ffi as O
from ranklib_amd import _native as N

target = int(sys.argv[1])
rng = np.random.default_rng(int(sys.argv[2]))
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
    sc_before = o.scores().copy()
    o.round(); g.boost_round()
    lo, lg = o.lambdas().copy(), g.array("LAMBDA")
    wo, wg = o.weights().copy(), g.array("WEIGHT")
    bad = np.nonzero((lo.view(np.int64) != lg.view(np.int64)) | (wo.view(np.int64) != wg.view(np.int64)))[0]
    print("round", m, "lambda/weight mismatches:", len(bad), "scores equal:", np.array_equal(o.scores(), g.array("SCORE")))
    if len(bad):
        q = np.searchsorted(qoff, bad[0], side="right") - 1
        a, b = qoff[q], qoff[q + 1]
        print(" first bad doc", bad[0], "query", q, "docs", b - a, "lambda oracle %.17g gpu %.17g  weight oracle %.17g gpu %.17g" % (lo[bad[0]], lg[bad[0]], wo[bad[0]], wg[bad[0]]))
        print(" bad docs in that query:", [int(x - a) for x in bad if a <= x < b][:20], "queries affected:", len(set((np.searchsorted(qoff, bad, side='right') - 1).tolist())))
        s = sc_before[a:b]
        print(" scores of the query before the round: min %.6g max %.6g, largest |difference| %.6g, any non-finite: %s" % (s.min(), s.max(), s.max() - s.min(), not np.isfinite(s).all()))
        print(" labels", lab[a:b][:30])
        break
