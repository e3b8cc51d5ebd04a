#!/usr/bin/env python
"""Where the default path (with the lazy Java-order tie-break, rl_tie.inc) and the oracle store different (feature, threshold) pairs.
  python tools/tie_diag.py n_docs n_feat kind leaves mls rounds seed
For every differing split: node size, both candidates, whether the two cuts are the same / mirrored sets, and the trainer's TIE_STATS."""
import os
import sys

os.environ["RLHIP_STEPLOG"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_ffi as O
from ranklib_amd import _native as N
from ranklib_amd import synth
from tree_equiv import node_members


def java_restatement(tree, members, x, lam, bins, nbins, thr, mls):
    """S of every candidate of node x in the Java's own summation order (FeatureHistogram.java:126-146,166-195,222-234,236-264), in plain Python"""
    parent = {}
    for n in range(len(tree["feature"])):
        if tree["feature"][n] != -1:
            parent[int(tree["left"][n])] = (n, True); parent[int(tree["right"][n])] = (n, False)
    F = bins.shape[0]

    def direct(n):          # (cumulative sums [F][T], sumResponse) accumulated from the node's samples in ascending order
        cum, tot = [], 0.0
        for k in members[n]:
            tot += lam[k]
        for f in range(F):
            T = int(nbins[f]); sm = [0.0] * T
            for k in members[n]:
                sm[int(bins[f][k])] += lam[k]
            for t in range(1, T):
                sm[t] += sm[t - 1]
            cum.append(sm)
        return cum, tot

    def hist(n):
        if n == 0 or parent[n][1]:
            return direct(n)
        p = parent[n][0]
        pc, pt = hist(p)
        lc, lt = direct(int(tree["left"][p]))
        return [[pc[f][t] - lc[f][t] for t in range(len(pc[f]))] for f in range(F)], pt - lt

    cum, tot = hist(x)
    cnt = [np.cumsum(np.bincount(bins[f][members[x]].astype(np.int64), minlength=int(nbins[f]))) for f in range(F)]
    n = len(members[x])
    best, top = (-1.0, -1, -1), []
    for f in range(F):
        for t in range(int(nbins[f])):
            cl = int(cnt[f][t]); cr = n - cl
            if cl < mls or cr < mls:
                continue
            sl = cum[f][t]; sr = tot - sl
            S = sl * sl / cl + sr * sr / cr
            top.append((S, f, t))
            if best[0] < S:
                best = (S, f, t)
    top.sort(key=lambda v: -v[0])
    print("    python restatement of the Java: best (f %d, t %d, thr %r) S %r; next: %s" %
          (best[1] + 1, best[2], float(thr[best[1]][best[2]]), best[0], ", ".join("(f %d t %d %r)" % (f + 1, t, S) for S, f, t in top[:6])))


def main():
    n_docs, n_feat, kind, leaves, mls, rounds, seed = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    tc = int(sys.argv[8]) if len(sys.argv) > 8 else 256
    ranker = sys.argv[9] if len(sys.argv) > 9 else "LAMBDAMART"
    X, lab, qoff = synth.make_dataset(n_docs, n_feat, kind, seed_offset=seed)
    o = O.Oracle(X, lab, qoff, n_trees=rounds, n_leaves=leaves, mls=mls, n_threshold=tc, ranker=ranker)
    g = N.Trainer(n_trees=rounds, n_leaves=leaves, min_leaf_support=mls, n_threshold=tc, ranker=ranker)
    g.set_train(X, lab, qoff)
    o.init(); g.init()
    bins, nbins, thr = g.array("BINS"), g.array("NBINS"), g.array("THRESHOLDS")
    for r in range(rounds):
        to, _, _, _ = o.round()
        tg, _, _, _ = g.boost_round()
        lam = g.array("LAMBDA")
        a, b = to.trimmed(), tg.trimmed()
        ma, mb = node_members(a, X), node_members(b, X)
        stack = [(0, 0, 0, "root")]
        while stack:
            na, nb, depth, path = stack.pop()
            if a["feature"][na] == -1 or b["feature"][nb] == -1:
                continue
            al, ar, bl, br = int(a["left"][na]), int(a["right"][na]), int(b["left"][nb]), int(b["right"][nb])
            same = a["feature"][na] == b["feature"][nb] and np.float32(a["threshold"][na]).view(np.uint32) == np.float32(b["threshold"][nb]).view(np.uint32)
            if np.array_equal(ma[al], mb[bl]):
                kind_ = "same sets"; nxt = [(al, bl, depth + 1, path + "L"), (ar, br, depth + 1, path + "R")]
            elif np.array_equal(ma[al], mb[br]):
                kind_ = "MIRRORED sets"; nxt = [(al, br, depth + 1, path + "L"), (ar, bl, depth + 1, path + "R")]
            else:
                print("round %d node %s: DIFFERENT partitions" % (r, path)); break
            if not same:
                print("round %d node %s (%d docs, left %d): oracle (f %d, thr %r)  gpu (f %d, thr %r)  %s" %
                      (r, path, len(ma[na]), len(ma[al]), a["feature"][na], float(a["threshold"][na]), b["feature"][nb], float(b["threshold"][nb]), kind_))
                java_restatement(a, ma, na, lam, bins, nbins, thr, mls)
            stack += nxt
        print("round %d: TIE_STATS %s" % (r, g.array("TIE_STATS").tolist()))
        log = g.array("STEP_LOG"); ne = min(int(log[0]), 8192); e = log[8:8 + 8 * ne].reshape(ne, 8)
        for row in e[(e[:, 1] == 0) & (e[:, 4] <= 20)]:
            print("    prepared node: step %d slot %d, %d docs (accumulated child %d), tie flag 0x%x" % (row[2], row[3], row[4], row[5], row[6]))
        for row in e[e[:, 1] == 1]:
            print("    committed tied split: tie flag 0x%x (0x80 = resolved by the host), right child %d, %d docs, chain of %d nodes" % (row[2], row[3] & 1, row[4], row[6]))


if __name__ == "__main__":
    main()
