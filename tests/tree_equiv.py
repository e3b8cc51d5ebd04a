"""Tree comparison used by the GPU parity tests.

Everything is compared bit for bit, with ONE documented exception (DESIGN.md "exact-arithmetic ties"):

Split candidates that cut a node into EXACTLY THE SAME two sample sets have equal gain
S = sl^2/cl + sr^2/cr in exact arithmetic (S is also symmetric under swapping the sides).  In the Java
their S values differ by f64 rounding noise whenever the sums were grouped differently -- a histogram
derived by subtraction (right children, learning/tree/FeatureHistogram.java:222-234) on an empty-bin
plateau, or two features that isolate the same samples of a small node -- so Java's strict-'<' arg-max
(FeatureHistogram.java:255) lands on a noise-chosen member of the tie.  The GPU accumulates exactly, sees
the tie and keeps the first candidate in scan order (lowest feature index, then lowest threshold).

Such a pair of trees is the same function on the training set: same hierarchy of sample sets (possibly
with the two children of a node mirrored), same leaf values.  `assert_equivalent` verifies exactly that
by replaying both trees on the training rows, and reports how many splits were tie-resolved differently.
Scores, lambdas and NDCG of later rounds are unaffected and are still compared bit for bit.
"""
import numpy as np


def node_members(tree, X, feature_ids=None):
    """replay a flat pre-order tree on X -> {node: sorted sample indices}"""
    col = {fid: c for c, fid in enumerate(feature_ids if feature_ids is not None else range(1, X.shape[1] + 1))}
    out = {}
    stack = [(0, np.arange(X.shape[0]))]
    while stack:
        n, idx = stack.pop()
        out[n] = idx
        if tree["feature"][n] != -1:
            v = X[idx, col[int(tree["feature"][n])]]
            m = v <= tree["threshold"][n]
            stack.append((int(tree["left"][n]), idx[m]))
            stack.append((int(tree["right"][n]), idx[~m]))
    return out


def assert_equivalent(to, tg, X, ctx="", stats=None, feature_ids=None):
    """to: oracle tree, tg: GPU tree (objects with .trimmed() and .n_nodes). Returns #tie-resolved splits."""
    a, b = to.trimmed(), tg.trimmed()
    assert to.n_nodes == tg.n_nodes, ctx
    n_split = int((a["feature"] != -1).sum())
    identical = (np.array_equal(a["feature"], b["feature"]) and
                 np.array_equal(a["threshold"].view(np.uint32), b["threshold"].view(np.uint32)) and
                 np.array_equal(a["left"], b["left"]) and np.array_equal(a["right"], b["right"]))
    ties = 0
    if identical:
        assert np.array_equal(a["count"], b["count"]), ctx
        assert np.array_equal(a["output"].view(np.uint32), b["output"].view(np.uint32)), (ctx, a["output"], b["output"])
    else:
        ma, mb = node_members(a, X, feature_ids), node_members(b, X, feature_ids)
        stack = [(0, 0)]
        while stack:
            na, nb = stack.pop()
            assert np.array_equal(ma[na], mb[nb]), (ctx, "nodes %d/%d hold different training samples" % (na, nb))
            assert a["count"][na] == b["count"][nb] == len(ma[na]), (ctx, na, nb)
            la, lb = a["feature"][na] == -1, b["feature"][nb] == -1
            assert la == lb, (ctx, "node %d/%d: leaf vs split" % (na, nb))
            if la:
                assert np.float32(a["output"][na]).view(np.uint32) == np.float32(b["output"][nb]).view(np.uint32), \
                    (ctx, na, nb, a["output"][na], b["output"][nb])
                continue
            al, ar, bl, br = int(a["left"][na]), int(a["right"][na]), int(b["left"][nb]), int(b["right"][nb])
            if np.array_equal(ma[al], mb[bl]):
                stack += [(al, bl), (ar, br)]
                if not (a["feature"][na] == b["feature"][nb] and
                        np.float32(a["threshold"][na]).view(np.uint32) == np.float32(b["threshold"][nb]).view(np.uint32)):
                    ties += 1
            elif np.array_equal(ma[al], mb[br]):          # same two sets, sides mirrored (S is symmetric)
                stack += [(al, br), (ar, bl)]
                ties += 1
            else:
                raise AssertionError((ctx, "node %d/%d: (feature %d, thr %r) and (feature %d, thr %r) cut the "
                                      "training samples into different sets" %
                                      (na, nb, a["feature"][na], a["threshold"][na], b["feature"][nb], b["threshold"][nb])))
    if stats is not None:
        stats["splits"] = stats.get("splits", 0) + n_split
        stats["plateau"] = stats.get("plateau", 0) + ties
    return ties
