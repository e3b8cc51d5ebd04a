"""Tree comparison used by the GPU parity tests.

Everything is compared bit for bit, with ONE documented exception (DESIGN.md "plateau ties"):
for a node whose histogram RankLib derives by subtraction (right children,
learning/tree/FeatureHistogram.java:222-234) the cumulative sums on an EMPTY-BIN PLATEAU are equal in
exact arithmetic but differ by f64 rounding noise in the Java, so Java's arg-max lands on a noise-chosen
threshold inside the plateau.  The GPU accumulates exactly, sees the tie and keeps the lowest threshold.
Both thresholds send every training sample of that node the same way; the check below verifies precisely
that (same feature, same sample partition) and reports how often it happened.
"""
import numpy as np


def node_members(tree, X, feature_ids=None):
    """replay a flat pre-order tree on X -> {node: sample indices}"""
    col = {fid: c for c, fid in enumerate(feature_ids if feature_ids is not None else range(1, X.shape[1] + 1))}
    out = {}

    def rec(n, idx):
        out[n] = idx
        if tree["feature"][n] != -1:
            v = X[idx, col[int(tree["feature"][n])]]
            m = v <= tree["threshold"][n]
            rec(int(tree["left"][n]), idx[m])
            rec(int(tree["right"][n]), idx[~m])

    rec(0, np.arange(X.shape[0]))
    return out


def assert_equivalent(to, tg, X, ctx="", stats=None, feature_ids=None):
    """to: oracle tree, tg: GPU tree (objects with .trimmed() and .n_nodes)"""
    a, b = to.trimmed(), tg.trimmed()
    assert to.n_nodes == tg.n_nodes, ctx
    assert np.array_equal(a["feature"], b["feature"]), (ctx, a["feature"], b["feature"])
    assert np.array_equal(a["left"], b["left"]) and np.array_equal(a["right"], b["right"]), ctx
    assert np.array_equal(a["count"], b["count"]), (ctx, a["count"], b["count"])
    assert np.array_equal(a["output"].view(np.uint32), b["output"].view(np.uint32)), (ctx, a["output"], b["output"])
    same = a["threshold"].view(np.uint32) == b["threshold"].view(np.uint32)
    n_split = int((a["feature"] != -1).sum())
    n_plateau = 0
    if not same.all():
        mem = node_members(a, X, feature_ids)
        col = {fid: c for c, fid in enumerate(feature_ids if feature_ids is not None else range(1, X.shape[1] + 1))}
        for n in np.nonzero(~same)[0]:
            v = X[mem[int(n)], col[int(a["feature"][n])]]
            assert np.array_equal(v <= a["threshold"][n], v <= b["threshold"][n]), \
                (ctx, "thresholds %r / %r at node %d split the training samples differently" %
                 (a["threshold"][n], b["threshold"][n], n))
            # the exact-arithmetic tie keeps the LOWEST threshold of the plateau
            assert b["threshold"][n] <= a["threshold"][n], (ctx, n)
            n_plateau += 1
    if stats is not None:
        stats["splits"] = stats.get("splits", 0) + n_split
        stats["plateau"] = stats.get("plateau", 0) + n_plateau
    return n_plateau
