"""Committed fixtures (tests/golden/*.npz, generated from the CPU oracle by tests/golden/make_fixtures.py).

CPU: the oracle still reproduces them bit for bit (guards the checker itself).
GPU: the HIP path reproduces them (trees up to the documented exact-arithmetic ties), with no oracle and no
reference present at run time."""
import os

import numpy as np
import pytest

import oracle_ffi as O
from tree_equiv import assert_equivalent

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["small_ns", "small_mslr_k3", "valid_estop", "mart_ndcg", "lmart_map", "lmart_err"]


class T:        # adapter for tree_equiv
    def __init__(self, d):
        self.d, self.n_nodes = d, len(d["feature"])

    def trimmed(self):
        return self.d


def load(name):
    z = np.load(os.path.join(HERE, name + ".npz"), allow_pickle=True)
    p = {k: v for k, v in z["params"]}
    return z, p


def fixture_tree(z, r):
    return T({k: z["tree%d_%s" % (r, k)] for k in ("feature", "threshold", "left", "right", "output", "count")})


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_fixture(name):
    z, p = load(name)
    o = O.Oracle(z["X"], z["labels"], z["qoff"], **p)
    if "Xv" in z:
        o.set_validation(z["Xv"], z["labels_v"], z["qoff_v"])
    o.init()
    for f in range(z["X"].shape[1]):
        assert np.array_equal(o.thresholds(f).view(np.uint32), z["thr%d" % f].view(np.uint32))
        assert np.array_equal(o.bins(f), z["bins%d" % f])
    for r in range(int(z["rounds"])):
        t, tm, vm, stop = o.round()
        tr = t.trimmed()
        for k in ("feature", "threshold", "left", "right", "output", "count"):
            assert np.array_equal(tr[k], z["tree%d_%s" % (r, k)]), (r, k)
        if r < 3:
            assert np.array_equal(o.lambdas().view(np.int64), z["lambda%d" % r].view(np.int64))
        assert tm == z["train_metric"][r]
    assert np.array_equal(o.scores().view(np.int64), z["scores"].view(np.int64))
    ts, vs = o.finish()
    assert ts == float(z["final_train"]) and o.trees_kept() == int(z["trees_kept"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_path_reproduces_fixture(name):
    from ranklib_amd import _native as N
    z, p = load(name)
    g = N.Trainer(n_trees=p["n_trees"], n_leaves=p["n_leaves"], learning_rate=p["lr"], n_threshold=p["n_threshold"],
                  min_leaf_support=p["mls"], metric_k=p["k"], early_stop_rounds=p.get("early_stop", 100),
                  metric=p.get("metric", "NDCG"), ranker=p.get("ranker", "LAMBDAMART"))
    X = z["X"]
    g.set_train(X, z["labels"], z["qoff"])
    if "Xv" in z:
        g.set_validation(z["Xv"], z["labels_v"], z["qoff_v"])
    g.init()
    thr, bins, nb = g.array("THRESHOLDS"), g.array("BINS"), g.array("NBINS")
    for f in range(X.shape[1]):
        assert nb[f] == z["nbins"][f]
        assert np.array_equal(thr[f, :nb[f]].view(np.uint32), z["thr%d" % f].view(np.uint32))
        assert np.array_equal(bins[f], z["bins%d" % f])
    ties = 0
    nr = int(z["rounds"])
    for r in range(nr):
        t, tm, vm, stop = g.boost_round()
        ties += assert_equivalent(fixture_tree(z, r), t, X, "%s round %d" % (name, r))
        if r < 3:
            assert np.array_equal(g.array("LAMBDA").view(np.int64), z["lambda%d" % r].view(np.int64))
            assert np.array_equal(g.array("WEIGHT").view(np.int64), z["weight%d" % r].view(np.int64))
        assert tm == z["train_metric"][r]
        if "Xv" in z and ties == 0:       # validation rows are unseen data: only comparable while no tie was resolved differently
            assert vm == z["valid_metric"][r]
            assert stop == (r == nr - 1 and nr < p["n_trees"])
    assert np.array_equal(g.array("SCORE").view(np.int64), z["scores"].view(np.int64))
    ts, vs = g.finish()
    if ties == 0:
        assert g.num_trees() == int(z["trees_kept"])
        assert ts == float(z["final_train"])
        assert np.array_equal(g.predict(X[:64]).view(np.uint32), z["predict_head"].view(np.uint32))
