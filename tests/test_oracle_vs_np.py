"""C oracle vs the independent Python restatement, bit for bit, on small problems.

Both are restatements of the same Java (neither can be checked against a JVM
here); agreement of two separately written readings is the pin.
"""
import numpy as np
import pytest

import np_restatement as R
import oracle_ffi as O
from ranklib_amd import synth


def small_problem(n_docs, n_features, seed, kind="ns"):
    X, lab, qoff = synth.make_dataset(n_docs, n_features, kind, seed_offset=seed)
    return X, lab, qoff


def compare_round(o, r):
    to, tmo, vmo, stopo = o.round()
    root, tmr, vmr, stopr = r.round()
    feat, thr, left, right, out = R.flatten_tree(root)
    tr = to.trimmed()
    assert list(tr["feature"]) == feat
    assert [float(v) for v in tr["threshold"]] == thr
    assert list(tr["left"]) == left and list(tr["right"]) == right
    assert [float(v) for v in tr["output"]] == out
    assert o.split_trace() == r.splits_trace
    assert float(tmo) == float(tmr)
    if vmr is not None:
        assert float(vmo) == float(vmr)
    assert stopo == stopr
    assert list(o.scores()) == r.model_scores
    assert list(o.lambdas()) == r.pseudo
    assert list(o.weights()) == r.weights
    return stopo


@pytest.mark.parametrize("n_docs,n_feat,leaves,mls,seed", [(120, 5, 4, 1, 0), (300, 9, 10, 1, 1), (257, 6, 6, 3, 2),
                                                            (600, 4, 7, 2, 3)])
def test_init_and_rounds_identical(n_docs, n_feat, leaves, mls, seed):
    X, lab, qoff = small_problem(n_docs, n_feat, seed)
    o = O.Oracle(X, lab, qoff, n_trees=4, n_leaves=leaves, mls=mls)
    r = R.LambdaMART(X, lab, qoff, n_trees=4, n_leaves=leaves, mls=mls)
    o.init(); r.init()
    for f in range(n_feat):
        assert [float(v) for v in o.thresholds(f)] == [float(v) for v in r.thresholds[f]]
        assert list(o.bins(f)) == r.bins[f]
        assert list(o.root_count(f)) == r.root_hist.count[f]
    for _ in range(4):
        compare_round(o, r)
    so, _ = o.finish()
    assert so == r.finish()
    assert [float(v) for v in o.predict(X[:50])] == [float(v) for v in r.predict(X[:50])]


def test_more_than_256_distinct_values_and_tc():
    X, lab, qoff = small_problem(700, 4, 5)
    for tc in (256, 16, -1):
        o = O.Oracle(X, lab, qoff, n_trees=2, n_leaves=5, n_threshold=tc)
        r = R.LambdaMART(X, lab, qoff, n_trees=2, n_leaves=5, n_threshold=tc)
        o.init(); r.init()
        assert o.n_bins(1) == len(r.thresholds[1])
        if tc == 256:
            assert o.n_bins(1) == 257
        for _ in range(2):
            compare_round(o, r)


def test_validation_early_stop_and_rollback():
    X, lab, qoff = small_problem(300, 6, 7)
    Xv, labv, qoffv = small_problem(200, 6, 8)
    o = O.Oracle(X, lab, qoff, n_trees=30, n_leaves=4, early_stop=2)
    r = R.LambdaMART(X, lab, qoff, n_trees=30, n_leaves=4, early_stop=2)
    o.set_validation(Xv, labv, qoffv); r.set_validation(Xv, labv, qoffv)
    o.init(); r.init()
    rounds = 0
    for m in range(30):
        rounds += 1
        if compare_round(o, r):
            break
    assert list(o.valid_scores()) == r.valid_scores
    so, vo = o.finish()
    sr = r.finish()
    assert so == sr
    assert o.trees_kept() == len(r.ensemble) == r.best_model_on_validation + 1
    assert o.best_valid()[0] == r.best_model_on_validation


def test_unlimited_leaves_and_threads_give_same_trees():
    X, lab, qoff = small_problem(400, 7, 9)
    a = O.Oracle(X, lab, qoff, n_trees=3, n_leaves=-1, mls=20, max_nodes=201)
    r = R.LambdaMART(X, lab, qoff, n_trees=3, n_leaves=-1, mls=20)
    b = O.Oracle(X, lab, qoff, n_trees=3, n_leaves=-1, mls=20, n_threads=3, max_nodes=201)
    a.init(); r.init(); b.init()
    for _ in range(3):
        compare_round(a, r)
        tb, tmb, _, _ = b.round()
        # multi-threaded partition (MyThreadPool.partition) must not change any result
        assert list(b.scores()) == list(a.scores())
        assert b.split_trace() == a.split_trace()


def test_shared_qid_between_train_and_validation():
    X, lab, qoff = small_problem(120, 4, 11)
    Xv, labv, qoffv = small_problem(90, 4, 12)
    Q, Qv = len(qoff) - 1, len(qoffv) - 1
    qk = list(range(Q)); qkv = list(range(Qv))      # validation ids collide with training ids
    o = O.Oracle(X, lab, qoff, n_trees=3, n_leaves=4, qkey=qk)
    r = R.LambdaMART(X, lab, qoff, n_trees=3, n_leaves=4, qids=["q%d" % i for i in qk])
    o.set_validation(Xv, labv, qoffv, qkey=qkv); r.set_validation(Xv, labv, qoffv, qids=["q%d" % i for i in qkv])
    o.init(); r.init()
    for _ in range(3):
        compare_round(o, r)
