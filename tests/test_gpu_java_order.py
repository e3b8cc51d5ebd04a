"""-m gpu: RL_FLAG_JAVA_ORDER -- the strict mode whose split gains come from the f64 histogram RankLib itself would hold
(sequential per-bin sums in ascending sample order, sequential prefix, right = parent - left; rl_java_order.inc).

Under the flag the trees must be IDENTICAL to the oracle's: same stored (feature id, threshold) in every split, no mirrored
children, no tie resolved differently (tree_equiv reports 0), and the validation path can be compared with the oracle
unconditionally (learning/tree/FeatureHistogram.java:236-264,302-309 -- VERDICT r01 row A12).
"""
import numpy as np
import pytest

import oracle_ffi as O
from ranklib_amd import _native as N
from ranklib_amd import synth
from tree_equiv import assert_equivalent

pytestmark = pytest.mark.gpu


def make(n_docs, n_feat, kind="ns", seed=0):
    return synth.make_dataset(n_docs, n_feat, kind, seed_offset=seed)


def pair(X, lab, qoff, **kw):
    args = dict(n_trees=kw.get("n_trees", 5), n_leaves=kw.get("n_leaves", 10))
    o = O.Oracle(X, lab, qoff, lr=kw.get("lr", 0.1), n_threshold=kw.get("n_threshold", 256), mls=kw.get("mls", 1),
                 k=kw.get("k", 10), early_stop=kw.get("early_stop", 100), ranker=kw.get("ranker", "LAMBDAMART"),
                 metric=kw.get("metric", "NDCG"), frate=kw.get("frate", 1.0), seed=kw.get("seed", 0), n_threads=kw.get("n_threads", 1), **args)
    g = N.Trainer(learning_rate=kw.get("lr", 0.1), n_threshold=kw.get("n_threshold", 256),
                  min_leaf_support=kw.get("mls", 1), metric_k=kw.get("k", 10), ranker=kw.get("ranker", "LAMBDAMART"),
                  metric=kw.get("metric", "NDCG"), feature_sampling_rate=kw.get("frate", 1.0), seed=kw.get("seed", 0),
                  early_stop_rounds=kw.get("early_stop", 100), flags=N.RL_FLAG_JAVA_ORDER, **args)
    g.set_train(X, lab, qoff)
    return o, g


def same_tree(to, tg, X, ctx):
    a, b = to.trimmed(), tg.trimmed()
    assert assert_equivalent(to, tg, X, ctx) == 0, ctx
    for k in ("feature", "left", "right", "count"):
        assert np.array_equal(a[k], b[k]), (ctx, k)
    assert np.array_equal(a["threshold"].view(np.uint32), b["threshold"].view(np.uint32)), ctx
    assert np.array_equal(a["output"].view(np.uint32), b["output"].view(np.uint32)), ctx
    assert np.array_equal(a["deviance"].view(np.int64), b["deviance"].view(np.int64)), ctx      # Split.deviance: the Java's own f64 sums


def test_root_histogram_equals_the_java_order_sums():
    X, lab, qoff = make(20000, 12, "ns", 3)
    o, g = pair(X, lab, qoff, n_trees=2, n_leaves=4)
    o.init(); g.init()
    for r in range(2):
        o.round(); g.boost_round()
        js = g.array("ROOT_SUM_JAVA")
        nb = g.array("NBINS")
        for f in range(12):
            if nb[f] <= 2:
                continue            # a single distinct value: never scanned
            assert np.array_equal(js[f, :nb[f]].view(np.int64), o.root_sum(f).view(np.int64)), (r, f)


@pytest.mark.parametrize("n_docs,n_feat,kind,leaves,mls,rounds,seed", [
    (3000, 10, "ns", 10, 1, 8, 0),
    (8000, 136, "ns", 10, 1, 6, 1),       # c0 shape
    (12000, 20, "mslr", 31, 1, 5, 2),
    (4000, 8, "ns", 6, 25, 6, 3),
    (2500, 5, "mslr", 31, 1, 4, 4),
    (9000, 24, "mslr", 12, 1, 5, 3),
    (700, 3, "ns", 64, 1, 5, 11),         # deep trees over a handful of documents per node: where the default path ties most
])
def test_trees_identical_to_the_oracle(n_docs, n_feat, kind, leaves, mls, rounds, seed):
    X, lab, qoff = make(n_docs, n_feat, kind, seed)
    o, g = pair(X, lab, qoff, n_trees=rounds, n_leaves=leaves, mls=mls)
    o.init(); g.init()
    for r in range(rounds):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        same_tree(to, tg, X, "round %d trace %s" % (r, o.split_trace()))
        assert np.float32(tmo).view(np.uint32) == np.float32(tmg).view(np.uint32), r
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r
    so, _ = o.finish()
    sg, _ = g.finish()
    assert so == sg
    assert np.array_equal(g.predict(X[:777]).view(np.uint32), o.predict(X[:777]).view(np.uint32))


@pytest.mark.parametrize("ranker,metric,k,frate", [("MART", "NDCG", 10, 1.0), ("LAMBDAMART", "MAP", 0, 1.0), ("LAMBDAMART", "ERR", 10, 1.0),
                                                   ("LAMBDAMART", "DCG", 3, 1.0), ("LAMBDAMART", "NDCG", 10, 0.4)])
def test_other_rankers_metrics_and_feature_sampling(ranker, metric, k, frate):
    X, lab, qoff = make(5000, 14, "mslr", 8)
    o, g = pair(X, lab, qoff, n_trees=4, n_leaves=12, ranker=ranker, metric=metric, k=k, frate=frate, seed=77)
    o.init(); g.init()
    for r in range(4):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        same_tree(to, tg, X, "%s %s round %d" % (ranker, metric, r))
        assert tmo == tmg
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r


def test_oracle_thread_count_does_not_matter():
    """MyThreadPool splits the features over the threads; every (feature, bin) chain stays in one thread, so the Java's sums --
    and this mode's -- are the same for any -thread (utilities/MyThreadPool.java:77-87)."""
    X, lab, qoff = make(6000, 16, "mslr", 2)
    o, g = pair(X, lab, qoff, n_trees=3, n_leaves=16, n_threads=5)
    o.init(); g.init()
    for r in range(3):
        to, _, _, _ = o.round()
        tg, _, _, _ = g.boost_round()
        same_tree(to, tg, X, "round %d" % r)


def test_validation_early_stop_rollback_against_the_oracle():
    """every round's validation metric, the stop flag, the rollback and the final scores against the ORACLE (the default path
    can only be compared with a replay of its own trees once a tie has been resolved differently)"""
    X, lab, qoff = make(4000, 12, "ns", 5)
    Xv, labv, qoffv = make(2500, 12, "ns", 6)
    o, g = pair(X, lab, qoff, n_trees=40, n_leaves=5, early_stop=3)
    o.set_validation(Xv, labv, qoffv); g.set_validation(Xv, labv, qoffv)
    o.init(); g.init()
    for r in range(40):
        to, tmo, vmo, so = o.round()
        tg, tmg, vmg, sg = g.boost_round()
        same_tree(to, tg, X, "round %d" % r)
        assert tmo == tmg and vmo == vmg and so == sg, r
        assert np.array_equal(g.array("VALID_SCORE").view(np.int64), o.valid_scores().view(np.int64)), r
        if sg:
            break
    ts_o, vs_o = o.finish()
    ts_g, vs_g = g.finish()
    assert ts_o == ts_g and vs_o == vs_g and g.num_trees() == o.trees_kept()
    assert g.best_validation()[0] == o.best_valid()[0]


def test_flag_is_rejected_with_sharding():
    X, lab, qoff = make(2000, 6, "ns", 1)
    g = N.Trainer(n_trees=2, n_leaves=4, flags=N.RL_FLAG_JAVA_ORDER)
    g.set_train(X, lab, qoff)
    g.dist_init_callback(0, 1, lambda arr, op: None, lambda src: src.copy())
    with pytest.raises(N.RankLibError):
        g.init()


def test_first_version_kernel_and_wide_threshold_tables(monkeypatch):
    """k_jhist (one wavefront per 64 bins, scalar-load walk) still serves threshold tables beyond k_jhist2's 512 bins: -tc -1 on a
    column with ~1500 distinct values takes it by itself; RLHIP_JHIST_V1 forces it on an ordinary table.  Same trees either way."""
    X, lab, qoff = make(3000, 6, "ns", 17)
    X = X.copy()
    X[:, 1] = np.round(X[:, 1] * 1500) / 1500            # ~1500 distinct values: 1501 bins with -tc -1
    for tc, force in ((-1, False), (256, True)):
        if force:
            monkeypatch.setenv("RLHIP_JHIST_V1", "1")
        o, g = pair(X, lab, qoff, n_trees=3, n_leaves=9, n_threshold=tc)
        o.init(); g.init()
        assert (g.array("NBINS").max() > 512) == (tc == -1)
        for r in range(3):
            to, tmo, _, _ = o.round()
            tg, tmg, _, _ = g.boost_round()
            same_tree(to, tg, X, "tc %d round %d" % (tc, r))
            assert tmo == tmg
