"""MART and the other train metrics (DCG, MAP, ERR): hand-derived known answers for the oracle, and the C oracle against
the independent Python restatement bit for bit.

Reference behaviour restated: metric/{DCG,AP,ERR}Scorer.java (swapChange / score), learning/tree/MART.java:47-65,
and the pair loop of learning/tree/LambdaMART.java:361-396 whose `j > cutoff && k > cutoff` break is NOT neutral for
MAP (APScorer's k is 0: only pairs with the top-ranked document are visited).
"""
import math

import numpy as np
import pytest

import np_restatement as R
import oracle_ffi as O
from ranklib_amd import synth
from test_oracle_vs_np import compare_round, small_problem


# ---- known answers ------------------------------------------------------------------------------------------------
def test_map_score_and_the_cutoff_quirk():
    # ranked labels 1,0,1: AP = (1/1 + 2/3) / 2
    s = [3.0, 2.0, 1.0]
    lab = [1.0, 0.0, 1.0]
    assert O.query_score("MAP", s, lab, 0) == (1.0 + 2.0 / 3.0) / 2
    # swapChange[0][1]: diff = -1: ((1-1)*0 - 1*1)/1 + (-2 * -1)/2 ... relCount = [1,1,2]
    #   = (-1)/1 + 0 (no k between) + (-(1)*(-1))/2 = -1 + 0.5 = -0.5 ; / rdCount 2 = -0.25
    # With k = 0 only pairs that include position 0 are visited: (0,1) [label 1 > 0]; pair (2,1) is never visited.
    lam, w = O.query_lambdas_metric("MAP", [0.0, 0.0, 0.0], lab, 0)
    assert lam[0] == 0.5 * 0.25 and lam[1] == -0.5 * 0.25 and lam[2] == 0.0
    assert w[0] == 0.25 * 0.25 and w[1] == 0.25 * 0.25 and w[2] == 0.0
    # MAP@5 (cutoff 5) visits every pair of this list: document 2 now takes part
    lam5, _ = O.query_lambdas_metric("MAP", [0.0, 0.0, 0.0], lab, 5)
    assert lam5[2] != 0.0 and lam5[0] == lam[0]


def test_dcg_is_ndcg_without_the_ideal():
    s = [0.3, 0.1, 0.2, 0.0]
    lab = [2.0, 0.0, 1.0, 3.0]
    # ranked: 0(2), 2(1), 1(0), 3(3)
    d = [1.0 / (math.log(i + 2) / math.log(2)) for i in range(4)]
    assert O.query_score("DCG", s, lab, 10) == ((3 * d[0] + 1 * d[1]) + 0 * d[2]) + 7 * d[3]
    assert O.query_score("DCG", s, lab, 2) == 3 * d[0] + 1 * d[1]
    lam_d, w_d = O.query_lambdas_metric("DCG", s, lab, 2)
    lam_n, w_n = O.query_lambdas(s, lab, k=2, ideal=1.0)           # NDCG with ideal 1.0 divides by exactly 1
    assert list(lam_d) == list(lam_n) and list(w_d) == list(w_n)


def test_err_score_by_hand():
    # R = (2^l - 1)/16 ; ERR = sum_i p_i R_i / i with p *= (1 - R)
    lab = [4.0, 0.0, 2.0]
    s = [3.0, 2.0, 1.0]
    R4, R0, R2 = 15 / 16.0, 0.0, 3 / 16.0
    want = 0.0
    p = 1.0
    for i, r in enumerate((R4, R0, R2), 1):
        want += p * r / i
        p *= (1.0 - r)
    assert O.query_score("ERR", s, lab, 10) == want
    assert O.query_score("ERR", s, lab, 1) == R4


def test_mart_first_tree_is_the_label_mean_per_leaf():
    # one feature that separates the labels exactly: residuals = labels, leaves = float mean of their labels
    X = np.array([[0.0], [0.0], [1.0], [1.0], [2.0], [2.0]], np.float32)
    lab = np.array([0, 0, 1, 1, 4, 4], np.float32)
    qoff = np.array([0, 3, 6], np.int32)
    o = O.Oracle(X, lab, qoff, n_trees=1, n_leaves=3, ranker="MART")
    o.init()
    t, _, _, _ = o.round()
    out = sorted(float(v) for v, f in zip(t.trimmed()["output"], t.trimmed()["feature"]) if f == -1)
    assert out == [0.0, 1.0, 4.0]
    assert list(o.lambdas()) == [0.0, 0.0, 1.0, 1.0, 4.0, 4.0]        # label - 0
    assert [float(v) for v in o.scores()] == [float(np.float32(0.1)) * v for v in (0.0, 0.0, 1.0, 1.0, 4.0, 4.0)]


# ---- oracle == independent restatement ----------------------------------------------------------------------------
@pytest.mark.parametrize("metric,k", [("DCG", 10), ("DCG", 3), ("MAP", 0), ("MAP", 4), ("ERR", 10), ("ERR", 2), ("NDCG", 5)])
def test_query_lambdas_and_scores_match(metric, k):
    rng = np.random.RandomState(7)
    scorer = R.SCORERS[metric](k)
    for n in (1, 2, 7, 23, 40):
        scores = np.round(rng.randn(n), 1)                      # ties on purpose
        lab = rng.choice([0, 0, 1, 2, 3, 4], n).astype(np.float32)
        order = R.stable_desc(list(scores))
        ranked = [float(lab[i]) for i in order]
        assert O.query_score(metric, scores, lab, k) == scorer.score(ranked, "q%d" % n)    # (NDCG caches the ideal per qid)
        if metric == "NDCG":
            continue
        ch = scorer.swap_change(ranked, "q")
        lam = [0.0] * n; w = [0.0] * n
        for j in range(n):
            for kk in range(n):
                if j > k and kk > k:
                    break
                if ranked[j] > ranked[kk]:
                    d = abs(ch[j][kk])
                    if d > 0:
                        rho = 1.0 / (1 + R.jexp(scores[order[j]] - scores[order[kk]]))
                        lam[order[j]] += rho * d; lam[order[kk]] -= rho * d
                        w[order[j]] += rho * (1.0 - rho) * d; w[order[kk]] += rho * (1.0 - rho) * d
        lo, wo = O.query_lambdas_metric(metric, scores, lab, k)
        assert list(lo) == lam and list(wo) == w


@pytest.mark.parametrize("ranker,metric,k,seed", [("MART", "NDCG", 10, 0), ("MART", "ERR", 10, 1), ("LAMBDAMART", "MAP", 0, 2),
                                                  ("LAMBDAMART", "DCG", 5, 3), ("LAMBDAMART", "ERR", 10, 4),
                                                  ("LAMBDAMART", "MAP", 3, 5)])
def test_training_rounds_identical(ranker, metric, k, seed):
    X, lab, qoff = small_problem(260, 6, seed)
    o = O.Oracle(X, lab, qoff, n_trees=4, n_leaves=6, k=k, ranker=ranker, metric=metric)
    r = R.LambdaMART(X, lab, qoff, n_trees=4, n_leaves=6, k=k, ranker=ranker, metric=metric)
    o.init(); r.init()
    for _ in range(4):
        compare_round(o, r)
    so, _ = o.finish()
    assert so == r.finish()


def test_err_max_static_changes_err_like_gmax():
    """-gmax g sets ERRScorer.MAX = 2^g (eval/Evaluator.java:241-242): R = (2^label - 1) / MAX in score and swapChange"""
    import numpy as np
    import oracle_ffi as O
    from ranklib_amd import metric as M
    from ranklib_amd.learning import DataPoint, RankList
    lab = np.array([3, 0, 1, 2, 0], np.float32)
    sc = np.array([0.3, 0.9, 0.1, 0.5, 0.2])
    try:
        base = O.query_score("ERR", sc, lab, 10)
        O.set_err_max(8.0)
        M.ERRScorer.MAX = 8.0
        got = O.query_score("ERR", sc, lab, 10)
        assert got != base
        order = np.argsort(-sc, kind="stable")
        rl = RankList([DataPoint("%d qid:1 1:0" % int(lab[i])) for i in order])
        assert got == M.ERRScorer(10).scoreOne(rl)
        l8, w8 = O.query_lambdas_metric("ERR", sc, lab, 10)
        O.set_err_max(16.0)
        l16, w16 = O.query_lambdas_metric("ERR", sc, lab, 10)
        assert not np.array_equal(l8, l16)
    finally:
        O.set_err_max(16.0)
        M.ERRScorer.MAX = 16.0


def test_gain_of_labels_above_30_is_java_int_arithmetic():
    """(1 << i) - 1 with Java ints: shift count mod 32, wrapping subtraction (metric/DCGScorer.java:28-31,137-139) -- host mirror and oracle agree"""
    from ranklib_amd import metric as M
    assert [M.java_pow2m1(i) for i in (0, 1, 4, 30, 31, 32, 33, 63, 64, 1000)] == [0, 1, 15, 2 ** 30 - 1, 2 ** 31 - 1, 0, 1, 2 ** 31 - 1, 0, 255]
    lab = np.array([31, 32, 33, 2, 0, 40], np.float32)
    X = np.arange(12, dtype=np.float32).reshape(6, 2)
    qoff = np.array([0, 6], np.int32)
    o = O.Oracle(X, lab, qoff, n_trees=1, n_leaves=2, metric="DCG", k=10)
    o.init()
    _, tm, _, _ = o.round()
    # the scorer of the host mirror on the same ranking gives the oracle's DCG
    scores = o.scores()
    order = sorted(range(6), key=lambda i: (-scores[i], i))
    want = 0.0
    for pos, i in enumerate(order):
        want += M.gain(int(lab[i])) * M.discount(pos)
    assert np.float32(want) == np.float32(tm)
