"""Known-answer tests that pin the CPU oracle.

The reference's own tests hold no numeric vectors for LambdaMART, and there is
no JVM here, so these answers are derived by hand from the Java's formulas
(citations relative to /root/reference/src/main/java/ciir/umass/edu/).
"""
import math

import numpy as np
import pytest

import oracle_ffi as O
import np_restatement as R

F32 = np.float32


def ulp_diff(a, b):
    ia = np.array([a], np.float64).view(np.int64)[0]
    ib = np.array([b], np.float64).view(np.int64)[0]
    return abs(int(ia) - int(ib))


def test_exp_restatement_close_to_libm_and_identical_across_restatements():
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-40, 40, 4000), rng.uniform(-1e-3, 1e-3, 500), rng.uniform(-745, 709, 500),
                         [0.0, -0.0, 1.0, -1.0, 0.5 * math.log(2), 1.5 * math.log(2), 2.0 ** -29, 710.0, -746.0]])
    L = O.lib()
    for x in xs:
        a = L.ro_exp(float(x))
        assert a == R.jexp(float(x))          # C oracle == Python restatement, bit for bit
        e = math.exp(x) if x < 709.78 else math.inf
        if e not in (0.0, math.inf) and e > 1e-300:
            assert ulp_diff(a, e) <= 1, (x, a, e)
    assert L.ro_exp(0.0) == 1.0
    # fdlibm (== Java StrictMath.exp) is 1 ulp above Math.E here: the well-known StrictMath.exp(1.0) != Math.E
    assert L.ro_exp(1.0) == 2.7182818284590455
    assert L.ro_exp(float("-inf")) == 0.0


def test_discount_and_gain_tables():
    # metric/DCGScorer.java:26: 1/log2(i+2) ; :30: 2^l - 1
    L = O.lib()
    assert L.ro_discount(0) == 1.0
    assert abs(L.ro_discount(1) - 0.6309297535714574) < 1e-16
    assert L.ro_discount(2) == 0.5
    for i in range(50):
        assert L.ro_discount(i) == 1.0 / (math.log(i + 2) / math.log(2))
        assert L.ro_discount(i) == R.discount(i)


def test_three_doc_query_lambda_kat():
    # labels [2,0,1], all scores 0, NDCG@10 -- SURVEY.md 8c worked example
    d = [1.0, 1.0 / (math.log(3) / math.log(2)), 0.5]
    ideal = 3 * d[0] + 1 * d[1] + 0 * d[2]
    assert ideal == 3.6309297535714573
    d01 = abs((d[0] - d[1]) * (3.0 - 0.0) / ideal)
    d02 = abs((d[0] - d[2]) * (3.0 - 1.0) / ideal)
    d12 = abs((d[1] - d[2]) * (0.0 - 1.0) / ideal)
    exp_l = [0.5 * d01 + 0.5 * d02, -(0.5 * d01) - 0.5 * d12, -(0.5 * d02) + 0.5 * d12]
    exp_w = [0.25 * d01 + 0.25 * d02, 0.25 * d01 + 0.25 * d12, 0.25 * d02 + 0.25 * d12]
    lam, w = O.query_lambdas([0.0, 0.0, 0.0], [2, 0, 1], k=10)
    assert list(lam) == exp_l
    assert list(w) == exp_w
    assert np.allclose(lam, [0.2901750904452134, -0.17049909759879334, -0.11967599284642003], rtol=0, atol=1e-16)
    assert np.allclose(w, [0.1450875452226067, 0.08524954879939667, 0.07786777976488334], rtol=0, atol=1e-16)
    assert O.query_ndcg([0.0, 0.0, 0.0], [2, 0, 1]) == (3 * d[0] + 0 * d[1] + 1 * d[2]) / ideal


def test_delta_uses_true_discount_beyond_cutoff():
    # metric/NDCGScorer.java:151-157: i < size, j up to n: discount(j) is the true one for j >= k
    n, k = 6, 2
    labels = [0, 1, 0, 0, 2, 0]
    lam, w = O.query_lambdas([0.0] * n, labels, k=k)
    ideal = 3 * 1.0 + 1 * R.discount(1)
    # pair (pos 4 label 2) vs (pos 0 label 0): min pos 0 < size -> active with discount(4)
    d = abs((R.discount(0) - R.discount(4)) * (0.0 - 3.0) / ideal)
    # pair (4,1): (disc(1)-disc(4))*(1-3)/ideal ; pairs (4,2),(4,3),(4,5): min pos >= size -> 0
    d41 = abs((R.discount(1) - R.discount(4)) * (1.0 - 3.0) / ideal)
    assert lam[4] == 0.5 * d + 0.5 * d41
    # doc 1 (label 1): vs 0: (d0-d1)(0-1); vs 2,3,5: (d1-dj)(1-0); minus pair with 4
    # accumulate in Java order for position r=1: j=0: none; j=1: k=0,2,3,5 ; j=4: (4,1)
    acc = 0.0
    for kk in (0, 2, 3, 5):
        a, b = min(1, kk), max(1, kk)
        acc += 0.5 * abs((R.discount(a) - R.discount(b)) * (R.gain(labels[a]) - R.gain(labels[b])) / ideal)
    acc -= 0.5 * d41
    assert lam[1] == acc


def test_stable_descending_sort_ties_keep_file_order():
    s = [0.5, 1.0, 0.5, 1.0, -0.0, 0.0, 0.5]
    assert list(O.sort_desc(s)) == [1, 3, 0, 2, 6, 4, 5]
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 17, 64, 257):
        v = rng.integers(0, 5, n).astype(np.float64)
        ref = sorted(range(n), key=lambda i: -v[i])
        assert list(O.sort_desc(v)) == ref


def test_float_running_sum_is_java_float_chain():
    # LambdaMART.java:401-408: float s += double  ==  s = (float)((double)s + x)
    x = np.array([0.1] * 10 + [1e-9] * 5 + [16777216.0, 1.0, 1.0])
    s = F32(0)
    for v in x:
        s = F32(float(s) + v)
    assert O.float_chain(x) == s
    assert O.float_chain(np.array([16777216.0, 1.0, 1.0])) == F32(16777216.0)   # 1.0 is lost twice
    assert float(O.float_chain(np.array([0.1] * 10))) != sum([0.1] * 10)


def _col_problem(col, labels=None, n_threshold=256):
    col = np.asarray(col, np.float32)
    N = len(col)
    X = col.reshape(N, 1)
    lab = np.zeros(N, np.float32) if labels is None else np.asarray(labels, np.float32)
    o = O.Oracle(X, lab, [0, N], n_threshold=n_threshold)
    o.init()
    return o


def test_thresholds_distinct_values_branch():
    # LambdaMART.java:135-140
    o = _col_problem([3, 1, 2, 1, 3, 3, 0.5])
    thr = o.thresholds(0)
    assert list(thr) == [0.5, 1.0, 2.0, 3.0, np.float32(3.4028234663852886e38)]
    assert list(o.bins(0)) == [3, 1, 2, 1, 3, 3, 0]
    assert list(o.root_count(0)) == [1, 3, 4, 7, 7]


def test_thresholds_step_branch_float_accumulation():
    # LambdaMART.java:141-149: > nThreshold distinct -> fmin + j*step with FLOAT adds, then MAX_VALUE
    rng = np.random.default_rng(5)
    col = rng.random(1000).astype(np.float32)
    o = _col_problem(col)
    thr = o.thresholds(0)
    assert len(thr) == 257
    fmin, fmax = col.min(), col.max()
    step = F32(abs(F32(fmax - fmin))) / F32(256)
    exp = [fmin]
    for j in range(1, 256):
        exp.append(F32(exp[-1] + step))
    exp.append(np.finfo(np.float32).max)
    assert [float(v) for v in thr] == [float(v) for v in exp]
    b = o.bins(0)
    # bin = smallest t with value <= thr[t]   (FeatureHistogram.java:88-107)
    exp_b = np.searchsorted(np.array(exp, np.float32), col, side="left")
    assert list(b) == list(exp_b)
    assert b.max() == 256                      # H4: the 257th bin is populated
    assert o.root_count(0)[-1] == 1000
    # -tc -1: always the distinct values
    o2 = _col_problem(col, n_threshold=-1)
    assert o2.n_bins(0) == len(np.unique(col)) + 1


def _evaluator_test_data(seed=7):
    # test:eval/EvaluatorTest.java:65-76 writeRandomData: ONE query, 100 x (label 1, f1=1.0, f2=+-1),
    # 100 x (label 0, f1=0.9, f2=+-1), interleaved P,N,P,N...
    rng = np.random.default_rng(seed)
    rows, labels = [], []
    for _ in range(100):
        rows.append([1.0, 1.0 if rng.random() < 0.5 else -1.0]); labels.append(1)
        rows.append([0.9, 1.0 if rng.random() < 0.5 else -1.0]); labels.append(0)
    return np.array(rows, np.float32), np.array(labels, np.float32), [0, 200]


def test_first_tree_on_reference_test_shape():
    X, lab, qoff = _evaluator_test_data()
    o = O.Oracle(X, lab, qoff, n_trees=5, n_leaves=10)
    o.init()
    t, tm, _, _ = o.round()
    tr = t.trimmed()
    # root split: feature id 1 at threshold 0.9 (the only informative cut)
    assert tr["feature"][0] == 1 and tr["threshold"][0] == np.float32(0.9)
    leaves = [i for i in range(t.n_nodes) if tr["feature"][i] == -1]
    # only 2 x 2 distinct feature vectors exist -> at most 4 leaves
    assert len(leaves) <= 4
    # round 0: rho == 0.5 exactly => lambda = +-0.5*sum(delta), w = 0.25*sum(delta): output is exactly +-2
    # for leaves whose docs took part in an active pair, 0 otherwise (s2 == 0)
    for i in leaves:
        assert tr["output"][i] in (np.float32(2.0), np.float32(-2.0), np.float32(0.0))
    assert np.float32(2.0) in tr["output"][leaves] and np.float32(-2.0) in tr["output"][leaves]
    for _ in range(4):
        o.round()
    s = o.scores()
    assert s[lab == 1].min() > s[lab == 0].max()      # behavioural property of EvaluatorTest.testRanker
    ts, _ = o.finish()
    assert ts == 1.0


def test_tiebreak_duplicate_feature_lower_index_wins_and_empty_bins_lowest_t():
    # FeatureHistogram.java:255-260 strict '<' => first maximum wins
    rng = np.random.default_rng(11)
    N = 400
    base = rng.integers(0, 8, N).astype(np.float32) * 2.0      # values 0,2,..,14 ; gaps are empty bins? no: distinct
    noise = rng.random(N).astype(np.float32)
    X = np.stack([noise, base, base.copy(), noise.copy()], axis=1)
    lab = (base >= 8).astype(np.float32) * 2
    qoff = list(range(0, N + 1, 20))
    o = O.Oracle(X, lab, qoff, n_trees=2, n_leaves=4)
    o.init()
    t, _, _, _ = o.round()
    tr = t.trimmed()
    for i in range(t.n_nodes):
        assert tr["feature"][i] in (-1, 1, 2)        # ids 3 (dup of 2) and 4 (dup of 1) never win
    f, tt, S, nn, nl = o.split_trace()[0]
    assert f == 1 and o.thresholds(1)[tt] == np.float32(6.0)


def test_tiebreak_empty_bins_lowest_threshold_index_wins():
    # One feature with values 0..9.  Query A holds values {0,1,2,3,8,9} with the relevant docs at {8,9};
    # bins 4..7 are empty, so cumulative (sum,count) at t=3..7 are identical => S ties exactly and the
    # strict '<' (FeatureHistogram.java:255) keeps the lowest t: threshold 3.0, not 7.0.
    vals = [0, 1, 2, 3, 8, 9, 4, 5, 6, 7]
    X = np.array(vals, np.float32).reshape(-1, 1)
    lab = np.array([0, 0, 0, 0, 1, 1, 0, 0, 0, 0], np.float32)
    o = O.Oracle(X, lab, [0, 6, 10], n_trees=1, n_leaves=2)
    o.init()
    t, _, _, _ = o.round()
    tr = t.trimmed()
    # docs 6..9 (query B) have no relevant doc -> lambda 0; best cut separates {8,9}: candidates t=3 and t=7
    # differ (docs with values 4..7 exist in the node with lambda 0 -> counts differ) so use S directly:
    f, tt, S, nn, nl = o.split_trace()[0]
    lam = o.lambdas()
    order = np.argsort(np.array(vals))
    best, best_t = -1.0, -1
    for cut in range(10):
        left = [k for k in range(10) if vals[k] <= cut]
        right = [k for k in range(10) if vals[k] > cut]
        if not left or not right:
            continue
        sl = 0.0
        for b in range(cut + 1):                       # per-bin sums in ascending doc order, then prefix
            sb = 0.0
            for k in range(10):
                if vals[k] == b:
                    sb += lam[k]
            sl = sb if b == 0 else sl + sb
        sr = float(np.add.reduce([0.0])) + (sum_seq(lam) - sl)
        Sc = sl * sl / len(left) + sr * sr / len(right)
        if best < Sc:
            best, best_t = Sc, cut
    assert tt == best_t and S == best


def sum_seq(x):
    s = 0.0
    for v in x:
        s += v
    return s


def test_growth_queue_inserts_before_equal_deviance():
    # RegressionTree.java:147-157: new node goes BEFORE existing nodes of equal deviance
    q = []
    a, b, c = R.Split([0], None, 1.0), R.Split([0], None, 1.0), R.Split([0], None, 2.0)
    R.LambdaMART._insert(q, a); R.LambdaMART._insert(q, b); R.LambdaMART._insert(q, c)
    assert q == [c, b, a]


def test_min_leaf_support_and_single_leaf_tree():
    # all labels equal -> no active pair -> all lambdas 0 -> S = 0 for every candidate; first candidate with
    # both sides >= mls wins (S=0 > -1).  Deviance of children is exactly 0 -> never split again.
    X = np.arange(12, dtype=np.float32).reshape(12, 1)
    lab = np.ones(12, np.float32)
    o = O.Oracle(X, lab, [0, 6, 12], n_trees=1, n_leaves=10, mls=3)
    o.init()
    t, tm, _, _ = o.round()
    tr = t.trimmed()
    assert t.n_nodes == 3 and tr["feature"][0] == 1
    assert tr["threshold"][0] == np.float32(2.0)              # countLeft = 3 is the first t with cl >= mls
    assert list(tr["output"][1:3]) == [0.0, 0.0]              # s2 == 0 -> 0  (LambdaMART.java:409-411)
    assert tm == np.float32(1.0)


def test_ideal_cache_quirk_shared_qid():
    # metric/NDCGScorer.java:114-122,134-143: ideal DCG is cached per qid by score(); swapChange only reads.
    # Two lists with the SAME qid: round 0 lambdas use their own ideal, later rounds the first list's ideal.
    X = np.array([[1.0], [0.0], [0.5], [1.0], [0.2], [0.1]], np.float32)
    lab = np.array([2, 0, 1, 1, 0, 0], np.float32)
    qoff = [0, 3, 6]
    shared = O.Oracle(X, lab, qoff, n_trees=3, n_leaves=3, qkey=[7, 7])
    own = O.Oracle(X, lab, qoff, n_trees=3, n_leaves=3, qkey=[7, 8])
    for o in (shared, own):
        o.init()
        o.compute_lambdas()
    assert np.array_equal(shared.lambdas(), own.lambdas())      # round 0: cache still empty
    _, tm_s, _, _ = shared.round()
    _, tm_o, _, _ = own.round()
    shared.compute_lambdas(); own.compute_lambdas()
    ls, lo = shared.lambdas(), own.lambdas()
    assert np.array_equal(ls[:3], lo[:3])
    ideal1 = 3.0 + R.discount(1)
    ideal2 = 1.0
    assert np.allclose(ls[3:] * ideal1, lo[3:] * ideal2, rtol=1e-14, atol=0)
    assert not np.array_equal(ls[3:], lo[3:])


def test_round_metric_is_float_accumulated():
    # LambdaMART.java:469-483: float s; s += double; s / Q  (float)
    X, lab, qoff = _evaluator_test_data(3)
    X = np.concatenate([X, X[:50]]); lab = np.concatenate([lab, lab[:50]]); qoff = [0, 70, 200, 250]
    o = O.Oracle(X, lab, qoff, n_trees=1, n_leaves=3)
    o.init()
    _, tm, _, _ = o.round()
    sc = o.scores()
    s = F32(0)
    for q in range(3):
        s = F32(float(s) + O.query_ndcg(sc[qoff[q]:qoff[q + 1]], lab[qoff[q]:qoff[q + 1]]))
    assert tm == F32(s / F32(3))
