"""Host mirror of the RankLib surface.  The CPU tests cover parsing / flags / factories; the GPU test restates the
reference's own LambdaMART test (test:eval/EvaluatorTest.java:186-195 -> :207-260): write writeRandomData's
file, train through the command line, save, load, rank to an Indri run file, check the behavioural property."""
import os
import random

import numpy as np
import pytest

from ranklib_amd import evaluator, features, learning, metric
from ranklib_amd._native import RankLibError


_STATICS = [(learning.LambdaMART, ("nTrees", "learningRate", "nThreshold", "nRoundToStopEarly", "nTreeLeaves", "minLeafSupport")),
            (learning.RFRanker, ("nBag", "subSamplingRate", "featureSamplingRate", "rType", "nTrees", "nTreeLeaves", "learningRate",
                                 "nThreshold", "minLeafSupport", "seed")),
            (learning.FeatureHistogram, ("samplingRate", "seed")), (learning.DataPoint, ("missingZero",))]


@pytest.fixture(autouse=True)
def _restore_process_global_parameters():
    """the hyper-parameters are process-global statics, as in the Java (and RFRanker.init never restores what it overwrites)"""
    saved = [(c, n, getattr(c, n)) for c, names in _STATICS for n in names]
    yield
    for c, n, v in saved:
        setattr(c, n, v)


def write_random_data(path, seed=0):
    # test:eval/EvaluatorTest.java:65-76: ONE query, 100 x (1 qid:x 1:1.0 2:+-1 # P<i>), 100 x (0 qid:x 1:0.9 2:+-1 # N<i>)
    rnd = random.Random(seed)
    with open(path, "w") as f:
        for i in range(100):
            w1 = 1 if rnd.random() < 0.5 else -1
            w2 = 1 if rnd.random() < 0.5 else -1
            f.write("1 qid:x 1:1.0 2:%d # P%d\n" % (w1, i))
            f.write("0 qid:x 1:0.9 2:%d # N%d\n" % (w2, i))


def test_letor_parsing_and_datapoint_rules(tmp_path):
    p = tmp_path / "a.txt"
    p.write_text("# comment\n2 qid:10 1:0.5 3:1e-3 # doc A\n0 qid:10 2:7\n\n1 qid:11 1:1 2:2 3:3 #x\n")
    rls = features.FeatureManager.readInput(str(p))
    assert [rl.size() for rl in rls] == [2, 1] and rls[0].getID() == "10"
    dp = rls[0].get(0)
    assert dp.getLabel() == 2.0 and dp.getDescription() == "# doc A"
    assert dp.getFeatureValue(1) == np.float32(0.5) and dp.getFeatureValue(2) == 0 and dp.getFeatureValue(3) == np.float32(1e-3)
    with pytest.raises(RankLibError):
        rls[0].get(1).getFeatureValue(3)              # beyond the row's last feature (DenseDataPoint.java:22-27)
    learning.DataPoint.missingZero = True
    try:
        assert rls[0].get(1).getFeatureValue(3) == 0
    finally:
        learning.DataPoint.missingZero = False
    assert features.FeatureManager.getFeatureFromSampleVector(rls) == [1, 2, 3]
    with pytest.raises(RankLibError):
        learning.flatten(rls, [1, 3])                 # the reference dies the same way in init() without -missingZero
    learning.DataPoint.missingZero = True
    try:
        X, lab, qoff, qkey = learning.flatten(rls, [1, 3])
    finally:
        learning.DataPoint.missingZero = False
    assert X.tolist() == [[0.5, float(np.float32(1e-3))], [0.0, 0.0], [1.0, 3.0]]
    assert list(qoff) == [0, 2, 3] and list(lab) == [2, 0, 1] and list(qkey) == [0, 1]
    with pytest.raises(RankLibError):
        learning.DataPoint("-1 qid:1 1:2")             # negative label (DataPoint.java:71-73)
    with pytest.raises(RankLibError):
        learning.DataPoint("1 qid:1 0:2")              # feature ids start at 1 (:80-82)


def test_metric_factory_and_ndcg_scorer():
    mf = metric.MetricScorerFactory()
    s = mf.createScorer("ndcg@5")
    assert s.name() == "NDCG@5" and s.getK() == 5
    assert mf.createScorer("NDCG").getK() == 10
    m = mf.createScorer("map")
    assert m.name() == "MAP" and m.getK() == 0                      # metric/APScorer.java:37-39
    assert mf.createScorer("MAP@7").getK() == 7
    assert mf.createScorer("ERR").name() == "ERR@10" and mf.createScorer("dcg@3").name() == "DCG@3"
    assert mf.createScorer("P@5").name() == "P@5" and mf.createScorer("RR").getK() == 0 and mf.createScorer("best").name() == "Best@10"
    with pytest.raises(RankLibError):
        mf.createScorer("nope")
    rl = learning.RankList([learning.DataPoint("%d qid:q 1:0" % l) for l in (2, 0, 1)])
    d = [1.0, 1.0 / (np.log(3) / np.log(2)), 0.5]
    assert mf.createScorer("NDCG@10").score(rl) == (3 * d[0] + 0 * d[1] + 1 * d[2]) / 3.6309297535714573
    assert mf.createScorer("DCG@2").score(rl) == 3 * d[0] + 0 * d[1]
    assert mf.createScorer("MAP").score(rl) == (1.0 + 2.0 / 3.0) / 2
    assert mf.createScorer("P@2").score(rl) == 0.5 and mf.createScorer("RR").score(rl) == 0.0 and mf.createScorer("RR@3").score(rl) == 1.0
    assert mf.createScorer("Best@2").score(rl) == 2.0
    R = [3 / 16.0, 0.0, 1 / 16.0]
    assert mf.createScorer("ERR@10").score(rl) == R[0] / 1 + (1 - R[0]) * R[1] / 2 + (1 - R[0]) * (1 - R[1]) * R[2] / 3


def test_cli_flag_quirks():
    with pytest.raises(RankLibError) as e:
        evaluator.main(["-train", "x", "-silent"])      # documented but unparsed in the reference (Evaluator.java:122,369-371)
    assert "Unknown command-line parameter" in str(e.value)
    with pytest.raises(RankLibError):
        evaluator.main(["-train", "x", "-ranker", "4"])
    assert learning.LambdaMART.nTrees == 1000 and learning.LambdaMART.nTreeLeaves == 10 and learning.LambdaMART.nThreshold == 256
    assert learning.java_round(0.123449, 4) == 0.1234 and learning.java_round(0.12345, 4) == 0.1235


@pytest.mark.gpu
def test_reference_lambdamart_behaviour_through_the_cli(tmp_path):
    data, model, run, sc = (str(tmp_path / n) for n in ("data.txt", "model.txt", "run.txt", "scores.txt"))
    write_random_data(data)
    saved = (learning.LambdaMART.nTrees, learning.LambdaMART.nTreeLeaves)
    try:
        evaluator.main(["-train", data, "-metric2t", "NDCG@10", "-ranker", "6", "-tree", "30", "-save", model])
        evaluator.main(["-rank", data, "-load", model, "-indri", run])
        evaluator.main(["-rank", data, "-load", model, "-score", sc])
    finally:
        learning.LambdaMART.nTrees, learning.LambdaMART.nTreeLeaves = saved
    text = open(model).read()
    assert text.startswith("## LambdaMART\n## No. of trees = 30\n## No. of leaves = 10\n")
    # test:eval/EvaluatorTest.java:238-255
    best_p, best_n = 10 ** 9, 10 ** 9
    for line in open(run):
        row = line.split()
        assert row[1] == "Q0" and row[5] == "indri"
        score = float(row[4])
        assert np.isfinite(score)
        rank = int(row[3])
        if row[2].startswith("P"):
            best_p = min(best_p, rank)
        else:
            best_n = min(best_n, rank)
    assert best_p == 1 and best_p < best_n
    rows = [l.split("\t") for l in open(sc)]
    assert len(rows) == 200 and rows[0][0] == "x" and rows[5][1] == "5"


@pytest.mark.gpu
@pytest.mark.parametrize("rnum,header", [(6, "## LambdaMART\n## No. of trees = 1000\n"), (0, "## MART\n## No. of trees = 1000\n"),
                                         (8, "## Random Forests\n## No. of bags = 10\n## Sub-sampling = 1.0\n## Feature-sampling = 1.0\n"
                                             "## No. of trees = 1\n## No. of leaves = 100\n## No. of threshold candidates = 256\n"
                                             "## Learning rate = 0.1\n\n<ensemble>\n")])
def test_reference_test_flow_verbatim(tmp_path, rnum, header):
    """test:eval/EvaluatorTest.java:128-137 (testMART), :186-195 (testLambdaMART), :94-103 (testRF) -> testRanker :207-260, flag for flag:
    `-metric2t map`, the other rankers' parameters passed along, default 1000 trees."""
    data, model, run = (str(tmp_path / n) for n in ("data.txt", "model.txt", "run.txt"))
    write_random_data(data)
    evaluator.main(["-train", data, "-metric2t", "map", "-ranker", str(rnum), "-frate", "1.0", "-bag", "10", "-round", "10",
                    "-epoch", "10", "-save", model])
    evaluator.main(["-rank", data, "-load", model, "-indri", run])
    text = open(model).read()
    assert text.startswith(header)
    if rnum == 8:
        assert text.count("<ensemble>") == 10 and text.count("</ensemble>\n\n") == 10
    p_rank = n_rank = 2 ** 31 - 1
    for line in open(run):
        row = line.split()
        assert row[1] == "Q0"
        rank, score = int(row[3]), float(row[4])
        assert np.isfinite(score) and rank > 0
        if row[2].startswith("P"):
            p_rank = min(rank, p_rank)
        else:
            n_rank = min(rank, n_rank)
        assert p_rank < n_rank and p_rank == 1


def _lists(n):
    return [learning.RankList([learning.DataPoint("%d qid:q%d 1:%d" % (i % 3, i, i))]) for i in range(n)]


def test_split_helpers_follow_feature_manager():
    s = _lists(11)
    tr, te = features.FeatureManager.prepareSplit(s, 0.7)               # (int)(11 * 0.7) = 7, file order kept
    assert [r.getID() for r in tr] == ["q%d" % i for i in range(7)] and [r.getID() for r in te] == ["q7", "q8", "q9", "q10"]
    trains, valis, tests = features.FeatureManager.prepareCV(s, 3, 0.75)   # folds of 3, the last takes the remainder (5)
    assert [[r.getID() for r in t] for t in tests] == [["q0", "q1", "q2"], ["q3", "q4", "q5"], ["q6", "q7", "q8", "q9", "q10"]]
    # fold 0: 8 training lists, (int)(8 * 0.25) = 2 go to validation, taken from the END, back to front
    assert [r.getID() for r in valis[0]] == ["q10", "q9"] and [r.getID() for r in trains[0]] == ["q3", "q4", "q5", "q6", "q7", "q8"]
    trains, valis, tests = features.FeatureManager.prepareCV(s, 2)
    assert valis == [] and [len(t) for t in tests] == [5, 6] and [len(t) for t in trains] == [6, 5]


@pytest.mark.gpu
def test_cli_tvs_tts_kcv_flows(tmp_path):
    data = str(tmp_path / "d.txt")
    rng = np.random.RandomState(3)
    with open(data, "w") as f:
        for q in range(24):
            for d in range(8):
                x = rng.rand(3)
                f.write("%d qid:%d 1:%f 2:%f 3:%f # d%d_%d\n" % (int(x[0] * 3), q, x[0], x[1], x[2], q, d))
    saved = (learning.LambdaMART.nTrees, learning.LambdaMART.nTreeLeaves)
    try:
        evaluator.main(["-train", data, "-ranker", "6", "-metric2t", "NDCG@5", "-tree", "8", "-leaf", "4", "-tvs", "0.75",
                        "-save", str(tmp_path / "m1.txt")])
        evaluator.main(["-train", data, "-ranker", "6", "-metric2t", "NDCG@5", "-tree", "8", "-leaf", "4", "-tts", "0.5"])
        evaluator.main(["-train", data, "-ranker", "0", "-metric2t", "NDCG@5", "-tree", "5", "-leaf", "4", "-kcv", "3", "-tvs", "0.8",
                        "-kcvmd", str(tmp_path / "cv"), "-kcvmn", "m"])
    finally:
        learning.LambdaMART.nTrees, learning.LambdaMART.nTreeLeaves = saved
    assert open(tmp_path / "m1.txt").read().startswith("## LambdaMART\n## No. of trees = 8\n")
    assert sorted(p.name for p in (tmp_path / "cv").iterdir()) == ["f1.m", "f2.m", "f3.m"]
    assert open(tmp_path / "cv" / "f2.m").read().startswith("## MART\n")


def test_sampler_and_float_strings():
    pool = list(range(20))
    sp = learning.Sampler(11)
    bag = sp.doSampling(pool, 0.75, True)                  # (int)(0.75f * 20) = 15 draws WITH replacement (Sampler.java:31-45)
    assert len(bag) == 15 and set(bag) | set(sp.getRemains()) == set(pool) and not (set(bag) & set(sp.getRemains()))
    assert learning.Sampler(11).doSampling(pool, 0.75, True) == bag and learning.Sampler(12).doSampling(pool, 0.75, True) != bag
    sp = learning.Sampler(5)
    bag = sp.doSampling(pool, 0.5, False)                  # without replacement (:46-59)
    assert len(bag) == 10 == len(set(bag)) and sorted(bag + sp.getRemains()) == pool
    for v, txt in ((0.3, "0.3"), (1.0, "1.0"), (0.1, "0.1"), (1e-4, "1.0E-4"), (1.5e7, "1.5E7"), (0.001, "0.001"), (0.0, "0.0")):
        assert learning.java_float_str(v) == txt             # Float.toString in the model header (RFRanker.java:131-143)
    assert learning.RankerFactory().createRanker("RANDOM_FOREST").name() == "Random Forests"
    assert learning.RFRanker.rType is learning.RankerType.MART and learning.RFRanker.nBag == 300


def test_cli_sets_both_sets_of_statics_like_the_reference():
    """eval/Evaluator.java:326-352: -tree/-leaf/-shrinkage/-mls reach LambdaMART AND RFRanker, -tc only LambdaMART"""
    with pytest.raises(RankLibError):       # the unknown flag stops the run after the earlier flags were applied
        evaluator.main(["-tree", "7", "-leaf", "5", "-shrinkage", "0.25", "-mls", "3", "-tc", "11", "-bag", "4", "-srate", "0.5",
                        "-frate", "0.2", "-rtype", "6", "-seed", "9", "-nosuchflag"])
    assert (learning.LambdaMART.nTrees, learning.RFRanker.nTrees) == (7, 7) and (learning.LambdaMART.nTreeLeaves, learning.RFRanker.nTreeLeaves) == (5, 5)
    assert learning.RFRanker.learningRate == 0.25 and learning.RFRanker.minLeafSupport == 3
    assert learning.LambdaMART.nThreshold == 11 and learning.RFRanker.nThreshold == 256
    assert (learning.RFRanker.nBag, learning.RFRanker.subSamplingRate, learning.RFRanker.featureSamplingRate) == (4, 0.5, 0.2)
    assert learning.RFRanker.rType is learning.RankerType.LAMBDAMART and learning.RFRanker.seed == 9
    with pytest.raises(RankLibError) as e:
        evaluator.main(["-rtype", "4"])
    assert "cannot be bagged" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("rtype,frate,srate", [("MART", 0.5, 1.0), ("LAMBDAMART", 0.4, 0.6)])
def test_random_forests_bags_match_the_oracle(tmp_path, rtype, frate, srate):
    """learning/tree/RFRanker.java:72-115: every bag = Sampler draw of the lists WITH replacement, trained with feature sampling;
    eval = mean of the bags' Ensemble.eval.  Each bag is re-trained by the oracle on the same draw and the same seeds."""
    import oracle_ffi as O
    from tree_equiv import assert_equivalent
    rng = np.random.RandomState(5)
    lines = []
    for q in range(60):
        for d in range(rng.randint(3, 12)):
            x = rng.rand(6)
            lines.append("%d qid:%d %s # d%d_%d" % (int(3 * x[0] * x[1] + x[2]), q, " ".join("%d:%f" % (j + 1, x[j]) for j in range(6)), q, d))
    data = tmp_path / "d.txt"
    data.write_text("\n".join(lines) + "\n")
    samples = features.FeatureManager.readInput(str(data))
    feats = features.FeatureManager.getFeatureFromSampleVector(samples)
    R = learning.RFRanker
    R.nBag, R.featureSamplingRate, R.subSamplingRate, R.rType, R.nTrees, R.nTreeLeaves, R.seed = 3, frate, srate, learning.RankerType[rtype], 2, 6, 77
    scorer = metric.MetricScorerFactory().createScorer("NDCG@10")
    rf = learning.RankerTrainer().train(learning.RankerType.RANDOM_FOREST, samples, feats, scorer)
    assert rf.name() == "Random Forests" and len(rf.getEnsembles()) == 3
    allrows = learning.RankList([dp for rl in samples for dp in rl.rl])
    got = np.array(rf.evalList(allrows))
    Xall, _, _, _ = learning.flatten(samples, feats)
    want = np.zeros(len(Xall))
    for i in range(3):
        bs = R.bag_seed(77, i)
        bag = learning.Sampler(bs).doSampling(samples, srate, True)
        assert len(bag) == int(np.float32(srate) * np.float32(len(samples)))
        X, lab, qoff, qkey = learning.flatten(bag, feats)
        o = O.Oracle(X, lab, qoff, n_trees=2, n_leaves=6, ranker=rtype, frate=frate, seed=bs, early_stop=-1, feature_ids=feats, qkey=qkey)
        o.init()
        for _ in range(2):
            o.round()
        o.finish()
        want += o.predict(Xall).astype(np.float64)            # (float) Ensemble.eval, widened  (:111-113)
    want /= 3
    assert np.array_equal(got, want)
    # save / load round trip through the factory: same scores, same text
    path = str(tmp_path / "rf.txt")
    rf.save(path)
    back = learning.RankerFactory().loadRankerFromFile(path)
    assert back.name() == "Random Forests" and np.array_equal(np.array(back.evalList(allrows)), got)
    assert back.toString() == rf.toString() and open(path).read().startswith("## Random Forests\n## No. of bags = 3\n")


def _same_lists(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x.getID() == y.getID() and len(x) == len(y) and x.getFeatureCount() == y.getFeatureCount()
        for p, q in zip(x.rl, y.rl):
            assert p.label == q.label and p.id == q.id and p.description == q.description
            assert np.array_equal(p.fVals.view(np.uint32), q.fVals.view(np.uint32))


def test_native_letor_reader_equals_the_python_reader(tmp_path):
    """rl_letor_* (host code in librlhip.so) parses the plain grammar on all host threads and hands every other line to
    DataPoint's own parser: same lists, same values bit for bit, same errors (learning/DataPoint.java:58-110,
    features/FeatureManager.java:199-235)."""
    import gzip
    rng = np.random.RandomState(9)
    lines = ["# a comment line", "", "   \t  "]
    for q in range(40):
        for d in range(rng.randint(1, 9)):
            feats = sorted(rng.choice(np.arange(1, 30), rng.randint(1, 12), replace=False))
            toks = " ".join("%d:%s" % (f, rng.choice(["%.6f" % rng.randn(), "%e" % rng.rand(), "%d" % rng.randint(-5, 5), "+1.5", "007", ".25", "3."]))
                            for f in feats)
            sep = rng.choice([" ", "\t", "  "])
            desc = rng.choice(["", " # doc %d_%d" % (q, d), "#x # y:1", "   #"])
            lines.append(("%d%sqid:%s%s%s%s" % (rng.randint(0, 5), sep, rng.choice(["q%d" % q, "a:b:%d" % q]), sep, toks, desc)) + rng.choice(["", " ", "\r"]))
    lines += ["2 qid:zz 1:nan 2:1e400 3:-Infinity 4:1E2 # odd floats go through the slow path", "1 qid:zz 5:1.5",
              "0 qid:zz"]                                     # a line without features
    f1 = tmp_path / "a.txt"
    f1.write_text("\n".join(lines) + "\n")
    with gzip.open(tmp_path / "a.txt.gz", "wt") as g:
        g.write("\n".join(lines) + "\n")
    ref = None
    for native, path in ((False, f1), (True, f1), (True, tmp_path / "a.txt.gz")):
        features.FeatureManager.native = native
        try:
            got = features.FeatureManager.readInput(str(path))
        finally:
            features.FeatureManager.native = True
        if ref is None:
            ref = got
        else:
            _same_lists(ref, got)
    assert sum(len(r) for r in ref) == len([ln for ln in lines if ln.strip() and not ln.strip().startswith("#")])
    # malformed lines raise what the Python reader raises, whichever line comes first
    for bad, msg in (("-1 qid:1 1:0.5", "Relevance label cannot be negative"), ("1 qid:1 0:0.5", "less than or equal to zero"),
                     ("1 qid:1 abc", "Error in DataPoint::parse()"), ("x qid:1 1:1", "Error in DataPoint::parse()")):
        fb = tmp_path / "bad.txt"
        fb.write_text("1 qid:0 1:1.0\n" + bad + "\n2 qid:1 1:2.0\n")
        errs = []
        for native in (False, True):
            features.FeatureManager.native = native
            try:
                with pytest.raises(RankLibError) as e:
                    features.FeatureManager.readInput(str(fb))
                errs.append(str(e.value))
            finally:
                features.FeatureManager.native = True
        assert errs[0] == errs[1] and msg in errs[0], errs
    # mustHaveRelDoc drops the lists without a relevant document on both paths
    f2 = tmp_path / "rel.txt"
    f2.write_text("0 qid:a 1:1\n0 qid:a 1:2\n1 qid:b 1:3\n0 qid:c 1:4\n")
    features.FeatureManager.native = False
    r0 = features.FeatureManager.readInput(str(f2), True)
    features.FeatureManager.native = True
    r1 = features.FeatureManager.readInput(str(f2), True)
    _same_lists(r0, r1)
    assert [r.getID() for r in r1] == ["b"]


def test_java_double_str_follows_double_toString():
    """Double.toString: decimal for 1e-3 <= |v| < 1e7, d.dE-n otherwise (score / indri files, the log table)"""
    from ranklib_amd.learning import java_double_str as j
    assert [j(v) for v in (5e-4, 1e-5, 0.001, 0.4000000059604645, 12345678.0, 9999999.0, 1e7, 123.0, -0.0, 0.0, 1.5e-10, 3.0e22, -2.5e-7)] == \
        ["5.0E-4", "1.0E-5", "0.001", "0.4000000059604645", "1.2345678E7", "9999999.0", "1.0E7", "123.0", "-0.0", "0.0", "1.5E-10", "3.0E22", "-2.5E-7"]
    assert j(float("nan")) == "NaN" and j(float("inf")) == "Infinity" and j(float("-inf")) == "-Infinity"


def test_binary_cache_of_a_parsed_letor_file(tmp_path):
    """FeatureManager.cache: `<file>.rlcache.npz` gives back the same ranked lists as the text (rows, labels, qids, descriptions),
    is ignored once the text is newer, and is never written for files with irregular lines"""
    import os
    import time
    from ranklib_amd.features import FeatureManager
    p = str(tmp_path / "d.txt")
    with open(p, "w") as f:
        for q in range(7):
            for d in range(3 + q):
                f.write("%d qid:q%d 1:%s 2:%d 4:0.25 # doc %d-%d\n" % ((q + d) % 3, q, repr(0.1 * d + q), d, q, d))
    try:
        FeatureManager.cache = True
        a = FeatureManager.readInput(p)
        assert os.path.exists(p + ".rlcache.npz")
        b = FeatureManager.readInput(p)                  # from the cache
        assert len(a) == len(b) == 7
        for x, y in zip(a, b):
            assert x.getID() == y.getID() and x.size() == y.size()
            for i in range(x.size()):
                assert x.get(i).getLabel() == y.get(i).getLabel() and x.get(i).getDescription() == y.get(i).getDescription()
                assert [x.get(i).getFeatureValue(f) for f in (1, 2, 3, 4)] == [y.get(i).getFeatureValue(f) for f in (1, 2, 3, 4)]
        time.sleep(0.05)
        with open(p, "a") as f:
            f.write("2 qid:q9 1:5 2:5 4:5 # late\n")
        os.utime(p, (time.time() + 5, time.time() + 5))
        c = FeatureManager.readInput(p)                  # the text is newer: parsed again
        assert len(c) == 8
    finally:
        FeatureManager.cache = False


def test_hr_drops_lists_without_a_relevant_document_and_idv_format(tmp_path):
    """-hr (eval/Evaluator.java:367-368 -> FeatureManager.readInput mustHaveRelDoc, :230-233); the -idv line format
    `<metric>   <qid>   <Double.toString(score)>` with the closing "all" line (:915-944, :1343-1352) checked on a hand-made model"""
    from ranklib_amd import evaluator as E
    from ranklib_amd.features import FeatureManager
    p = str(tmp_path / "t.txt")
    with open(p, "w") as f:
        f.write("0 qid:a 1:1 2:0\n0 qid:a 1:0 2:1\n")          # no relevant document
        f.write("2 qid:b 1:1 2:0\n0 qid:b 1:0 2:1\n1 qid:b 1:0.5 2:0.5\n")
    assert len(FeatureManager.readInput(p)) == 2 and len(FeatureManager.readInput(p, True)) == 1
    model = str(tmp_path / "m.txt")
    with open(model, "w") as f:
        f.write("## LambdaMART\n## No. of trees = 1\n## No. of leaves = 2\n## No. of threshold candidates = 256\n## Learning rate = 0.1\n## Stop early = 100\n\n"
                "<ensemble>\n\t<tree id=\"1\" weight=\"0.1\">\n\t\t<split>\n\t\t\t<feature>1 </feature>\n\t\t\t<threshold> 0.25 </threshold>\n"
                "\t\t\t<split pos=\"left\">\n\t\t\t\t<output>-1.0 </output>\n\t\t\t</split>\n\t\t\t<split pos=\"right\">\n\t\t\t\t<output>2.0 </output>\n\t\t\t</split>\n"
                "\t\t</split>\n\t</tree>\n</ensemble>\n")
    E.Evaluator.mustHaveRelDoc = True
    try:
        assert len(E._read_input(p)) == 1
    finally:
        E.Evaluator.mustHaveRelDoc = False


@pytest.mark.gpu
def test_cli_idv_and_hr_flow(tmp_path):
    """`-load m -test f -idv out [-hr]` (eval/Evaluator.java:915-944,1343-1352): one `<metric>   <qid>   <score>` line per ranked list that
    survives -hr, then the mean as `all`; scores printed as Double.toString does"""
    p = str(tmp_path / "t.txt")
    with open(p, "w") as f:
        f.write("0 qid:a 1:1 2:0\n0 qid:a 1:0 2:1\n")          # no relevant document: dropped by -hr
        f.write("2 qid:b 1:1 2:0\n0 qid:b 1:0 2:1\n1 qid:b 1:0.5 2:0.5\n")
        f.write("0 qid:c 1:1 2:0\n1 qid:c 1:0 2:1\n")
    model = str(tmp_path / "m.txt")
    with open(model, "w") as f:
        f.write("## LambdaMART\n## No. of trees = 1\n## No. of leaves = 2\n## No. of threshold candidates = 256\n## Learning rate = 0.1\n## Stop early = 100\n\n"
                "<ensemble>\n\t<tree id=\"1\" weight=\"0.1\">\n\t\t<split>\n\t\t\t<feature>1 </feature>\n\t\t\t<threshold> 0.25 </threshold>\n"
                "\t\t\t<split pos=\"left\">\n\t\t\t\t<output>-1.0 </output>\n\t\t\t</split>\n\t\t\t<split pos=\"right\">\n\t\t\t\t<output>2.0 </output>\n\t\t\t</split>\n"
                "\t\t</split>\n\t</tree>\n</ensemble>\n")
    out = str(tmp_path / "idv.txt")
    evaluator.main(["-load", model, "-test", p, "-metric2T", "NDCG@10", "-idv", out])
    rows = [l.split("   ") for l in open(out).read().splitlines()]
    assert [r[1] for r in rows] == ["a", "b", "c", "all"] and all(r[0] == "NDCG@10" for r in rows)
    # the model ranks feature-1 > 0.25 first: b = [2, 1, 0] (perfect), c = [0, 1]; a has no relevant document (NDCG 0)
    assert rows[0][2] == "0.0" and rows[1][2] == "1.0"
    c = (2 ** 1 - 1) / np.log2(3) / 1.0
    assert abs(float(rows[2][2]) - c) < 1e-12 and abs(float(rows[3][2]) - (0.0 + 1.0 + c) / 3) < 1e-12
    evaluator.main(["-load", model, "-test", p, "-metric2T", "NDCG@10", "-idv", out, "-hr"])
    rows = [l.split("   ") for l in open(out).read().splitlines()]
    assert [r[1] for r in rows] == ["b", "c", "all"] and abs(float(rows[2][2]) - (1.0 + c) / 2) < 1e-12


# ---- -norm: features/SumNormalizor.java, ZScoreNormalizor.java, LinearNormalizer.java ----------------------------------------------
def _literal_normalize(kind, rows, fids):
    """the Java loops, one scalar operation at a time (numpy scalars for the float / double distinction)"""
    f32, f64 = np.float32, np.float64
    val = lambda r, f: (f32(0) if np.isnan(r[f]) else f32(r[f]))      # noqa: E731  getFeatureValue
    n = len(rows)
    if kind == "sum":
        norm = [f64(0)] * len(fids)
        for r in rows:
            for j, f in enumerate(fids):
                norm[j] = norm[j] + f64(abs(val(r, f)))
        for r in rows:
            for j, f in enumerate(fids):
                if norm[j] > 0:
                    r[f] = f32(f64(val(r, f)) / norm[j])
    elif kind == "zscore":
        means = [f64(0)] * len(fids)
        for r in rows:
            for j, f in enumerate(fids):
                means[j] = means[j] + f64(val(r, f))
        for j, f in enumerate(fids):
            means[j] = means[j] / f64(n)
            std = f64(0)
            for r in rows:
                x = f64(val(r, f)) - means[j]
                std = std + x * x
            with np.errstate(divide="ignore", invalid="ignore"):
                std = np.sqrt(std / f64(n - 1))
            if std > 0:
                for r in rows:
                    r[f] = f32((f64(val(r, f)) - means[j]) / std)
    else:
        for j, f in enumerate(fids):
            lo, hi = np.finfo(f32).max, f32(1.4e-45)
            for r in rows:
                lo = min(lo, val(r, f)); hi = max(hi, val(r, f))
            for r in rows:
                r[f] = (val(r, f) - lo) / (hi - lo) if hi > lo else f32(0)


@pytest.mark.parametrize("kind", ["sum", "zscore", "linear"])
def test_normalizers_follow_the_java_loops_bit_for_bit(kind):
    from ranklib_amd import normalizer as NM
    from ranklib_amd.learning import DataPoint as DP, RankList as RL
    rng = np.random.RandomState(3)
    nm = NM.create(kind)
    assert nm.name() == kind
    for case in range(40):
        n, F = int(rng.randint(1, 30)), 6
        rows = (rng.randn(n, F + 1) * rng.choice([1e-3, 1.0, 1e4])).astype(np.float32)
        rows[:, 2] = 7.5                                   # a constant column
        rows[:, 3] = -np.abs(rows[:, 3])                   # no positive value: LinearNormalizer's max stays at Float.MIN_VALUE
        rows[rng.rand(n) < 0.2, 4] = np.nan                # unknown values read as 0
        fids = [1, 2, 3, 4, 6, 3] if case % 2 else None    # a feature list with a duplicate, or every feature
        pts = [DP.from_parsed(1.0, "q", "", rows[i].copy()) for i in range(n)]
        rl = RL(pts)
        want = [rows[i].copy() for i in range(n)]
        _literal_normalize(kind, want, sorted(set(fids)) if fids else list(range(1, F + 1)))
        nm.normalize(rl, fids)
        for i in range(n):
            assert np.array_equal(pts[i].fVals[1:].view(np.uint32), want[i][1:].view(np.uint32)), (kind, case, i)


def test_normalizer_on_ragged_rows_with_a_feature_subset():
    """rows end at their own last feature id (DataPoint.parse): a subset whose largest id lies below the shortest row must not need equal row lengths"""
    from ranklib_amd import normalizer as NM
    from ranklib_amd.learning import DataPoint as DP, RankList as RL
    a = np.array([np.nan, 1.0, 3.0, 5.0], np.float32)      # features 1..3
    b = np.array([np.nan, 3.0, 1.0], np.float32)           # features 1..2
    rl = RL([DP.from_parsed(1.0, "q", "", a.copy()), DP.from_parsed(0.0, "q", "", b.copy())])
    NM.create("sum").normalize(rl, [1, 2])
    assert np.array_equal(rl.rl[0].fVals[1:3], np.array([0.25, 0.75], np.float32)) and rl.rl[0].fVals[3] == 5.0
    assert np.array_equal(rl.rl[1].fVals[1:3], np.array([0.75, 0.25], np.float32))


def test_norm_flag_changes_the_training_data_and_unknown_normalizer_is_an_error(tmp_path):
    from ranklib_amd import evaluator as E
    with pytest.raises(Exception) as ei:
        E.main(["-train", "x", "-norm", "bogus"])
    assert "Unknown normalizor: bogus" in str(ei.value)
    E.Evaluator.normalize = False


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sum", "zscore", "linear"])
def test_cli_norm_trains_on_the_normalised_lists(tmp_path, kind):
    """-norm <method> (eval/Evaluator.java:256-267, :687-695): the model equals the one trained through the API on lists normalised by
    hand, differs from the model of the raw data, and -rank / -load normalise the test lists with the model's features (:1173-1175)"""
    from ranklib_amd import normalizer as NM
    from ranklib_amd.features import FeatureManager
    data, m_cli, m_raw, run = (str(tmp_path / n) for n in ("data.txt", "cli.txt", "raw.txt", "run.txt"))
    write_random_data(data)
    saved = (learning.LambdaMART.nTrees, learning.LambdaMART.nTreeLeaves)
    try:
        evaluator.main(["-train", data, "-ranker", "6", "-metric2t", "NDCG@10", "-tree", "6", "-leaf", "6", "-norm", kind, "-save", m_cli])
        evaluator.main(["-rank", data, "-load", m_cli, "-norm", kind, "-indri", run])
        evaluator.main(["-train", data, "-ranker", "6", "-metric2t", "NDCG@10", "-tree", "6", "-leaf", "6", "-save", m_raw])
        assert evaluator.Evaluator.normalize is False          # the flag does not outlive a command line (:86)
        lists = FeatureManager.readInput(data)
        feats = FeatureManager.getFeatureFromSampleVector(lists)
        NM.create(kind).normalizeAll(lists, feats)
        learning.LambdaMART.nTrees, learning.LambdaMART.nTreeLeaves = 6, 6
        from ranklib_amd.metric import MetricScorerFactory
        r = learning.RankerTrainer().train(learning.RankerType.LAMBDAMART, lists, feats, MetricScorerFactory().createScorer("NDCG@10"))
    finally:
        learning.LambdaMART.nTrees, learning.LambdaMART.nTreeLeaves = saved
    cli = open(m_cli).read()
    assert cli[cli.index("<ensemble>"):] == r.model()[r.model().index("<ensemble>"):]
    assert cli[cli.index("<ensemble>"):] != open(m_raw).read()[open(m_raw).read().index("<ensemble>"):]
    assert len(open(run).read().splitlines()) > 0


# ---- -qrel ---------------------------------------------------------------------------------------------------------------------------
def test_qrel_file_feeds_the_ndcg_cache_and_the_map_counts(tmp_path):
    """metric/NDCGScorer.java:50-96 (one idealGains entry per RUN of equal qids, ideal DCG of ALL judged documents at min(k, #judged));
    metric/APScorer.java:45-66, :86-94 (relDocCount per qid; a list whose qid is not in the file scores 0)"""
    from ranklib_amd import metric as M
    from ranklib_amd.learning import DataPoint as DP, RankList as RL
    q = tmp_path / "qrels.txt"
    q.write_text("q1 0 d1 2\nq1 0 d2 0\nq1 0 d3 1.6\n\nq2 0 d1 0\nq2 0 d9 3\nq1 0 d7 1\n")
    nd = M.NDCGScorer(2)
    nd.loadExternalRelevanceJudgment(str(q))
    assert nd.idealGains["q2"] == M.gain(3) * M.discount(0) + M.gain(0) * M.discount(1)
    assert nd.idealGains["q1"] == M.gain(1) * M.discount(0)              # the later run of q1 (one document) overwrote the first
    ap = M.APScorer()
    ap.loadExternalRelevanceJudgment(str(q))
    assert ap.relDocCount == {"q1": 3, "q2": 1}                          # rint(1.6) = 2 > 0 counts
    mk = lambda qid, labels: RL([DP.from_parsed(float(l), qid, "", np.array([np.nan, 1.0], np.float32)) for l in labels])   # noqa: E731
    assert ap.score(mk("q1", [1, 0, 1])) == (1.0 / 1 + 2.0 / 3) / 3      # three relevant documents judged, two retrieved
    assert ap.score(mk("zz", [1, 1])) == 0.0                             # qid not in the file: rdCount stays 0
    assert M.APScorer().score(mk("zz", [1, 1])) == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("m2t", ["NDCG@10", "MAP"])
def test_cli_qrel_reaches_the_trainer(tmp_path, m2t):
    """-qrel <file> (eval/Evaluator.java:243-244, :580-591): the saved model differs from the one trained without the judgments and equals
    the one trained through the API with a scorer that loaded the same file"""
    from ranklib_amd.features import FeatureManager
    from ranklib_amd.metric import MetricScorerFactory
    data, qrel, m_q, m_raw = (str(tmp_path / n) for n in ("data.txt", "qrels.txt", "q.txt", "raw.txt"))
    rng = np.random.RandomState(11)
    with open(data, "w") as f:
        for q in range(40):
            for d in range(int(rng.randint(4, 15))):
                x = rng.rand(4)
                f.write("%d qid:%d 1:%f 2:%f 3:%f 4:%f # d%d_%d\n" % (int(3 * x[0] * x[1] + rng.rand()), q, x[0], x[1], x[2], x[3], q, d))
    lists = FeatureManager.readInput(data)
    with open(qrel, "w") as f:            # judgments for two thirds of the queries: more relevant documents than the lists retrieved
        for rl in lists[: 2 * len(lists) // 3]:
            for d in range(rl.size() + 5):
                f.write("%s 0 doc%d %d\n" % (rl.getID(), d, int(rng.randint(0, 5))))
    saved = (learning.LambdaMART.nTrees, learning.LambdaMART.nTreeLeaves)
    try:
        evaluator.main(["-train", data, "-ranker", "6", "-metric2t", m2t, "-tree", "5", "-leaf", "6", "-qrel", qrel, "-save", m_q])
        evaluator.main(["-train", data, "-ranker", "6", "-metric2t", m2t, "-tree", "5", "-leaf", "6", "-save", m_raw])
        assert evaluator.Evaluator.qrelFile == ""
        learning.LambdaMART.nTrees, learning.LambdaMART.nTreeLeaves = 5, 6
        sc = MetricScorerFactory().createScorer(m2t)
        sc.loadExternalRelevanceJudgment(qrel)
        feats = FeatureManager.getFeatureFromSampleVector(lists)
        r = learning.RankerTrainer().train(learning.RankerType.LAMBDAMART, lists, feats, sc)
    finally:
        learning.LambdaMART.nTrees, learning.LambdaMART.nTreeLeaves = saved
    cli, raw = open(m_q).read(), open(m_raw).read()
    assert cli[cli.index("<ensemble>"):] == r.model()[r.model().index("<ensemble>"):]
    assert cli[cli.index("<ensemble>"):] != raw[raw.index("<ensemble>"):]
