"""Host-side mirror of RankLib's plugin surface for the `-ranker 6` path.

Same class and method names, argument meaning and error behaviour as the Java, so code and tests written against
RankLib's API read the same here:

    DataPoint / DenseDataPoint   learning/DataPoint.java:22-200, learning/DenseDataPoint.java:10-51
    RankList                     learning/RankList.java:21-114
    Ranker (abstract)            learning/Ranker.java:36-186
    LambdaMART                   learning/tree/LambdaMART.java:33-329   (init/learn run on the GPU through librlhip.so)
    RankerType / RankerFactory   learning/RankerType.java, learning/RankerFactory.java:36-118
    RankerTrainer                learning/RankerTrainer.java:23-56

There is no JVM in this environment; the Java drop-in class that does the same over JNI is in integration/.
All numeric work happens behind the C ABI (ranklib_amd/_native.py); nothing here computes a histogram, a
lambda or a tree on the CPU.
"""
import enum
import random
import logging
import math
import time

import numpy as np

from . import _native as N
from ._native import RankLibError
from .metric import TRAINABLE, ERRScorer

logger = logging.getLogger("ranklib_amd")


# ---------------------------------------------------------------------------------------------------------
class DataPoint:
    """learning/DataPoint.java + DenseDataPoint: `label qid:ID fid:val ... # description`"""
    missingZero = False               # DataPoint.missingZero (static)  learning/DataPoint.java:23
    __slots__ = ("label", "id", "description", "fVals", "cached")

    def __init__(self, text=None):
        self.label = 0.0
        self.id = ""
        self.description = ""
        self.fVals = None
        self.cached = -1.0
        if text is not None:
            self._parse(text)

    def _parse(self, text):           # learning/DataPoint.java:58-110
        try:
            idx = text.find("#")
            if idx != -1:
                self.description = text[idx:]
                text = text[:idx].strip()
            fs = text.split()
            self.label = float(np.float32(fs[0]))
            if self.label < 0:
                raise RankLibError("Relevance label cannot be negative. System will now exit.")
            self.id = fs[1][fs[1].rfind(":") + 1:]
            last = 0
            pairs = []
            for tok in fs[2:]:
                f = int(tok[:tok.index(":")])
                if f <= 0:
                    raise RankLibError("Cannot use feature numbering less than or equal to zero. Start your features at 1.")
                pairs.append((f, np.float32(tok[tok.rfind(":") + 1:])))
                last = max(last, f)
            fv = np.full(last + 1, np.nan, dtype=np.float32)      # fVals[0] is unused, UNKNOWN = NaN
            for f, v in pairs:
                fv[f] = v
            self.fVals = fv
        except RankLibError:
            raise
        except Exception as ex:       # noqa: BLE001 -- the reference wraps everything
            raise RankLibError("Error in DataPoint::parse() %s" % ex)

    @classmethod
    def from_parsed(cls, label, qid, description, fvals):
        """a DataPoint from already parsed fields (features.FeatureManager.readInput's native path)"""
        dp = cls.__new__(cls)
        dp.label = label
        dp.id = qid
        dp.description = description
        dp.fVals = fvals
        dp.cached = -1.0
        return dp

    def getFeatureValue(self, fid):   # learning/DenseDataPoint.java:21-32
        if fid <= 0 or fid >= len(self.fVals):
            if DataPoint.missingZero:
                return np.float32(0)
            raise RankLibError("Error in DenseDataPoint::getFeatureValue(): requesting unspecified feature, fid=%d" % fid)
        v = self.fVals[fid]
        return np.float32(0) if np.isnan(v) else v

    def getFeatureCount(self):
        return len(self.fVals) - 1

    def getLabel(self):
        return self.label

    def getID(self):
        return self.id

    def getDescription(self):
        return self.description


DenseDataPoint = DataPoint


class RankList:
    """learning/RankList.java: an ordered list of DataPoints of one query"""

    def __init__(self, rl, idx=None, offset=0):
        pts = rl.rl if isinstance(rl, RankList) else list(rl)
        self.rl = [pts[i - offset] for i in idx] if idx is not None else list(pts)
        self.featureCount = max((dp.getFeatureCount() for dp in self.rl), default=0)

    def getID(self):
        return self.rl[0].getID()

    def size(self):
        return len(self.rl)

    def __len__(self):
        return len(self.rl)

    def get(self, k):
        return self.rl[k]

    def getFeatureCount(self):
        return self.featureCount


def flatten(samples, features):
    """List[RankList] -> (X [n, len(features)] via getFeatureValue, labels, qoff, qkey): what LambdaMART.init()
    walks (learning/tree/LambdaMART.java:71-91); equal qid strings get equal keys (NDCGScorer cache quirk)."""
    n = sum(rl.size() for rl in samples)
    F = len(features)
    X = np.zeros((n, F), np.float32)
    labels = np.zeros(n, np.float32)
    qoff = np.zeros(len(samples) + 1, np.int32)
    keys, qkey = {}, np.zeros(len(samples), np.int32)
    fa = np.asarray(features, np.int64)
    k = 0
    for q, rl in enumerate(samples):
        qkey[q] = keys.setdefault(rl.getID(), len(keys))
        for dp in rl.rl:
            fv = dp.fVals
            if fa.size and (fa.min() <= 0 or fa.max() >= len(fv)):
                X[k] = [dp.getFeatureValue(int(f)) for f in features]
            else:
                row = fv[fa]
                X[k] = np.where(np.isnan(row), np.float32(0), row)
            labels[k] = dp.label
            k += 1
        qoff[q + 1] = k
    return X, labels, qoff, qkey


# ---------------------------------------------------------------------------------------------------------
def stable_desc_order(scores):
    """MergeSorter.sort(double[], false): stable, descending (utilities/MergeSorter.java:134-189)"""
    return np.argsort(-np.asarray(scores, np.float64), kind="stable")


class Ranker:
    """learning/Ranker.java:36-186"""

    def __init__(self, samples=None, features=None, scorer=None):
        self.samples = samples if samples is not None else []
        self.features = features
        self.scorer = scorer
        self.scoreOnTrainingData = 0.0
        self.bestScoreOnValidationData = 0.0
        self.validationSamples = None
        self._logbuf = ""

    def setTrainingSet(self, samples):
        self.samples = samples

    def setFeatures(self, features):
        self.features = features

    def setValidationSet(self, samples):
        self.validationSamples = samples

    def setMetricScorer(self, scorer):
        self.scorer = scorer

    def getScoreOnTrainingData(self):
        return self.scoreOnTrainingData

    def getScoreOnValidationData(self):
        return self.bestScoreOnValidationData

    def getFeatures(self):
        return self.features

    def rank(self, rl):               # Ranker.rank(RankList) / rank(List<RankList>)  :88-103
        if isinstance(rl, RankList):
            return RankList(rl, list(stable_desc_order(self.evalList(rl))))
        return [self.rank(x) for x in rl]

    def evalList(self, rl):
        return [self.eval(dp) for dp in rl.rl]

    def save(self, modelFile):        # :106-122
        d = modelFile.rsplit("/", 1)[0] if "/" in modelFile else None
        if d:
            import os
            os.makedirs(d, exist_ok=True)
        with open(modelFile, "w", encoding="ascii") as f:
            f.write(self.model())

    # fixed-width log table  learning/Ranker.java:124-155
    def printLog(self, lens, msgs):
        for ln, msg in zip(lens, msgs):
            self._logbuf += (msg[:ln] if len(msg) > ln else msg + " " * (ln - len(msg))) + " | "

    def printLogLn(self, lens, msgs):
        self.printLog(lens, msgs)
        self.flushLog()

    def flushLog(self):
        if self._logbuf:
            logger.info(self._logbuf)
            self._logbuf = ""

    def init(self):
        raise NotImplementedError

    def learn(self):
        raise NotImplementedError

    def eval(self, p):                # noqa: A003 -- RankLib's name
        return -1.0

    def createNew(self):
        raise NotImplementedError

    def model(self):
        raise NotImplementedError

    def loadFromString(self, fullText):
        raise NotImplementedError

    def name(self):
        raise NotImplementedError

    def printParameters(self):
        raise NotImplementedError


def java_round(val, n):               # utilities/SimpleMath.java:54-60
    p = 10 ** n
    return math.floor(val * p + .5) / p


class FeatureHistogram:
    """Only the process-global knob of learning/tree/FeatureHistogram.java:34 lives on the host: the fraction of the features
    every split attempt looks at (set by RFRanker.init, never restored -- like the Java static)."""
    samplingRate = 1.0
    seed = 0          # not in the Java (it draws from an unseeded Random): makes the draw reproducible, see rlhip.h rl_params.seed


class LambdaMART(Ranker):
    """learning/tree/LambdaMART.java with init()/learn() executed on an MI355X (librlhip.so)."""
    # process-global parameters, like the Java statics (:37-42)
    nTrees = 1000
    learningRate = 0.1
    nThreshold = 256
    nRoundToStopEarly = 100
    nTreeLeaves = 10
    minLeafSupport = 1
    device = 0
    _RANKER = "LAMBDAMART"

    def __init__(self, samples=None, features=None, scorer=None):
        super().__init__(samples, features, scorer)
        self.ensemble = None          # list of FlatTree (pre-order) after learn(); scoring model after loadFromString
        self.impacts = None
        self._trainer = None
        self._model = None
        self._model_text = None

    def init(self):                   # :68-166
        logger.info("Initializing... ")
        metric = self.scorer.name().split("@")[0].upper() if self.scorer is not None else None
        if metric not in TRAINABLE:
            raise RankLibError("rlhip: the train metric must be one of NDCG, DCG, MAP, ERR (got %s)" % (self.scorer.name() if self.scorer else None))
        cls = type(self)
        self.impacts = np.zeros(len(self.features))
        X, lab, qoff, qkey = flatten(self.samples, self.features)
        nk = int(qkey.max()) + 1 if len(qkey) else 0
        N.set_err_max(ERRScorer.MAX)              # the reference's static ERRScorer.MAX (-gmax) reaches the kernels through the library's static
        t = N.Trainer(n_trees=cls.nTrees, n_leaves=cls.nTreeLeaves, learning_rate=cls.learningRate, n_threshold=cls.nThreshold,
                      min_leaf_support=cls.minLeafSupport, early_stop_rounds=cls.nRoundToStopEarly, metric_k=self.scorer.getK(),
                      device=cls.device, metric=metric, ranker=self._RANKER,
                      feature_sampling_rate=FeatureHistogram.samplingRate, seed=FeatureHistogram.seed)
        t.set_train(X, lab, qoff, feature_ids=self.features, qkey=qkey)
        if self.validationSamples is not None:
            Xv, lv, qv, _ = flatten(self.validationSamples, self.features)
            ids = {}
            for q, rl in enumerate(self.samples):
                ids.setdefault(rl.getID(), int(qkey[q]))
            vkey = np.array([ids.setdefault(rl.getID(), nk + i) for i, rl in enumerate(self.validationSamples)], np.int32)
            t.set_validation(Xv, lv, qv, qkey=vkey)
        # what the scorer object already holds reaches the trainer list by list: idealGains entries (-qrel, NDCGScorer.java:50-96 -- or any
        # earlier use of the same scorer object: a cached entry is a cached entry) and relDocCount (-qrel, APScorer.java:45-66)
        for validation, lists in ((False, self.samples), (True, self.validationSamples)):
            if lists is None:
                continue
            ideal = rdc = None
            gains = getattr(self.scorer, "idealGains", None)
            if metric == "NDCG" and gains:
                ideal = np.array([gains.get(rl.getID(), np.nan) for rl in lists], np.float64)
            counts = getattr(self.scorer, "relDocCount", None)
            if metric == "MAP" and counts is not None:
                rdc = np.array([counts.get(rl.getID(), 0) for rl in lists], np.int32)
            if ideal is not None or rdc is not None:
                t.set_external_judgments(validation, ideal, rdc)
        t.init()
        self._trainer = t

    def learn(self):                  # :169-272
        cls = type(self)
        t = self._trainer
        logger.info("Training starts...")
        nm = self.scorer.name()
        if self.validationSamples is not None:
            self.printLogLn([7, 9, 9], ["#iter", nm + "-T", nm + "-V"])
        else:
            self.printLogLn([7, 9], ["#iter", nm + "-T"])
        for m in range(cls.nTrees):
            self.printLog([7], [str(m + 1)])
            _, tm, vm, stop = t.boost_round(want_tree=False)
            self.printLog([9], [java_double_str(java_round(float(tm), 4))])
            if vm is not None:
                self.printLog([9], [java_double_str(java_round(float(vm), 4))])
            self.flushLog()
            if stop:
                break
        ts, vs = t.finish()           # rollback to the best validation model + scorer.score(rank(samples))
        self.scoreOnTrainingData = ts
        logger.info("Finished sucessfully.")
        logger.info("%s on training data: %s", nm, java_round(ts, 4))
        if vs is not None:
            self.bestScoreOnValidationData = vs
            logger.info("%s on validation data: %s", nm, java_round(vs, 4))
        self.ensemble = [t.get_tree(i) for i in range(t.num_trees())]
        self._model_text = t.model_text()
        self._model = N.Model(self._model_text, cls.device)
        logger.info("-- FEATURE IMPACTS")          # impacts[] is never written in the reference either (:58,80,267-271)
        for i, f in enumerate(self.features):
            logger.info(" Feature %d reduced error %s", f, self.impacts[i])

    # --- scoring: Ensemble.eval (float accumulation) on the GPU -------------------------------------------
    def _rows(self, dps):
        width = max([int(max(self._model.features(), default=0)) + 1] + [len(dp.fVals) for dp in dps])
        rows = np.zeros((len(dps), width), np.float32)
        for i, dp in enumerate(dps):
            fv = dp.fVals
            rows[i, :len(fv)] = np.where(np.isnan(fv), np.float32(0), fv)
        if not DataPoint.missingZero:
            need = int(max(self._model.features(), default=0))
            for dp in dps:
                if need >= len(dp.fVals):
                    raise RankLibError("Error in DenseDataPoint::getFeatureValue(): requesting unspecified feature, fid=%d" % need)
        return rows

    def eval(self, dp):               # noqa: A003  :275-277
        return float(self._model.predict_rows(self._rows([dp]))[0])

    def evalList(self, rl):
        return [float(v) for v in self._model.predict_rows(self._rows(rl.rl))]

    def createNew(self):
        return LambdaMART()

    def toString(self):
        return self.model().split("\n\n", 1)[1]

    def model(self):                  # :290-301
        return self._model_text

    def loadFromString(self, fullText):   # :304-310
        self._model_text = fullText
        self._model = N.Model(fullText, type(self).device)
        self.features = [int(f) for f in self._model.features()]

    def name(self):
        return "LambdaMART"

    def getEnsemble(self):
        return self.ensemble

    def printParameters(self):        # :313-320
        cls = type(self)
        logger.info("No. of trees: %d", cls.nTrees)
        logger.info("No. of leaves: %d", cls.nTreeLeaves)
        logger.info("No. of threshold candidates: %d", cls.nThreshold)
        logger.info("Min leaf support: %d", cls.minLeafSupport)
        logger.info("Learning rate: %s", cls.learningRate)
        logger.info("Stop early: %d rounds without performance gain on validation data", cls.nRoundToStopEarly)


class MART(LambdaMART):
    """learning/tree/MART.java: LambdaMART with residual pseudo-responses and mean leaf outputs ("Inherits *ALL*
    parameters from LambdaMART": the class attributes above are shared, like the Java statics)."""
    _RANKER = "MART"

    def createNew(self):              # :36-39
        return MART()

    def name(self):                   # :41-44
        return "MART"


def java_double_str(v):
    """Double.toString of a Java double (score files, indri files, the per-round log table): shortest digits that round-trip (JDK >= 19),
    decimal notation for 1e-3 <= |v| < 1e7, computerised scientific notation ("1.0E-5") otherwise; repr() switches at 1e-4 / 1e16."""
    d = float(v)
    if d != d:
        return "NaN"
    if d in (float("inf"), float("-inf")):
        return "Infinity" if d > 0 else "-Infinity"
    if d == 0:
        return "-0.0" if math.copysign(1.0, d) < 0 else "0.0"
    a = abs(d)
    if 1e-3 <= a < 1e7:
        r = np.format_float_positional(np.float64(d), unique=True, trim="0")
        return r if "." in r else r + ".0"
    m, e = np.format_float_scientific(np.float64(d), unique=True, trim="0").split("e")
    return (m if "." in m else m + ".0") + "E" + str(int(e))


def java_float_str(v):
    """Float.toString of a Java float (header lines of the model files): shortest digits that round-trip, decimal notation for
    1e-3 <= |v| < 1e7, computerised scientific notation otherwise."""
    f = np.float32(v)
    if f == 0:
        return "-0.0" if np.signbit(f) else "0.0"
    a = abs(float(f))
    if 1e-3 <= a < 1e7:
        r = np.format_float_positional(f, unique=True, trim="0")
        return r if "." in r else r + ".0"
    m, e = np.format_float_scientific(f, unique=True, trim="0").split("e")
    return (m if "." in m else m + ".0") + "E" + str(int(e))


class Sampler:
    """learning/Sampler.java:25-68.  The Java draws from an unseeded java.util.Random; `seed` makes the bags reproducible."""

    def __init__(self, seed=None):
        self.rng = random.Random(seed)
        self.samples = None
        self.remains = None

    def doSampling(self, samplingPool, samplingRate, withReplacement):
        n = len(samplingPool)
        size = int(np.float32(samplingRate) * np.float32(n))          # (int) (samplingRate * samplingPool.size()), float arithmetic
        self.samples = []
        if withReplacement:
            used = [False] * n
            for _ in range(size):
                sel = self.rng.randrange(n)
                self.samples.append(samplingPool[sel])
                used[sel] = True
            self.remains = [samplingPool[i] for i in range(n) if not used[i]]
        else:
            pool = list(range(n))
            for _ in range(size):
                sel = self.rng.randrange(len(pool))
                self.samples.append(samplingPool[pool[sel]])
                del pool[sel]
            self.remains = [samplingPool[i] for i in pool]
        return self.samples

    def getSamples(self):
        return self.samples

    def getRemains(self):
        return self.remains


class RFRanker(Ranker):
    """learning/tree/RFRanker.java: bagging over MART / LambdaMART trained on the GPU.  Every bag is a sample of the training
    lists WITH replacement (Sampler), trained with feature sampling at every split attempt (FeatureHistogram.samplingRate);
    eval = mean over the bags of Ensemble.eval (:109-115)."""
    nBag = 300
    subSamplingRate = 1.0
    featureSamplingRate = 0.3
    rType = None                      # RankerType.MART, set below (the enum is defined after this class)
    nTrees = 1
    nTreeLeaves = 100
    learningRate = 0.1
    nThreshold = 256
    minLeafSupport = 1
    seed = 0                          # not in the Java: bag i samples with Random(mix(seed, i)) and draws features with the same seed

    def __init__(self, samples=None, features=None, scorer=None):
        super().__init__(samples, features, scorer)
        self.ensembles = None         # per bag: the "<ensemble>...</ensemble>\n" text (Ensemble.toString)
        self._models = None           # per bag: N.Model for scoring

    @staticmethod
    def bag_seed(seed, i):
        return (int(seed) * 0x9E3779B97F4A7C15 + (i + 1) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def init(self):                   # :57-69 -- overwrites LambdaMART's statics and never restores them, like the Java
        logger.info("Initializing... ")
        cls = type(self)
        self.ensembles = [None] * cls.nBag
        LambdaMART.nTrees = cls.nTrees
        LambdaMART.nTreeLeaves = cls.nTreeLeaves
        LambdaMART.learningRate = cls.learningRate
        LambdaMART.nThreshold = cls.nThreshold
        LambdaMART.minLeafSupport = cls.minLeafSupport
        LambdaMART.nRoundToStopEarly = -1          # no early stopping inside a bag
        FeatureHistogram.samplingRate = cls.featureSamplingRate

    def learn(self):                  # :72-107
        cls = type(self)
        rf = RankerFactory()
        logger.info("Training starts...")
        nm = self.scorer.name()
        self.printLogLn([9, 9, 11], ["bag", nm + "-B", nm + "-OOB"])
        impacts = None
        self._models = []
        for i in range(cls.nBag):
            bs = self.bag_seed(cls.seed, i)
            sp = Sampler(bs)
            bag = sp.doSampling(self.samples, cls.subSamplingRate, True)
            FeatureHistogram.seed = bs
            r = rf.createRanker(cls.rType, bag, self.features, self.scorer)
            r.init()
            r.learn()
            impacts = r.impacts if impacts is None else impacts + r.impacts
            self.printLogLn([9, 9], ["b[%d]" % (i + 1), java_double_str(java_round(r.getScoreOnTrainingData(), 4))])
            self.ensembles[i] = r.toString()
            self._models.append(r._model)
        self.scoreOnTrainingData = self.scorer.score(self.rank(self.samples))
        logger.info("Finished sucessfully.")
        logger.info("%s on training data: %s", nm, java_round(self.scoreOnTrainingData, 4))
        if self.validationSamples is not None:
            self.bestScoreOnValidationData = self.scorer.score(self.rank(self.validationSamples))
            logger.info("%s on validation data: %s", nm, java_round(self.bestScoreOnValidationData, 4))
        logger.info("-- FEATURE IMPACTS")
        for i, f in enumerate(self.features):
            logger.info(" Feature %d reduced error %s", f, impacts[i])

    def _rows(self, dps):
        need = max(int(max(m.features(), default=0)) for m in self._models)
        width = max([need + 1] + [len(dp.fVals) for dp in dps])
        rows = np.zeros((len(dps), width), np.float32)
        for i, dp in enumerate(dps):
            fv = dp.fVals
            rows[i, :len(fv)] = np.where(np.isnan(fv), np.float32(0), fv)
        if not DataPoint.missingZero:
            for dp in dps:
                if need >= len(dp.fVals):
                    raise RankLibError("Error in DenseDataPoint::getFeatureValue(): requesting unspecified feature, fid=%d" % need)
        return rows

    def evalList(self, rl):           # :109-115 for every document of the list: double s += (float) ensemble.eval; s / nBag
        rows = self._rows(rl.rl)
        s = np.zeros(len(rows), np.float64)
        for m in self._models:
            s += m.predict_rows(rows).astype(np.float64)
        return [float(v) for v in s / len(self._models)]

    def eval(self, dp):               # noqa: A003
        return self.evalList(RankList([dp]))[0]

    def createNew(self):
        return RFRanker()

    def toString(self):               # :122-128
        return "".join(e + "\n" for e in self.ensembles)

    def model(self):                  # :131-143
        cls = type(self)
        out = "## " + self.name() + "\n"
        out += "## No. of bags = %d\n" % cls.nBag
        out += "## Sub-sampling = %s\n" % java_float_str(cls.subSamplingRate)
        out += "## Feature-sampling = %s\n" % java_float_str(cls.featureSamplingRate)
        out += "## No. of trees = %d\n" % cls.nTrees
        out += "## No. of leaves = %d\n" % cls.nTreeLeaves
        out += "## No. of threshold candidates = %d\n" % cls.nThreshold
        out += "## Learning rate = %s\n" % java_float_str(cls.learningRate)
        out += "\n"
        return out + self.toString()

    def loadFromString(self, fullText):   # :146-178: every "<ensemble> ... </ensemble>" block is one bag
        text = "\n".join(ln for ln in fullText.split("\n") if not ln.startswith("##"))       # parsing/ModelLineProducer.java:43-78
        blocks = []
        pos = 0
        while True:
            a = text.find("<ensemble>", pos)
            if a < 0:
                break
            b = text.find("</ensemble>", a)
            if b < 0:
                raise RankLibError("Error in RFRanker::load(): unterminated <ensemble>")
            blocks.append(text[a:b + len("</ensemble>")])
            pos = b + len("</ensemble>")
        if not blocks:
            raise RankLibError("Error in RFRanker::load(): no <ensemble> in the model")
        self.ensembles = [blk + "\n" for blk in blocks]
        self._models = [N.Model("## LambdaMART\n\n" + blk + "\n", LambdaMART.device) for blk in blocks]
        feats = []
        for m in self._models:                  # insertion-ordered set (the Java uses a HashSet: iteration order unspecified)
            for f in m.features():
                if int(f) not in feats:
                    feats.append(int(f))
        self.features = feats

    def name(self):
        return "Random Forests"

    def getEnsembles(self):
        return self.ensembles

    def printParameters(self):        # :181-189
        cls = type(self)
        logger.info("No. of bags: %d", cls.nBag)
        logger.info("Sub-sampling: %s", java_float_str(cls.subSamplingRate))
        logger.info("Feature-sampling: %s", java_float_str(cls.featureSamplingRate))
        logger.info("No. of trees: %d", cls.nTrees)
        logger.info("No. of leaves: %d", cls.nTreeLeaves)
        logger.info("No. of threshold candidates: %d", cls.nThreshold)
        logger.info("Learning rate: %s", java_float_str(cls.learningRate))


# ---------------------------------------------------------------------------------------------------------
class RankerType(enum.Enum):          # learning/RankerType.java
    MART = 0
    RANKBOOST = 1
    RANKNET = 2
    ADARANK = 3
    COOR_ASCENT = 4
    LAMBDARANK = 5
    LAMBDAMART = 6
    LISTNET = 7
    RANDOM_FOREST = 8
    LINEAR_REGRESSION = 9


RFRanker.rType = RankerType.MART


class RankerFactory:                  # learning/RankerFactory.java:36-118
    def __init__(self):
        self.map = {"LAMBDAMART": LambdaMART, "MART": MART, "RANDOM_FOREST": RFRanker}
        self.names = {"LAMBDAMART": "LAMBDAMART", "MART": "MART", "RANDOM FORESTS": "RANDOM_FOREST"}     # name().toUpperCase() -> type (:44-53)

    def createRanker(self, rtype, samples=None, features=None, scorer=None):
        if isinstance(rtype, str):
            try:
                rtype = RankerType[rtype]
            except KeyError:
                raise RankLibError("Could find the class \"%s\" you specified. Make sure the jar library is in your classpath." % rtype)
        if rtype.name not in self.map:
            raise RankLibError("rlhip builds -ranker 6 (LambdaMART), 0 (MART) and 8 (Random Forests) only; %s is out of scope (SURVEY.md 8)" % rtype.name)
        r = self.map[rtype.name]()
        if samples is not None:
            r.setTrainingSet(samples)
            r.setFeatures(features)
            r.setMetricScorer(scorer)
        return r

    def loadRankerFromString(self, fullText):      # :108-118: the first line names the algorithm
        first = fullText.split("\n", 1)[0]
        name = first.replace("## ", "").strip()
        if name.upper() not in self.names:
            raise RankLibError("Model file does not start with '## LambdaMART', '## MART' or '## Random Forests' (got %r)" % first)
        r = self.createRanker(RankerType[self.names[name.upper()]])
        r.loadFromString(fullText)
        return r

    def loadRankerFromFile(self, modelFile):       # :104-106
        with open(modelFile, "r", encoding="ascii") as f:
            return self.loadRankerFromString(f.read())


class RankerTrainer:                  # learning/RankerTrainer.java:23-56
    def __init__(self):
        self.rf = RankerFactory()
        self.trainingTime = 0.0

    def train(self, rtype, train, validation_or_features, features_or_scorer, scorer=None):
        if scorer is None:
            validation, features, scorer = None, validation_or_features, features_or_scorer
        else:
            validation, features = validation_or_features, features_or_scorer
        ranker = self.rf.createRanker(rtype, train, features, scorer)
        if validation is not None:
            ranker.setValidationSet(validation)
        start = time.perf_counter_ns()
        ranker.init()
        ranker.learn()
        self.trainingTime = time.perf_counter_ns() - start
        return ranker

    def getTrainingTime(self):
        return self.trainingTime

    def printTrainingTime(self):
        logger.info("Training time: %s seconds", java_round(self.trainingTime / 1e9, 2))
