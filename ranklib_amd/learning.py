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
import logging
import math
import time

import numpy as np

from . import _native as N
from ._native import RankLibError
from .metric import TRAINABLE

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
        t = N.Trainer(n_trees=cls.nTrees, n_leaves=cls.nTreeLeaves, learning_rate=cls.learningRate, n_threshold=cls.nThreshold,
                      min_leaf_support=cls.minLeafSupport, early_stop_rounds=cls.nRoundToStopEarly, metric_k=self.scorer.getK(),
                      device=cls.device, metric=metric, ranker=self._RANKER)
        t.set_train(X, lab, qoff, feature_ids=self.features, qkey=qkey)
        if self.validationSamples is not None:
            Xv, lv, qv, _ = flatten(self.validationSamples, self.features)
            ids = {}
            for q, rl in enumerate(self.samples):
                ids.setdefault(rl.getID(), int(qkey[q]))
            vkey = np.array([ids.setdefault(rl.getID(), nk + i) for i, rl in enumerate(self.validationSamples)], np.int32)
            t.set_validation(Xv, lv, qv, qkey=vkey)
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
            self.printLog([9], [repr(java_round(float(tm), 4))])
            if vm is not None:
                self.printLog([9], [repr(java_round(float(vm), 4))])
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


class RankerFactory:                  # learning/RankerFactory.java:36-118
    def __init__(self):
        self.map = {"LAMBDAMART": LambdaMART, "MART": MART}

    def createRanker(self, rtype, samples=None, features=None, scorer=None):
        if isinstance(rtype, str):
            try:
                rtype = RankerType[rtype]
            except KeyError:
                raise RankLibError("Could find the class \"%s\" you specified. Make sure the jar library is in your classpath." % rtype)
        if rtype.name not in self.map:
            raise RankLibError("rlhip builds -ranker 6 (LambdaMART) and -ranker 0 (MART) only; %s is out of scope (SURVEY.md 8)" % rtype.name)
        r = self.map[rtype.name]()
        if samples is not None:
            r.setTrainingSet(samples)
            r.setFeatures(features)
            r.setMetricScorer(scorer)
        return r

    def loadRankerFromString(self, fullText):      # :108-118: the first line names the algorithm
        first = fullText.split("\n", 1)[0]
        name = first.replace("## ", "").strip()
        if name.upper() not in self.map:
            raise RankLibError("Model file does not start with '## LambdaMART' or '## MART' (got %r)" % first)
        r = self.createRanker(RankerType[name.upper()])
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
