"""Command line of the `-ranker 6` path: mirrors the flags of eval/Evaluator.java that reach LambdaMART
(:230-377) and the train / test / load / score / rank flows (:669-708, :1076-1094, :1168-1194).

    python -m ranklib_amd.evaluator -train f -ranker 6 -metric2t NDCG@10 -tree 1000 -leaf 31 -save model.txt
    python -m ranklib_amd.evaluator -load model.txt -rank f -score out.txt
"""
import logging
import math
import sys

from ._native import RankLibError
from . import normalizer
from .features import FeatureManager
from .learning import (DataPoint, FeatureHistogram, LambdaMART, RankerFactory, RankerTrainer, RankerType, RFRanker, java_double_str, java_round,
                       stable_desc_order)
from .metric import ERRScorer, MetricScorerFactory

logger = logging.getLogger("ranklib_amd")


class Evaluator:
    mustHaveRelDoc = False            # the reference's static of the same name (eval/Evaluator.java:551), set by -hr
    normalize = False                 # eval/Evaluator.java:553-554, set by -norm
    nml = normalizer.SumNormalizor()
    qrelFile = ""                     # :557, set by -qrel: TREC-style judgments, they only affect MAP and NDCG

    def __init__(self, rType, trainMetric, testMetric):
        self.type = rType
        mf = MetricScorerFactory()
        self.trainScorer = mf.createScorer(trainMetric)
        self.testScorer = mf.createScorer(testMetric)
        if Evaluator.qrelFile:                              # :579-582
            self.trainScorer.loadExternalRelevanceJudgment(Evaluator.qrelFile)
            self.testScorer.loadExternalRelevanceJudgment(Evaluator.qrelFile)
        self.rFact = RankerFactory()

    def evaluate(self, trainFile, validationFile=None, testFile=None, featureDefFile=None, modelFile=None):   # :669-708
        train = _read_input(trainFile)
        validation = _read_input(validationFile) if validationFile else None
        test = _read_input(testFile) if testFile else None
        features = FeatureManager.readFeature(featureDefFile) if featureDefFile else FeatureManager.getFeatureFromSampleVector(train)
        if Evaluator.normalize:                              # :687-695
            self.normalizeLists(train, features)
            if validation is not None:
                self.normalizeLists(validation, features)
            if test is not None:
                self.normalizeLists(test, features)
        trainer = RankerTrainer()
        if validation is not None:
            ranker = trainer.train(self.type, train, validation, features, self.trainScorer)
        else:
            ranker = trainer.train(self.type, train, features, self.trainScorer)
        trainer.printTrainingTime()
        if test is not None:
            s = self.testScorer.score(ranker.rank(test))
            logger.info("%s on test data: %s", self.testScorer.name(), java_round(s, 4))
        if modelFile:
            ranker.save(modelFile)
            logger.info("Model saved to: %s", modelFile)
        return ranker

    def _features(self, featureDefFile, samples):
        return FeatureManager.readFeature(featureDefFile) if featureDefFile else FeatureManager.getFeatureFromSampleVector(samples)

    def normalizeLists(self, samples, fids=None):            # Evaluator.normalize(List<RankList>[, fids]) :629-639 (the static flag has the name here)
        Evaluator.nml.normalizeAll(samples, fids)

    def _train(self, train, validation, features):
        trainer = RankerTrainer()
        if validation is not None:
            return trainer.train(self.type, train, validation, features, self.trainScorer)
        return trainer.train(self.type, train, features, self.trainScorer)

    def evaluate_tts(self, sampleFile, validationFile, featureDefFile, percentTrain, modelFile=None):     # -tts  :716-739
        samples = _read_input(sampleFile)
        features = self._features(featureDefFile, samples)
        if Evaluator.normalize:                              # prepareSplit :1329-1331: the whole file, before it is split
            self.normalizeLists(samples, features)
        train, test = FeatureManager.prepareSplit(samples, percentTrain)
        validation = _read_input(validationFile) if validationFile else None
        if validation is not None and Evaluator.normalize:   # :723-727
            self.normalizeLists(validation, features)
        ranker = self._train(train, validation, features)
        s = self.testScorer.score(ranker.rank(test))
        logger.info("%s on test data: %s", self.testScorer.name(), java_round(s, 4))
        if modelFile:
            ranker.save(modelFile)
            logger.info("Model saved to: %s", modelFile)
        return ranker, s

    def evaluate_tvs(self, trainFile, percentTrain, testFile, featureDefFile, modelFile=None):            # -tvs  :749-773
        samples = _read_input(trainFile)
        features = self._features(featureDefFile, samples)
        if Evaluator.normalize:
            self.normalizeLists(samples, features)
        train, validation = FeatureManager.prepareSplit(samples, percentTrain)
        test = _read_input(testFile) if testFile else None
        if test is not None and Evaluator.normalize:         # :756-760
            self.normalizeLists(test, features)
        ranker = self._train(train, validation, features)
        s = None
        if test is not None:
            s = self.testScorer.score(ranker.rank(test))
            logger.info("%s on test data: %s", self.testScorer.name(), java_round(s, 4))
        if modelFile:
            ranker.save(modelFile)
            logger.info("Model saved to: %s", modelFile)
        return ranker, s

    def evaluate_kcv(self, sampleFile, featureDefFile, nFold, tvs=-1.0, modelDir="", modelFile=""):      # -kcv  :798-873
        samples = _read_input(sampleFile)
        features = self._features(featureDefFile, samples)
        trainingData, validationData, testData = FeatureManager.prepareCV(samples, nFold, tvs)
        if Evaluator.normalize:
            # :816-822, kept as written: ALL folds are normalised once per fold, and the folds share their DataPoints (RankList's copy
            # constructor), so a list is normalised nFold x (the number of folds it appears in) times -- not idempotent in float arithmetic
            for _ in range(nFold):
                for group in (trainingData, validationData, testData):
                    for fold in group:
                        self.normalizeLists(fold, features)
        scores, scoreOnTrain, scoreOnTest, totalScoreOnTest, totalTestSampleSize = [], 0.0, 0.0, 0.0, 0
        for i in range(nFold):
            ranker = self._train(trainingData[i], validationData[i] if tvs > 0 else None, features)
            s2 = self.testScorer.score(ranker.rank(testData[i]))
            scoreOnTrain += ranker.getScoreOnTrainingData()
            scoreOnTest += s2
            totalScoreOnTest += s2 * len(testData[i])
            totalTestSampleSize += len(testData[i])
            scores.append((ranker.getScoreOnTrainingData(), s2))
            if modelDir:
                import os
                os.makedirs(modelDir, exist_ok=True)
                ranker.save(os.path.join(modelDir, "f%d.%s" % (i + 1, modelFile)))
                logger.info("Fold-%d model saved to: %s", i + 1, modelFile)
        logger.info("Summary:")
        logger.info("%s\t|   Train\t| Test", self.testScorer.name())
        for i, (a, b) in enumerate(scores):
            logger.info("Fold %d\t|   %s\t|  %s\t", i + 1, java_round(a, 4), java_round(b, 4))
        logger.info("Avg.\t|   %s\t|  %s\t", java_round(scoreOnTrain / nFold, 4), java_round(scoreOnTest / nFold, 4))
        logger.info("Total\t|   \t\t|  %s\t", java_round(totalScoreOnTest / totalTestSampleSize, 4))
        return scores

    def score(self, modelFile, testFile, outputFile):      # :1076-1094: qid \t index \t score
        ranker = self.rFact.loadRankerFromFile(modelFile)
        test = _read_input(testFile)
        if Evaluator.normalize:                              # :1080-1082
            self.normalizeLists(test, ranker.getFeatures())
        with open(outputFile, "w", encoding="utf-8") as out:
            for rl in test:
                for j, v in enumerate(ranker.evalList(rl)):
                    out.write("%s\t%d\t%s\n" % (rl.getID(), j, java_double_str(float(v))))

    def rank(self, modelFile, testFile, indriFile):        # :1168-1194: qid Q0 docno rank score indri
        ranker = self.rFact.loadRankerFromFile(modelFile)
        test = _read_input(testFile)
        if Evaluator.normalize:                              # :1173-1175
            self.normalizeLists(test, ranker.getFeatures())
        with open(indriFile, "w", encoding="utf-8") as out:
            for rl in test:
                sc = ranker.evalList(rl)
                for i, j in enumerate(stable_desc_order(sc)):
                    docno = rl.get(int(j)).getDescription().replace("#", "").strip()
                    out.write("%s Q0 %s %d %s indri\n" % (rl.getID(), docno, i + 1, java_double_str(java_round(float(sc[int(j)]), 5))))

    def test(self, modelFile, testFile, prpFile=""):       # evaluate a saved model (:915-944); -idv: performance per ranked list
        ranker = self.rFact.loadRankerFromFile(modelFile)
        test = _read_input(testFile)
        if Evaluator.normalize:                              # :919-921
            self.normalizeLists(test, ranker.getFeatures())
        ids, scores, rankScore = [], [], 0.0
        for rl in test:
            l = ranker.rank(rl)
            sc = self.testScorer.score(l)
            ids.append(l.getID()); scores.append(sc)
            rankScore += sc
        rankScore /= len(test)
        ids.append("all"); scores.append(rankScore)
        logger.info("%s on test data: %s", self.testScorer.name(), java_round(rankScore, 4))
        if prpFile:
            self.savePerRankListPerformanceFile(ids, scores, prpFile)
            logger.info("Per-ranked list performance saved to: %s", prpFile)
        return rankScore

    def savePerRankListPerformanceFile(self, ids, scores, prpFile):       # :1343-1352: "<metric>   <qid>   <Double.toString(score)>"
        with open(prpFile, "w", encoding="utf-8") as out:
            for i, sc in zip(ids, scores):
                out.write("%s   %s   %s\n" % (self.testScorer.name(), i, java_double_str(sc)))


def _read_input(inputFile):
    """Evaluator.readInput (eval/Evaluator.java:625-627): honours the static -hr switch"""
    return FeatureManager.readInput(inputFile, Evaluator.mustHaveRelDoc)


def main(argv=None):
    args = list(sys.argv[1:] if argv is None else argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    if not args:
        print("Usage: -train <file> -ranker 6|0|8 [-bag n -srate f -frate f -rtype 0|6 -seed n] [-metric2t NDCG@k|DCG@k|MAP|ERR@k] [-tree n] [-leaf n] [-shrinkage f] [-tc n] [-mls n] [-estop n] "
              "[-validate f] [-test f] [-feature f] [-norm sum|zscore|linear] [-qrel f] [-gmax g] [-save model] | -load model [-test f] [-rank f -indri out] [-score out]")
        return 0
    trainFile = validationFile = testFile = featureDescriptionFile = savedModelFile = rankFile = indriRankingFile = scoreFile = modelFile = prpFile = ""
    Evaluator.mustHaveRelDoc = False
    Evaluator.normalize = False                             # :86
    Evaluator.qrelFile = ""
    rankerType = 4                                          # the reference's default is Coordinate Ascent (:83)
    trainMetric, testMetric = "ERR@10", ""                  # the reference's default train metric (:84)
    ttSplit = tvSplit = 0.0
    foldCV, kcvModelDir, kcvModelFile = -1, "", ""
    i = 0
    while i < len(args):                                    # :230-372 (flags are matched case-insensitively)
        a = args[i].lower()

        def nxt():
            nonlocal i
            i += 1
            if i >= len(args):
                raise RankLibError("Missing value for " + args[i - 1])
            return args[i]
        if a == "-train": trainFile = nxt()
        elif a == "-ranker": rankerType = int(nxt())
        elif a == "-feature": featureDescriptionFile = nxt()
        elif a == "-metric2t": trainMetric = nxt()          # also captures -metric2T, exactly like the reference (:237-240)
        elif a == "-gmax": ERRScorer.MAX = math.pow(2.0, float(nxt()))      # eval/Evaluator.java:241-242
        elif a == "-validate": validationFile = nxt()
        elif a == "-test": testFile = nxt()
        elif a == "-save": modelFile = nxt()
        elif a == "-load": savedModelFile = nxt()
        elif a == "-rank": rankFile = nxt()
        elif a == "-score": scoreFile = nxt()
        elif a == "-indri": indriRankingFile = nxt()
        elif a == "-missingzero": DataPoint.missingZero = True
        elif a == "-cache": FeatureManager.cache = True      # not a RankLib flag: binary cache of the parsed LETOR files (features.py)
        elif a == "-sparse": pass                           # row storage only (:268-269)
        elif a == "-hr": Evaluator.mustHaveRelDoc = True    # :367-368: ranked lists without a relevant document are dropped by the reader
        elif a == "-idv": prpFile = nxt()                   # :281-282
        elif a == "-tree": LambdaMART.nTrees = RFRanker.nTrees = int(nxt())                 # :326-337: both sets of statics
        elif a == "-leaf": LambdaMART.nTreeLeaves = RFRanker.nTreeLeaves = int(nxt())
        elif a == "-shrinkage": LambdaMART.learningRate = RFRanker.learningRate = float(nxt())
        elif a == "-tc": LambdaMART.nThreshold = int(nxt())                                  # :300-303: NOT RFRanker.nThreshold
        elif a == "-mls": LambdaMART.minLeafSupport = RFRanker.minLeafSupport = int(nxt())
        elif a == "-estop": LambdaMART.nRoundToStopEarly = int(nxt())
        elif a == "-bag": RFRanker.nBag = int(nxt())                                         # :340-352
        elif a == "-srate": RFRanker.subSamplingRate = float(nxt())
        elif a == "-frate": RFRanker.featureSamplingRate = float(nxt())
        elif a == "-rtype":
            rt = int(nxt())
            if rt not in (0, 6):
                raise RankLibError("%s cannot be bagged. Random Forests only supports MART/LambdaMART." % rt)
            RFRanker.rType = RankerType(rt)
        elif a == "-seed": RFRanker.seed = FeatureHistogram.seed = int(nxt())               # rlhip extension: the Java draws are unseeded
        elif a == "-thread": nxt()                          # CPU thread pool of the reference: irrelevant here
        elif a == "-tts": ttSplit = float(nxt())            # :245-250
        elif a == "-tvs": tvSplit = float(nxt())
        elif a == "-kcv": foldCV = int(nxt())
        elif a == "-kcvmd": kcvModelDir = nxt()
        elif a == "-kcvmn": kcvModelFile = nxt()
        elif a == "-qrel": Evaluator.qrelFile = nxt()       # :243-244
        elif a in ("-nf", "-t"): nxt()                      # :359-364: newFeatureFile / topNew, statics that nothing in the reference reads
        elif a == "-keep": pass                             # keepOrigFeatures, likewise
        elif a == "-norm":                                   # :256-267
            Evaluator.nml = normalizer.create(nxt())
            Evaluator.normalize = True
        elif a in ("-round", "-epoch", "-tolerance", "-reg", "-r", "-i",
                   "-layer", "-node", "-lr", "-noeq", "-max", "-l2"):
            # parameters of the other rankers / of flows that are out of scope: parsed (the reference's own test passes
            # -round -epoch to every ranker, test:eval/EvaluatorTest.java:207-220) and ignored
            if a != "-noeq":
                nxt()
        elif a == "-device": LambdaMART.device = int(nxt())
        else:
            raise RankLibError("Unknown command-line parameter: " + args[i])     # :369-371 (incl. the documented -silent)
        i += 1
    if not testMetric:
        testMetric = trainMetric                            # :379-381
    if trainFile and rankerType not in (0, 6, 8):
        raise RankLibError("rlhip builds -ranker 6 (LambdaMART), -ranker 0 (MART) and -ranker 8 (Random Forests) only")
    e = Evaluator(RankerType(rankerType) if rankerType in (0, 6, 8) else RankerType.LAMBDAMART, trainMetric, testMetric)
    if trainFile:
        if foldCV != -1:                                    # :469-482
            if kcvModelDir and not kcvModelFile:
                kcvModelFile = "kcv"
            elif not kcvModelDir and kcvModelFile:
                kcvModelDir = "kcvmodels"
            e.evaluate_kcv(trainFile, featureDescriptionFile or None, foldCV, tvSplit if tvSplit > 0 else -1.0, kcvModelDir, kcvModelFile)
        elif ttSplit > 0.0:                                 # -tts overrides -tvs (:484-486)
            e.evaluate_tts(trainFile, validationFile or None, featureDescriptionFile or None, ttSplit, modelFile or None)
        elif tvSplit > 0.0:
            e.evaluate_tvs(trainFile, tvSplit, testFile or None, featureDescriptionFile or None, modelFile or None)
        else:
            e.evaluate(trainFile, validationFile or None, testFile or None, featureDescriptionFile or None, modelFile or None)
    elif savedModelFile:
        if rankFile and indriRankingFile:
            e.rank(savedModelFile, rankFile, indriRankingFile)
        elif rankFile and scoreFile:
            e.score(savedModelFile, rankFile, scoreFile)
        elif testFile:
            e.test(savedModelFile, testFile, prpFile)
    return 0


if __name__ == "__main__":
    sys.exit(main())
