"""Host-side metric objects: what `-metric2t` / `-metric2T` parse to and what Evaluator uses to report a ranked test
list.  Mirrors metric/MetricScorer.java, metric/{NDCG,DCG,AP,ERR,Precision,ReciprocalRank,BestAtK}Scorer.java and
metric/MetricScorerFactory.java.  The per-round training / validation metric (NDCG, DCG, MAP or ERR) is computed on the
GPU; these classes only score already-ranked lists for the test-time report, as Evaluator does."""
import math

import numpy as np

from ._native import RankLibError


class MetricScorer:                   # metric/MetricScorer.java:20-69
    def __init__(self, k=10):
        self.k = k

    def setK(self, k):
        self.k = k

    def getK(self):
        return self.k

    def loadExternalRelevanceJudgment(self, qrelFile):     # :42-44: only MAP and NDCG do something with it
        pass

    def score(self, rl):
        if isinstance(rl, list):      # score(List<RankList>): double mean  :46-52
            s = 0.0
            for x in rl:
                s += self.score(x)
            return s / len(rl)
        return self.scoreOne(rl)


def discount(i):                      # metric/DCGScorer.java:26, utilities/SimpleMath.java:24-26
    return 1.0 / (math.log(i + 2) / math.log(2))


def java_pow2m1(rel):                 # (1 << rel) - 1 in Java int arithmetic: shift count mod 32, wrapping subtraction
    v = ((1 << (rel & 31)) - 1) & 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def gain(rel):                        # metric/DCGScorer.java:28-31,137-139
    return float(java_pow2m1(rel))


def _qrel_lines(qrelFile):            # the loops of loadExternalRelevanceJudgment: trimmed lines split on ONE space, qid = s[0], label = rint(s[3])
    import gzip
    with (gzip.open(qrelFile, "rt", encoding="utf-8") if qrelFile.endswith(".gz") else open(qrelFile, "r", encoding="utf-8")) as f:   # FileUtils.smartReader
        for content in f:
            content = content.strip()
            if not content:
                continue
            s = content.split(" ")
            yield s[0].strip(), int(round_half_even(float(s[3].strip())))


def round_half_even(x):               # Math.rint
    return float(np.rint(x))


class NDCGScorer(MetricScorer):       # metric/NDCGScorer.java:29-175
    def __init__(self, k=10):
        super().__init__(k)
        self.idealGains = {}

    def getIdealDCG(self, rel, topK):  # DCGScorer.getIdealDCG via NDCGScorer.java:162-174: the topK largest labels, best first
        r = sorted(rel, reverse=True)
        dcg = 0.0
        for i in range(topK):
            dcg += gain(r[i]) * discount(i)
        return dcg

    def loadExternalRelevanceJudgment(self, qrelFile):     # :50-96: one idealGains entry per RUN of lines with the same qid (a later run overwrites)
        lastQID, rel = "", []
        for qid, label in _qrel_lines(qrelFile):
            if lastQID and lastQID != qid:
                self.idealGains[lastQID] = self.getIdealDCG(rel, self.k if len(rel) > self.k else len(rel))
                rel = []
            lastQID = qid
            rel.append(label)
        if rel:
            self.idealGains[lastQID] = self.getIdealDCG(rel, self.k if len(rel) > self.k else len(rel))

    def copy(self):
        return NDCGScorer()

    def name(self):
        return "NDCG@%d" % self.k

    def scoreOne(self, rl):           # :103-129
        n = rl.size()
        if n == 0:
            return 0.0
        size = self.k
        if self.k > n or self.k <= 0:
            size = n
        rel = [int(rl.get(i).getLabel()) for i in range(n)]
        ideal = self.idealGains.get(rl.getID())
        if ideal is None:
            r = sorted(rel, reverse=True)
            ideal = 0.0
            for i in range(size):
                ideal += gain(r[i]) * discount(i)
            self.idealGains[rl.getID()] = ideal
        if ideal <= 0.0:
            return 0.0
        dcg = 0.0
        for i in range(size):
            dcg += gain(rel[i]) * discount(i)
        return dcg / ideal


class DCGScorer(MetricScorer):        # metric/DCGScorer.java
    def __init__(self, k=10):
        super().__init__(k)

    def copy(self):
        return DCGScorer()

    def name(self):
        return "DCG@%d" % self.k

    def scoreOne(self, rl):           # :58-71
        n = rl.size()
        if n == 0:
            return 0.0
        size = n if (self.k > n or self.k <= 0) else self.k
        dcg = 0.0
        for i in range(size):
            dcg += gain(int(rl.get(i).getLabel())) * discount(i)
        return dcg


class APScorer(MetricScorer):         # metric/APScorer.java (K is ignored by score())
    def __init__(self):
        super().__init__(0)
        self.relDocCount = None       # :33

    def loadExternalRelevanceJudgment(self, qrelFile):     # :45-66
        self.relDocCount = {}
        for qid, label in _qrel_lines(qrelFile):
            if label > 0:
                self.relDocCount[qid] = self.relDocCount.get(qid, 0) + 1

    def copy(self):
        return APScorer()

    def name(self):
        return "MAP"

    def scoreOne(self, rl):           # :73-100
        ap, count = 0.0, 0
        for i in range(rl.size()):
            if rl.get(i).getLabel() > 0.0:
                count += 1
                ap += count / (i + 1)
        rdCount = count if self.relDocCount is None else self.relDocCount.get(rl.getID(), 0)      # :86-94
        return 0.0 if rdCount == 0 else ap / rdCount


class ERRScorer(MetricScorer):        # metric/ERRScorer.java
    MAX = 16.0

    def __init__(self, k=10):
        super().__init__(k)

    def copy(self):
        return ERRScorer()

    def name(self):
        return "ERR@%d" % self.k

    def scoreOne(self, rl):           # :45-64
        n = rl.size()
        size = n if (self.k > n or self.k <= 0) else self.k
        s, p = 0.0, 1.0
        for i in range(1, size + 1):
            R = java_pow2m1(int(rl.get(i - 1).getLabel())) / self.MAX
            s += p * R / i
            p *= (1.0 - R)
        return s


class PrecisionScorer(MetricScorer):  # metric/PrecisionScorer.java:28-40  (reporting only)
    def __init__(self, k=10):
        super().__init__(k)

    def copy(self):
        return PrecisionScorer()

    def name(self):
        return "P@%d" % self.k

    def scoreOne(self, rl):
        n = rl.size()
        size = n if (self.k > n or self.k <= 0) else self.k
        count = sum(1 for i in range(size) if rl.get(i).getLabel() > 0.0)
        return count / size


class ReciprocalRankScorer(MetricScorer):   # metric/ReciprocalRankScorer.java:21-35  (reporting only)
    def __init__(self):
        super().__init__(0)           # as written: k = 0 makes `size` 0, so plain "RR" always scores 0

    def copy(self):
        return ReciprocalRankScorer()

    def name(self):
        return "RR@%d" % self.k

    def scoreOne(self, rl):
        size = self.k if rl.size() > self.k else rl.size()
        for i in range(size):
            if rl.get(i).getLabel() > 0.0:
                return float(_f32(1.0) / _f32(i + 1))       # 1.0f / firstRank
        return 0.0


class BestAtKScorer(MetricScorer):    # metric/BestAtKScorer.java:28-56  (reporting only)
    def __init__(self, k=10):
        super().__init__(k)

    def copy(self):
        return BestAtKScorer()

    def name(self):
        return "Best@%d" % self.k

    def scoreOne(self, rl):
        size = self.k - 1
        if size < 0 or size > rl.size() - 1:
            size = rl.size() - 1
        mx, mi = -1.0, 0
        for i in range(size + 1):
            if mx < rl.get(i).getLabel():
                mx, mi = rl.get(i).getLabel(), i
        return float(rl.get(mi).getLabel())


def _f32(x):
    import numpy as np
    return np.float32(x)


TRAINABLE = ("NDCG", "DCG", "MAP", "ERR")     # metrics whose swapChange the GPU lambda kernels implement


class MetricScorerFactory:            # metric/MetricScorerFactory.java:17-60
    _map = {"MAP": APScorer, "NDCG": NDCGScorer, "DCG": DCGScorer, "P": PrecisionScorer, "RR": ReciprocalRankScorer,
            "BEST": BestAtKScorer, "ERR": ERRScorer}

    def createScorer(self, metric, k=None):
        m, kk = metric, k
        if "@" in metric:             # e.g. "NDCG@5"  :43-57
            m, ks = metric.split("@", 1)
            kk = int(ks)
        cls = self._map.get(m.upper())
        if cls is None:
            raise RankLibError("Unknown metric %r" % metric)
        s = cls()
        if kk is not None:
            s.setK(kk)
        return s
