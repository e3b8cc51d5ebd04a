"""Host-side metric objects of the `-ranker 6` path: what `-metric2t NDCG@k` parses to and what Evaluator uses to
report a ranked test list.  Mirrors metric/MetricScorer.java, metric/DCGScorer.java, metric/NDCGScorer.java and
metric/MetricScorerFactory.java.  (The per-round training / validation metric is computed on the GPU; this
class only scores already-ranked lists for the test-time report, as Evaluator does.)"""
import math

from ._native import RankLibError


class MetricScorer:                   # metric/MetricScorer.java:20-69
    def __init__(self, k=10):
        self.k = k

    def setK(self, k):
        self.k = k

    def getK(self):
        return self.k

    def score(self, rl):
        if isinstance(rl, list):      # score(List<RankList>): double mean  :46-52
            s = 0.0
            for x in rl:
                s += self.score(x)
            return s / len(rl)
        return self.scoreOne(rl)


def discount(i):                      # metric/DCGScorer.java:26, utilities/SimpleMath.java:24-26
    return 1.0 / (math.log(i + 2) / math.log(2))


def gain(rel):                        # metric/DCGScorer.java:28-31
    return float((1 << rel) - 1)


class NDCGScorer(MetricScorer):       # metric/NDCGScorer.java:29-175
    def __init__(self, k=10):
        super().__init__(k)
        self.idealGains = {}

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


class MetricScorerFactory:            # metric/MetricScorerFactory.java:17-60
    def createScorer(self, metric, k=None):
        m, kk = metric, k
        if "@" in metric:
            m, ks = metric.split("@", 1)
            kk = int(ks)
        m = m.upper()
        if m != "NDCG":
            raise RankLibError("rlhip builds NDCG@k only (SURVEY.md 8f lists MAP / ERR / DCG as next); got %r" % metric)
        s = NDCGScorer()
        if kk is not None:
            s.setK(kk)
        return s
