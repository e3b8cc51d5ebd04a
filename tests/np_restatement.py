"""Second, independent restatement of RankLib's LambdaMART path in plain Python / numpy.

TEST INFRASTRUCTURE ONLY.  It exists to cross-check oracle/rl_oracle.c on tiny
inputs: two independent readings of the same Java reduce transcription risk,
because no JVM is available to arbitrate.  It is written
object-style, close to the shape of the Java (explicit n x n swapChange matrix,
Split objects, a sorted linked list as the growth queue) and deliberately
shares no code with the C oracle.

Citations are relative to /root/reference/src/main/java/ciir/umass/edu/.
Float semantics: Python float == Java double; numpy.float32 == Java float.
"""
import math
import struct

import numpy as np

F32 = np.float32
FLT_MAX = F32(3.4028234663852886e38)


# ----------------------------------------------------------------------------
# exp: fdlibm e_exp, restated in Python (Java StrictMath.exp semantics)
# ----------------------------------------------------------------------------
def _bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def _from_bits(u):
    return struct.unpack("<d", struct.pack("<Q", u & 0xFFFFFFFFFFFFFFFF))[0]


_LN2HI = _from_bits(0x3FE62E42FEE00000)
_LN2LO = _from_bits(0x3DEA39EF35793C76)
_INVLN2 = _from_bits(0x3FF71547652B82FE)
_P = [_from_bits(v) for v in (0x3FC555555555553E, 0xBF66C16C16BEBD93, 0x3F11566AAF25DE2C,
                              0xBEBBBD41C5D26BF1, 0x3E66376972BEA4D0)]


def jexp(x):
    hx = (_bits(x) >> 32) & 0xFFFFFFFF
    xsb = (hx >> 31) & 1
    hx &= 0x7FFFFFFF
    if hx >= 0x40862E42:
        if hx >= 0x7FF00000:
            if math.isnan(x):
                return x
            return x if xsb == 0 else 0.0
        if x > 7.09782712893383973096e+02:
            return math.inf
        if x < -7.45133219101941108420e+02:
            return 0.0
    hi = lo = 0.0
    k = 0
    if hx > 0x3FD62E42:
        if hx < 0x3FF0A2B2:
            hi = x - (-_LN2HI if xsb else _LN2HI)
            lo = -_LN2LO if xsb else _LN2LO
            k = 1 - xsb - xsb
        else:
            k = int(_INVLN2 * x + (-0.5 if xsb else 0.5))
            t = float(k)
            hi = x - t * _LN2HI
            lo = t * _LN2LO
        x = hi - lo
    elif hx < 0x3E300000:
        return 1.0 + x
    t = x * x
    c = x - t * (_P[0] + t * (_P[1] + t * (_P[2] + t * (_P[3] + t * _P[4]))))
    if k == 0:
        return 1.0 - ((x * c) / (c - 2.0) - x)
    y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi)
    if k >= -1021:
        return _from_bits(_bits(y) + (k << 52))
    return _from_bits(_bits(y) + ((k + 1000) << 52)) * 2.0 ** -1000


# ----------------------------------------------------------------------------
# metric/DCGScorer.java, metric/NDCGScorer.java
# ----------------------------------------------------------------------------
def discount(i):  # DCGScorer.java:26 ; SimpleMath.java:24-26
    return 1.0 / (math.log(i + 2) / math.log(2))


def gain(rel):  # DCGScorer.java:28-31
    return float((1 << rel) - 1)


def stable_desc(scores):
    """MergeSorter.sort(double[], false): stable, descending (MergeSorter.java:134-217)."""
    return sorted(range(len(scores)), key=lambda i: -scores[i]) if all(s == s for s in scores) else None


def ideal_dcg(rel, topk):  # NDCGScorer.java:167-174
    r = sorted(rel, reverse=True)
    dcg = 0.0
    for i in range(topk):
        dcg += gain(r[i]) * discount(i)
    return dcg


class NDCG:
    def __init__(self, k=10):
        self.k = k
        self.ideal_gains = {}  # NDCGScorer.java:32

    def score(self, lab_ranked, qid):  # NDCGScorer.java:103-129
        rel_ranked = [int(l) for l in lab_ranked]  # MetricScorer.getRelevanceLabels :54-60
        n = len(rel_ranked)
        if n == 0:
            return 0.0
        size = self.k
        if self.k > n or self.k <= 0:
            size = n
        if qid in self.ideal_gains:
            ideal = self.ideal_gains[qid]
        else:
            ideal = ideal_dcg(rel_ranked, size)
            self.ideal_gains[qid] = ideal
        if ideal <= 0.0:
            return 0.0
        dcg = 0.0
        for i in range(size):
            dcg += gain(rel_ranked[i]) * discount(i)
        return dcg / ideal

    def swap_change(self, lab_ranked, qid):  # NDCGScorer.java:132-160
        rel_ranked = [int(l) for l in lab_ranked]
        n = len(rel_ranked)
        size = self.k if n > self.k else n
        ideal = self.ideal_gains.get(qid)
        if ideal is None:
            ideal = ideal_dcg(rel_ranked, size)
        changes = [[0.0] * n for _ in range(n)]
        for i in range(size):
            for j in range(i + 1, n):
                if ideal > 0:
                    v = (discount(i) - discount(j)) * (gain(rel_ranked[i]) - gain(rel_ranked[j])) / ideal
                    changes[i][j] = v
                    changes[j][i] = v
        return changes


class DCG:  # metric/DCGScorer.java
    def __init__(self, k=10):
        self.k = k

    def score(self, lab_ranked, qid):  # :58-71
        n = len(lab_ranked)
        if n == 0:
            return 0.0
        size = n if (self.k > n or self.k <= 0) else self.k
        dcg = 0.0
        for i in range(size):
            dcg += gain(int(lab_ranked[i])) * discount(i)
        return dcg

    def swap_change(self, lab_ranked, qid):  # :74-90
        n = len(lab_ranked)
        rel = [int(l) for l in lab_ranked]
        size = self.k if n > self.k else n
        changes = [[0.0] * n for _ in range(n)]
        for i in range(size):
            for j in range(i + 1, n):
                v = (discount(i) - discount(j)) * (gain(rel[i]) - gain(rel[j]))
                changes[i][j] = changes[j][i] = v
        return changes


class MAP:  # metric/APScorer.java  (k = 0 unless "MAP@k"; the score ignores k)
    def __init__(self, k=0):
        self.k = k

    def score(self, lab_ranked, qid):  # :73-100
        ap, count = 0.0, 0
        for i, l in enumerate(lab_ranked):
            if l > 0.0:
                count += 1
                ap += count / (i + 1)
        return 0.0 if count == 0 else ap / count

    def swap_change(self, lab_ranked, qid):  # :108-162
        n = len(lab_ranked)
        labels, rel_count, count = [], [], 0
        for l in lab_ranked:
            labels.append(1 if l > 0 else 0)
            count += labels[-1]
            rel_count.append(count)
        changes = [[0.0] * n for _ in range(n)]
        if count == 0:
            return changes
        for i in range(n - 1):
            for j in range(i + 1, n):
                change = 0.0
                if labels[i] != labels[j]:
                    diff = labels[j] - labels[i]
                    change += float((rel_count[i] + diff) * labels[j] - rel_count[i] * labels[i]) / (i + 1)
                    for k in range(i + 1, j):
                        if labels[k] > 0:
                            change += float(diff) / (k + 1)
                    change += float(-rel_count[j] * diff) / (j + 1)
                changes[i][j] = changes[j][i] = change / count
        return changes


class ERR:  # metric/ERRScorer.java
    MAX = 16.0

    def __init__(self, k=10):
        self.k = k

    def R(self, rel):  # :71-73
        return ((1 << rel) - 1) / self.MAX

    def score(self, lab_ranked, qid):  # :45-64
        n = len(lab_ranked)
        size = n if (self.k > n or self.k <= 0) else self.k
        s, p = 0.0, 1.0
        for i in range(1, size + 1):
            r = self.R(int(lab_ranked[i - 1]))
            s += p * r / i
            p *= (1.0 - r)
        return s

    def swap_change(self, lab_ranked, qid):  # :76-115 (arrays beyond `size` stay 0; np is p * (1 - R) with p *= np)
        n = len(lab_ranked)
        size = self.k if n > self.k else n
        labels, R, npp = [0] * n, [0.0] * n, [0.0] * n
        p = 1.0
        for i in range(size):
            labels[i] = int(lab_ranked[i])
            R[i] = self.R(labels[i])
            npp[i] = p * (1.0 - R[i])
            p *= npp[i]
        changes = [[0.0] * n for _ in range(n)]
        for i in range(size):
            base = 1 if i == 0 else npp[i - 1]
            v1 = 1.0 / (i + 1) * base
            for j in range(i + 1, n):
                if labels[i] == labels[j]:
                    change = 0.0
                else:
                    change = v1 * (R[j] - R[i])
                    p = base * (R[i] - R[j])
                    for k in range(i + 1, j):
                        change += p * R[k] / (1 + k)
                        p *= 1.0 - R[k]
                    with np.errstate(all="ignore"):
                        change += float((np.float64(npp[j - 1]) * (1.0 - R[j]) * R[i] / np.float64(1.0 - R[i]) - npp[j - 1] * R[j]) / (j + 1))
                changes[i][j] = changes[j][i] = change
        return changes


SCORERS = {"NDCG": NDCG, "DCG": DCG, "MAP": MAP, "ERR": ERR}

# ----------------------------------------------------------------------------
# learning/tree/*
# ----------------------------------------------------------------------------
class Hist:  # FeatureHistogram.java:36-44
    def __init__(self):
        self.sum = None
        self.count = None
        self.sum_response = 0.0
        self.sq_sum_response = 0.0


class Split:  # Split.java:22-38
    def __init__(self, samples, hist, deviance):
        self.feature_id = -1
        self.threshold = F32(0)
        self.output = 0.0
        self.is_root = False
        self.deviance = deviance
        self.samples = samples
        self.hist = hist
        self.left = None
        self.right = None
        self.n = len(samples)
        self.trace = None

    def leaves(self):  # Split.java:100-113
        if self.feature_id == -1:
            return [self]
        return self.left.leaves() + self.right.leaves()

    def eval(self, row, fid2col):  # Split.java:115-125
        n = self
        while n.feature_id != -1:
            if F32(row[fid2col[n.feature_id]]) <= n.threshold:
                n = n.left
            else:
                n = n.right
        return n.output


class LambdaMART:
    def __init__(self, X, labels, qoff, n_trees=5, n_leaves=10, lr=0.1, n_threshold=256, mls=1, k=10,
                 early_stop=100, feature_ids=None, qids=None, metric="NDCG", ranker="LAMBDAMART"):
        self.X = np.asarray(X, dtype=F32)
        self.N, self.F = self.X.shape
        self.labels = np.asarray(labels, dtype=F32)
        self.qoff = list(qoff)
        self.Q = len(self.qoff) - 1
        self.n_trees, self.n_leaves, self.lr = n_trees, n_leaves, F32(lr)
        self.n_threshold, self.mls, self.early_stop = n_threshold, mls, early_stop
        self.scorer = SCORERS[metric](k)
        self.ranker = ranker
        self.features = list(feature_ids) if feature_ids is not None else list(range(1, self.F + 1))
        self.qids = list(qids) if qids is not None else ["q%d" % i for i in range(self.Q)]
        self.valid = None
        self.best_model_on_validation = 2147483647 - 2  # LambdaMART.java:50
        self.best_score_on_validation = 0.0
        self.ensemble = []
        self.splits_trace = []

    def set_validation(self, X, labels, qoff, qids=None):
        Q = len(qoff) - 1
        self.valid = dict(X=np.asarray(X, dtype=F32), labels=np.asarray(labels, dtype=F32), qoff=list(qoff),
                          qids=list(qids) if qids is not None else ["v%d" % i for i in range(Q)])

    # ---- init()  LambdaMART.java:68-166 ----
    def init(self):
        N, F = self.N, self.F
        self.model_scores = [0.0] * N
        self.pseudo = [0.0] * N
        self.weights = [0.0] * N
        self.thresholds = []
        self.bins = []
        root = Hist()
        root.sum, root.count = [], []
        for f in range(F):
            col = self.X[:, f]
            order = sorted(range(N), key=lambda i: float(col[i]))  # stable ascending  :417-424
            values = []
            for i in order:  # distinct values in ascending order :114-133
                if not values or col[i] > values[-1]:
                    values.append(col[i])
            fmin, fmax = values[0], values[-1]
            if len(values) <= self.n_threshold or self.n_threshold == -1:  # :135-140
                thr = list(values) + [FLT_MAX]
            else:  # :141-149 (float arithmetic)
                step = F32(abs(F32(fmax - fmin))) / F32(self.n_threshold)
                thr = [fmin]
                for j in range(1, self.n_threshold):
                    thr.append(F32(thr[-1] + step))
                thr.append(FLT_MAX)
            self.thresholds.append(thr)
            # FeatureHistogram.construct  FeatureHistogram.java:75-112
            st = [0] * N
            c = [0] * len(thr)
            last = -1
            for t, th in enumerate(thr):
                j = last + 1
                while j < N:
                    kdoc = order[j]
                    if col[kdoc] > th:
                        break
                    st[kdoc] = t
                    j += 1
                last = j - 1
                c[t] = last + 1
            self.bins.append(st)
            root.sum.append([0.0] * len(thr))
            root.count.append(c)
        self.root_hist = root
        if self.valid is not None:
            self.valid_scores = [0.0] * len(self.valid["labels"])

    # ---- computePseudoResponses  LambdaMART.java:331-396 ----
    def compute_lambdas(self):
        N = self.N
        if self.ranker == "MART":  # learning/tree/MART.java:47-51 (weights stay untouched)
            self.pseudo = [float(self.labels[i]) - self.model_scores[i] for i in range(N)]
            self.weights = [0.0] * N
            return
        self.pseudo = [0.0] * N
        self.weights = [0.0] * N
        cutoff = self.scorer.k
        for q in range(self.Q):
            cur, end = self.qoff[q], self.qoff[q + 1]
            n = end - cur
            local = self.model_scores[cur:end]
            idx = [cur + i for i in stable_desc(local)]
            lab = [float(self.labels[i]) for i in idx]
            changes = self.scorer.swap_change(lab, self.qids[q])
            for j in range(n):
                mj = idx[j]
                for k in range(n):
                    if j > cutoff and k > cutoff:
                        break
                    mk = idx[k]
                    if self.labels[mj] > self.labels[mk]:
                        delta_ndcg = abs(changes[j][k])
                        if delta_ndcg > 0:
                            rho = 1.0 / (1 + jexp(self.model_scores[mj] - self.model_scores[mk]))
                            lam = rho * delta_ndcg
                            self.pseudo[mj] += lam
                            self.pseudo[mk] -= lam
                            delta = rho * (1.0 - rho) * delta_ndcg
                            self.weights[mj] += delta
                            self.weights[mk] += delta

    # ---- FeatureHistogram.update  :114-146 ----
    def hist_update(self):
        h = self.root_hist
        h.sum_response = 0.0
        h.sq_sum_response = 0.0
        for f in range(self.F):
            s = [0.0] * len(self.thresholds[f])
            b = self.bins[f]
            for k in range(self.N):
                s[b[k]] += self.pseudo[k]
            for t in range(1, len(s)):
                s[t] += s[t - 1]
            h.sum[f] = s
        for k in range(self.N):
            h.sum_response += self.pseudo[k]
            h.sq_sum_response += self.pseudo[k] * self.pseudo[k]

    def _construct_left(self, soi):  # :148-195
        h = Hist()
        h.sum, h.count = [], []
        for f in range(self.F):
            T = len(self.thresholds[f])
            s, c = [0.0] * T, [0] * T
            b = self.bins[f]
            for k in soi:
                s[b[k]] += self.pseudo[k]
                c[b[k]] += 1
            for t in range(1, T):
                s[t] += s[t - 1]
                c[t] += c[t - 1]
            h.sum.append(s)
            h.count.append(c)
        for k in soi:
            h.sum_response += self.pseudo[k]
            h.sq_sum_response += self.pseudo[k] * self.pseudo[k]
        return h

    def _split(self, sp):  # FeatureHistogram.findBestSplit  :266-359
        h = sp.hist
        if sp.deviance >= 0.0 and sp.deviance <= 0.0:
            return False
        bestS, bf, bt = -1.0, -1, -1
        total = h.count[0][-1]
        for f in range(self.F):  # :236-264
            for t in range(len(self.thresholds[f])):
                cl = h.count[f][t]
                cr = total - cl
                if cl < self.mls or cr < self.mls:
                    continue
                sl = h.sum[f][t]
                sr = h.sum_response - sl
                S = sl * sl / cl + sr * sr / cr
                if bestS < S:
                    bestS, bf, bt = S, f, t
        if bestS == -1:
            return False
        left = [k for k in sp.samples if self.bins[bf][k] <= bt]
        right = [k for k in sp.samples if not self.bins[bf][k] <= bt]
        lh = self._construct_left(left)
        rh = Hist()  # :197-234
        rh.sum_response = h.sum_response - lh.sum_response
        rh.sq_sum_response = h.sq_sum_response - lh.sq_sum_response
        rh.sum = [[a - b for a, b in zip(h.sum[f], lh.sum[f])] for f in range(self.F)]
        rh.count = [[a - b for a, b in zip(h.count[f], lh.count[f])] for f in range(self.F)]
        var = h.sq_sum_response - h.sum_response * h.sum_response / len(sp.samples)
        var_l = lh.sq_sum_response - lh.sum_response * lh.sum_response / len(left)
        var_r = rh.sq_sum_response - rh.sum_response * rh.sum_response / len(right)
        sp.feature_id = self.features[bf]
        sp.threshold = F32(self.thresholds[bf][bt])
        sp.deviance = var
        sp.left = Split(left, lh, var_l)
        sp.right = Split(right, rh, var_r)
        self.splits_trace.append((bf, bt, bestS, len(sp.samples), len(left)))
        sp.samples = None
        sp.hist = None
        return True

    @staticmethod
    def _insert(queue, s):  # RegressionTree.java:147-157
        i = 0
        while i < len(queue):
            if queue[i].deviance > s.deviance:
                i += 1
            else:
                break
        queue.insert(i, s)

    def fit_tree(self):  # RegressionTree.java:58-87
        self.splits_trace = []
        root = Split(list(range(self.N)), self.root_hist, float(FLT_MAX))
        root.is_root = True
        queue = []
        if self._split(root):
            self._insert(queue, root.left)
            self._insert(queue, root.right)
        taken = 0
        while (self.n_leaves == -1 or taken + len(queue) < self.n_leaves) and len(queue) > 0:
            leaf = queue.pop(0)
            if leaf.n < 2 * self.mls:
                taken += 1
                continue
            if not self._split(leaf):
                taken += 1
            else:
                self._insert(queue, leaf.left)
                self._insert(queue, leaf.right)
        return root

    def _metric(self, scores, labels, qoff, qids):  # LambdaMART.java:442-483 / :485-518
        s = F32(0)
        Q = len(qoff) - 1
        for q in range(Q):
            cur, end = qoff[q], qoff[q + 1]
            order = stable_desc(scores[cur:end])
            lab = [float(labels[cur + i]) for i in order]
            s = F32(float(s) + self.scorer.score(lab, qids[q]))
        return F32(s / F32(Q))

    def round(self):  # one iteration of LambdaMART.java:180-251
        m = len(self.ensemble)  # (== m while no early stop happened)
        self.compute_lambdas()
        self.hist_update()
        root = self.fit_tree()
        leaves = root.leaves()
        if root.feature_id == -1:
            root.samples = list(range(self.N))
        for lf in leaves:  # updateTreeOutput :398-415
            s1 = F32(0)
            s2 = F32(0)
            for k in lf.samples:
                s1 = F32(float(s1) + self.pseudo[k])
                s2 = F32(float(s2) + self.weights[k])
            if self.ranker == "MART":  # MART.java:54-65: float sum / int count
                lf.output = float(F32(s1 / F32(len(lf.samples))))
            else:
                lf.output = 0.0 if s2 == 0 else float(F32(s1 / s2))
        for lf in leaves:  # :203-210
            for k in lf.samples:
                self.model_scores[k] += float(self.lr) * lf.output
        self.ensemble.append(root)
        train_metric = self._metric(self.model_scores, self.labels, self.qoff, self.qids)
        valid_metric = None
        if self.valid is not None:
            fid2col = {fid: c for c, fid in enumerate(self.features)}
            v = self.valid
            for i in range(len(v["labels"])):
                self.valid_scores[i] += float(self.lr) * root.eval(v["X"][i], fid2col)
            valid_metric = self._metric(self.valid_scores, v["labels"], v["qoff"], v["qids"])
            if float(valid_metric) > self.best_score_on_validation:
                self.best_score_on_validation = float(valid_metric)
                self.best_model_on_validation = len(self.ensemble) - 1
        stop = (m - self.best_model_on_validation) > self.early_stop
        return root, train_metric, valid_metric, stop

    def predict(self, X):  # Ensemble.eval  Ensemble.java:110-116
        fid2col = {fid: c for c, fid in enumerate(self.features)}
        out = []
        for row in np.asarray(X, dtype=F32):
            s = F32(0)
            for tree in self.ensemble:
                s = F32(float(s) + tree.eval(row, fid2col) * float(self.lr))
            out.append(s)
        return np.array(out, dtype=F32)

    def finish(self):  # LambdaMART.java:254-259
        while len(self.ensemble) > self.best_model_on_validation + 1:
            self.ensemble.pop()
        sc = [float(v) for v in self.predict(self.X)]
        total = 0.0
        for q in range(self.Q):
            cur, end = self.qoff[q], self.qoff[q + 1]
            order = stable_desc(sc[cur:end])
            total += self.scorer.score([float(self.labels[cur + i]) for i in order], self.qids[q])
        return total / self.Q


def flatten_tree(root):
    """pre-order flattening identical to the oracle's ro_tree layout"""
    feat, thr, left, right, out = [], [], [], [], []

    def rec(n):
        me = len(feat)
        feat.append(n.feature_id)
        thr.append(float(n.threshold))
        out.append(float(n.output))
        left.append(-1)
        right.append(-1)
        if n.feature_id != -1:
            left[me] = rec(n.left)
            right[me] = rec(n.right)
        return me

    rec(root)
    return feat, thr, left, right, out
