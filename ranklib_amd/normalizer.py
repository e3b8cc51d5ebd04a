"""Feature normalisation of the reference's command line (`-norm sum | zscore | linear`, eval/Evaluator.java:256-267): every ranked list
is normalised on its own, in place, before training / scoring.  Mirrors features/SumNormalizor.java, ZScoreNormalizor.java and
LinearNormalizer.java operation by operation -- the sums are sequential f64 sums over the list in document order, the casts to float
happen where the Java's do -- because the normalised values are what LambdaMART.init() builds its thresholds from.

Host-side preprocessing (numpy), as the reference's is host-side Java; nothing here runs on the GPU."""
import numpy as np

from .learning import DataPoint, RankLibError


def _unique(fids):                    # features/Normalizer.java:31-44 (a HashSet: order is irrelevant, the features are independent)
    seen, out = set(), []
    for f in fids:
        if f not in seen:
            seen.add(f)
            out.append(int(f))
    return out


def _gather(rl, fids, who):
    """[documents, len(fids)] float32 through DataPoint.getFeatureValue (NaN = unknown -> 0; a feature past a row's end is an error
    unless -missingZero, learning/DenseDataPoint.java:21-32)"""
    if rl.size() == 0:
        raise RankLibError("Error in %s::normalize(): The input ranked list is empty" % who)
    if fids and min(fids) > 0 and all(len(dp.fVals) > max(fids) for dp in rl.rl):
        # every row names every requested feature: one fancy index per list instead of a Python visit per cell
        # (rows are ragged -- each ends at its own last feature id -- so every row is cut to the requested ids before the stack)
        return np.nan_to_num(np.stack([np.asarray(dp.fVals, np.float32)[fids] for dp in rl.rl]), nan=0.0, posinf=np.inf, neginf=-np.inf)
    M = np.zeros((rl.size(), len(fids)), np.float32)
    for i, dp in enumerate(rl.rl):
        fv = dp.fVals
        for j, f in enumerate(fids):
            if f <= 0 or f >= len(fv):
                if not DataPoint.missingZero:
                    raise RankLibError("Error in DenseDataPoint::getFeatureValue(): requesting unspecified feature, fid=%d" % f)
            elif not np.isnan(fv[f]):
                M[i, j] = fv[f]
    return M


def _scatter(rl, fids, M, mask):
    """DataPoint.setFeatureValue for the columns in `mask` (learning/DenseDataPoint.java:35-40: a feature past the row's end is an error)"""
    cols = [j for j in range(len(fids)) if mask[j]]
    if cols and min(fids[j] for j in cols) > 0 and all(len(dp.fVals) > max(fids[j] for j in cols) for dp in rl.rl):
        ids = [fids[j] for j in cols]
        for i, dp in enumerate(rl.rl):
            dp.fVals[ids] = M[i, cols]
        return
    for i, dp in enumerate(rl.rl):
        fv = dp.fVals
        for j, f in enumerate(fids):
            if mask[j]:
                if f <= 0 or f >= len(fv):
                    raise RankLibError("Error in DenseDataPoint::setFeatureValue(): feature (id=%d) not found." % f)
                fv[f] = M[i, j]


def _seq_sum(A):                      # `acc += a[i]` over the documents, in order, in f64 (np.sum would add pairwise)
    return np.add.accumulate(A.astype(np.float64), axis=0)[-1]


class Normalizer:
    def normalize(self, rl, fids=None):
        raise NotImplementedError

    def normalizeAll(self, samples, fids=None):       # Normalizer.normalize(List<RankList>[, fids])
        for rl in samples:
            self.normalize(rl, fids)

    @staticmethod
    def _fids(rl, fids):
        return list(range(1, rl.getFeatureCount() + 1)) if fids is None else _unique(fids)


class SumNormalizor(Normalizer):      # features/SumNormalizor.java:18-70
    def name(self):
        return "sum"

    def normalize(self, rl, fids=None):
        fids = self._fids(rl, fids)
        M = _gather(rl, fids, "SumNormalizor")
        norm = _seq_sum(np.abs(M))                                   # norm[j] += Math.abs(value): double += float
        ok = norm > 0
        with np.errstate(divide="ignore", invalid="ignore"):
            out = (M.astype(np.float64) / norm).astype(np.float32)   # (float)(value / norm[j])
        _scatter(rl, fids, out, ok)


class ZScoreNormalizor(Normalizer):   # features/ZScoreNormalizor.java:18-92
    def name(self):
        return "zscore"

    def normalize(self, rl, fids=None):
        fids = self._fids(rl, fids)
        M = _gather(rl, fids, "ZScoreNormalizor").astype(np.float64)
        n = M.shape[0]
        means = _seq_sum(M) / n
        d = M - means                                                # x = value - mean (double)
        with np.errstate(divide="ignore", invalid="ignore"):
            std = np.sqrt(_seq_sum(d * d) / (n - 1))                 # a list of one document: 0 / 0 = NaN, `std > 0` is false
            out = (d / std).astype(np.float32)
        _scatter(rl, fids, out, std > 0)


class LinearNormalizer(Normalizer):   # features/LinearNormalizer.java:18-68
    def name(self):
        return "linear"

    def normalize(self, rl, fids=None):
        fids = self._fids(rl, fids)
        M = _gather(rl, fids, "LinearNormalizor")
        # min starts at Float.MAX_VALUE, max at Float.MIN_VALUE -- the smallest POSITIVE float (:38-39): a column without a positive value keeps it
        lo = np.minimum(M.min(axis=0), np.finfo(np.float32).max)
        hi = np.maximum(M.max(axis=0), np.float32(1.4e-45))
        span = hi - lo                                               # float arithmetic throughout (:52)
        ok = hi > lo
        with np.errstate(divide="ignore", invalid="ignore"):
            out = np.where(ok, (M - lo) / span, np.float32(0)).astype(np.float32)
        _scatter(rl, fids, out, np.ones(len(fids), bool))            # else: setFeatureValue(fid, 0)


def create(name):                     # eval/Evaluator.java:258-267
    n = name.lower()
    if n == "sum":
        return SumNormalizor()
    if n == "zscore":
        return ZScoreNormalizor()
    if n == "linear":
        return LinearNormalizer()
    raise RankLibError("Unknown normalizor: " + name)
