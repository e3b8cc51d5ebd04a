"""ctypes binding of the CPU oracle (oracle/librl_oracle.so) -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "librl_oracle.so")


class RoParams(C.Structure):
    _fields_ = [("n_trees", C.c_int32), ("n_leaves", C.c_int32), ("n_threshold", C.c_int32),
                ("min_leaf_support", C.c_int32), ("early_stop", C.c_int32), ("learning_rate", C.c_float),
                ("metric_k", C.c_int32), ("n_threads", C.c_int32), ("ranker", C.c_int32), ("metric", C.c_int32),
                ("feature_sampling_rate", C.c_float), ("seed", C.c_uint64)]


RANKER = dict(MART=0, LAMBDAMART=6)
METRIC = dict(NDCG=0, DCG=1, MAP=2, ERR=3)


class RoTree(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("cap", C.c_int32), ("feature", C.POINTER(C.c_int32)),
                ("threshold", C.POINTER(C.c_float)), ("left", C.POINTER(C.c_int32)),
                ("right", C.POINTER(C.c_int32)), ("output", C.POINTER(C.c_float)),
                ("deviance", C.POINTER(C.c_double)), ("count", C.POINTER(C.c_int32))]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def build():
    src = os.path.join(_ORACLE_DIR, "rl_oracle.c")
    if (not os.path.exists(_SO)) or (os.path.exists(src) and os.path.getmtime(_SO) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.ro_create.restype = C.c_void_p
        L.ro_create.argtypes = [C.POINTER(RoParams), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                C.c_int32, C.c_void_p, C.c_void_p]
        L.ro_set_validation.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_void_p]
        L.ro_destroy.argtypes = [C.c_void_p]
        L.ro_root_hash.restype = C.c_uint64; L.ro_root_hash.argtypes = [C.c_uint64, C.c_int32]
        L.ro_child_hash.restype = C.c_uint64; L.ro_child_hash.argtypes = [C.c_uint64, C.c_int32]
        L.ro_feature_key.restype = C.c_uint64; L.ro_feature_key.argtypes = [C.c_uint64, C.c_int32]
        L.ro_feature_order.restype = C.c_int32; L.ro_feature_order.argtypes = [C.c_uint64, C.c_int32, C.c_float, C.c_void_p]
        L.ro_init.argtypes = [C.c_void_p]
        L.ro_round.restype = C.c_int
        L.ro_round.argtypes = [C.c_void_p, C.POINTER(RoTree), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ro_compute_lambdas.argtypes = [C.c_void_p]
        L.ro_hist_update_only.argtypes = [C.c_void_p]
        L.ro_n_bins.restype = C.c_int32
        L.ro_n_bins.argtypes = [C.c_void_p, C.c_int32]
        for name, rt in (("ro_thresholds", C.POINTER(C.c_float)), ("ro_bins", C.POINTER(C.c_int32)),
                         ("ro_root_count", C.POINTER(C.c_int32)), ("ro_root_sum", C.POINTER(C.c_double))):
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [C.c_void_p, C.c_int32]
        for name in ("ro_lambdas", "ro_weights", "ro_scores", "ro_valid_scores"):
            getattr(L, name).restype = C.POINTER(C.c_double)
            getattr(L, name).argtypes = [C.c_void_p]
        L.ro_trees_kept.restype = C.c_int32
        L.ro_trees_kept.argtypes = [C.c_void_p]
        L.ro_best_valid_round.restype = C.c_int32
        L.ro_best_valid_round.argtypes = [C.c_void_p]
        L.ro_best_valid_score.restype = C.c_double
        L.ro_best_valid_score.argtypes = [C.c_void_p]
        L.ro_last_split_trace.restype = C.c_int32
        L.ro_last_split_trace.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 5
        L.ro_finish_with_rows.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ro_get_tree.restype = C.c_int32
        L.ro_get_tree.argtypes = [C.c_void_p, C.c_int32, C.POINTER(RoTree)]
        L.ro_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.ro_exp.restype = C.c_double
        L.ro_exp.argtypes = [C.c_double]
        L.ro_discount.restype = C.c_double
        L.ro_discount.argtypes = [C.c_int32]
        L.ro_sort_desc.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.ro_query_lambdas.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p,
                                       C.c_void_p]
        L.ro_query_ndcg.restype = C.c_double
        L.ro_query_ndcg.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double]
        L.ro_query_lambdas_metric.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.ro_query_score.restype = C.c_double
        L.ro_query_score.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.ro_eval_flat_model.argtypes = [C.c_int32, C.c_int32] + [C.c_void_p] * 7 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
        L.ro_set_err_max.argtypes = [C.c_double]
        L.ro_set_external.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ro_float_chain.restype = C.c_float
        L.ro_float_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        _lib = L
    return _lib


class Tree:
    """numpy-backed flat tree in the oracle's pre-order layout"""

    def __init__(self, cap):
        self.cap = cap
        self.feature = np.full(cap, -1, np.int32)
        self.threshold = np.zeros(cap, np.float32)
        self.left = np.full(cap, -1, np.int32)
        self.right = np.full(cap, -1, np.int32)
        self.output = np.zeros(cap, np.float32)
        self.deviance = np.zeros(cap, np.float64)
        self.count = np.zeros(cap, np.int32)
        self.n_nodes = 0

    def c(self):
        return RoTree(0, self.cap, _p(self.feature, C.c_int32), _p(self.threshold, C.c_float),
                      _p(self.left, C.c_int32), _p(self.right, C.c_int32), _p(self.output, C.c_float),
                      _p(self.deviance, C.c_double), _p(self.count, C.c_int32))

    def trimmed(self):
        n = self.n_nodes
        return dict(feature=self.feature[:n].copy(), threshold=self.threshold[:n].copy(), left=self.left[:n].copy(),
                    right=self.right[:n].copy(), output=self.output[:n].copy(), deviance=self.deviance[:n].copy(),
                    count=self.count[:n].copy())


class Oracle:
    def __init__(self, X, labels, qoff, n_trees=10, n_leaves=10, lr=0.1, n_threshold=256, mls=1, k=10,
                 early_stop=100, n_threads=1, feature_ids=None, qkey=None, max_nodes=None, ranker="LAMBDAMART",
                 metric="NDCG", frate=1.0, seed=0):
        self.L = lib()
        self.X = np.ascontiguousarray(X, dtype=np.float32)
        self.labels = np.ascontiguousarray(labels, dtype=np.float32)
        self.qoff = np.ascontiguousarray(qoff, dtype=np.int32)
        self.N, self.F = self.X.shape
        self.Q = len(self.qoff) - 1
        self.p = RoParams(n_trees, n_leaves, n_threshold, mls, early_stop, lr, k, n_threads, RANKER[ranker], METRIC[metric], frate, seed)
        fid = None if feature_ids is None else np.ascontiguousarray(feature_ids, dtype=np.int32)
        qk = None if qkey is None else np.ascontiguousarray(qkey, dtype=np.int32)
        self._keep = (fid, qk)
        self.h = self.L.ro_create(C.byref(self.p), self.X.ctypes.data, self.N, self.F, self.labels.ctypes.data,
                                  self.qoff.ctypes.data, self.Q, None if fid is None else fid.ctypes.data,
                                  None if qk is None else qk.ctypes.data)
        self.cap = max_nodes or (max(3, 2 * n_leaves - 1) if n_leaves > 0 else 2 * self.N + 1)
        self.has_valid = False

    def set_validation(self, X, labels, qoff, qkey=None):
        Xv = np.ascontiguousarray(X, dtype=np.float32)
        lv = np.ascontiguousarray(labels, dtype=np.float32)
        qv = np.ascontiguousarray(qoff, dtype=np.int32)
        qk = None if qkey is None else np.ascontiguousarray(qkey, dtype=np.int32)
        self.L.ro_set_validation(self.h, Xv.ctypes.data, Xv.shape[0], lv.ctypes.data, qv.ctypes.data, len(qv) - 1,
                                 None if qk is None else qk.ctypes.data)
        self.nv = Xv.shape[0]
        self.has_valid = True

    def set_external(self, validation, ideal=None, rel_count=None):
        idl = None if ideal is None else np.ascontiguousarray(ideal, dtype=np.float64)
        rdc = None if rel_count is None else np.ascontiguousarray(rel_count, dtype=np.int32)
        self.L.ro_set_external(self.h, 1 if validation else 0, None if idl is None else idl.ctypes.data, None if rdc is None else rdc.ctypes.data)

    def init(self):
        self.L.ro_init(self.h)

    def round(self):
        t = Tree(self.cap)
        ct = t.c()
        tm, vm = C.c_float(0), C.c_float(0)
        stop = self.L.ro_round(self.h, C.byref(ct), C.byref(tm), C.byref(vm))
        t.n_nodes = ct.n_nodes
        return t, np.float32(tm.value), (np.float32(vm.value) if self.has_valid else None), bool(stop)

    def compute_lambdas(self):
        self.L.ro_compute_lambdas(self.h)

    def hist_update(self):
        self.L.ro_hist_update_only(self.h)

    def n_bins(self, f):
        return self.L.ro_n_bins(self.h, f)

    def thresholds(self, f):
        return np.ctypeslib.as_array(self.L.ro_thresholds(self.h, f), (self.n_bins(f),)).copy()

    def bins(self, f):
        return np.ctypeslib.as_array(self.L.ro_bins(self.h, f), (self.N,)).copy()

    def root_count(self, f):
        return np.ctypeslib.as_array(self.L.ro_root_count(self.h, f), (self.n_bins(f),)).copy()

    def root_sum(self, f):
        return np.ctypeslib.as_array(self.L.ro_root_sum(self.h, f), (self.n_bins(f),)).copy()

    def _vec(self, fn, n):
        return np.ctypeslib.as_array(fn(self.h), (n,)).copy()

    def lambdas(self):
        return self._vec(self.L.ro_lambdas, self.N)

    def weights(self):
        return self._vec(self.L.ro_weights, self.N)

    def scores(self):
        return self._vec(self.L.ro_scores, self.N)

    def valid_scores(self):
        return self._vec(self.L.ro_valid_scores, self.nv)

    def split_trace(self):
        cap = max(self.cap, 1)
        f = np.zeros(cap, np.int32)
        t = np.zeros(cap, np.int32)
        S = np.zeros(cap, np.float64)
        nn = np.zeros(cap, np.int32)
        nl = np.zeros(cap, np.int32)
        n = self.L.ro_last_split_trace(self.h, cap, f.ctypes.data, t.ctypes.data, S.ctypes.data, nn.ctypes.data,
                                       nl.ctypes.data)
        return [(int(f[i]), int(t[i]), float(S[i]), int(nn[i]), int(nl[i])) for i in range(n)]

    def finish(self):
        ts, vs = C.c_double(0), C.c_double(0)
        self.L.ro_finish_with_rows(self.h, self.X.ctypes.data, C.byref(ts), C.byref(vs))
        return ts.value, (vs.value if self.has_valid else None)

    def trees_kept(self):
        return self.L.ro_trees_kept(self.h)

    def best_valid(self):
        return self.L.ro_best_valid_round(self.h), self.L.ro_best_valid_score(self.h)

    def get_tree(self, i):
        t = Tree(self.cap)
        ct = t.c()
        n = self.L.ro_get_tree(self.h, i, C.byref(ct))
        t.n_nodes = n
        return t

    def predict(self, X):
        X = np.ascontiguousarray(X, dtype=np.float32)
        out = np.zeros(X.shape[0], np.float32)
        self.L.ro_predict(self.h, X.ctypes.data, X.shape[0], out.ctypes.data)
        return out

    def close(self):
        if self.h:
            self.L.ro_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def query_lambdas(scores, labels, k=10, ideal=-1.0):
    s = np.ascontiguousarray(scores, np.float64)
    l = np.ascontiguousarray(labels, np.float32)
    lam = np.zeros(len(s))
    w = np.zeros(len(s))
    lib().ro_query_lambdas(s.ctypes.data, l.ctypes.data, len(s), k, ideal, lam.ctypes.data, w.ctypes.data)
    return lam, w


def query_lambdas_metric(metric, scores, labels, k):
    s = np.ascontiguousarray(scores, np.float64)
    l = np.ascontiguousarray(labels, np.float32)
    lam = np.zeros(len(s))
    w = np.zeros(len(s))
    lib().ro_query_lambdas_metric(METRIC[metric], s.ctypes.data, l.ctypes.data, len(s), k, lam.ctypes.data, w.ctypes.data)
    return lam, w


def query_score(metric, scores, labels, k):
    s = np.ascontiguousarray(scores, np.float64)
    l = np.ascontiguousarray(labels, np.float32)
    return lib().ro_query_score(METRIC[metric], s.ctypes.data, l.ctypes.data, len(s), k)


def query_ndcg(scores, labels, k=10, ideal=-1.0):
    s = np.ascontiguousarray(scores, np.float64)
    l = np.ascontiguousarray(labels, np.float32)
    return lib().ro_query_ndcg(s.ctypes.data, l.ctypes.data, len(s), k, ideal)


def sort_desc(scores):
    s = np.ascontiguousarray(scores, np.float64)
    idx = np.zeros(len(s), np.int32)
    lib().ro_sort_desc(s.ctypes.data, len(s), idx.ctypes.data)
    return idx


def feature_order(node_hash, n_features, rate):
    out = np.zeros(n_features, np.int32)
    n = lib().ro_feature_order(node_hash, n_features, rate, out.ctypes.data)
    return out[:n]


def set_err_max(m):
    lib().ro_set_err_max(float(m))


def float_chain(x, idx=None):
    x = np.ascontiguousarray(x, np.float64)
    if idx is None:
        return np.float32(lib().ro_float_chain(x.ctypes.data, None, len(x)))
    idx = np.ascontiguousarray(idx, np.int32)
    return np.float32(lib().ro_float_chain(x.ctypes.data, idx.ctypes.data, len(idx)))


def eval_flat_model(trees, rows, n_threads=1, weight=0.1):
    """trees: list of dicts (feature ids, threshold, left, right, output) in any node order with root 0; rows[:, f] = feature f"""
    nt = len(trees)
    maxn = max(len(t["feature"]) for t in trees)
    feat = np.full((nt, maxn), -1, np.int32); left = np.zeros((nt, maxn), np.int32); right = np.zeros((nt, maxn), np.int32)
    thr = np.zeros((nt, maxn), np.float32); outv = np.zeros((nt, maxn), np.float32)
    for i, t in enumerate(trees):
        n = len(t["feature"])
        feat[i, :n] = t["feature"]; left[i, :n] = t["left"]; right[i, :n] = t["right"]; thr[i, :n] = t["threshold"]; outv[i, :n] = t["output"]
    w = np.full(nt, weight, np.float32)
    rows = np.ascontiguousarray(rows, np.float32)
    res = np.zeros(rows.shape[0], np.float32)
    lib().ro_eval_flat_model(nt, maxn, feat.ctypes.data, thr.ctypes.data, left.ctypes.data, right.ctypes.data, outv.ctypes.data,
                             w.ctypes.data, rows.ctypes.data, rows.shape[0], rows.shape[1], n_threads, res.ctypes.data)
    return res
