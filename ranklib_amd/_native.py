"""ctypes binding of librlhip.so (include/rlhip.h).

The library is built in-tree (ranklib_amd/lib/librlhip.so) by ranklib_amd/csrc/Makefile.
There is no fallback: if the shared object is missing or no gfx950 device is visible the calls raise
RankLibError, exactly like the reference's unchecked error convention (utilities/RankLibError.java).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RLHIP_LIB") or os.path.join(_HERE, "lib", "librlhip.so")     # RLHIP_LIB: A/B builds (tools/)


class RankLibError(RuntimeError):
    """Mirrors ciir.umass.edu.utilities.RankLibError (utilities/RankLibError.java:9-43)."""


class RlParams(C.Structure):
    _fields_ = [("n_trees", C.c_int32), ("n_leaves", C.c_int32), ("n_threshold", C.c_int32),
                ("min_leaf_support", C.c_int32), ("early_stop_rounds", C.c_int32), ("learning_rate", C.c_float),
                ("metric", C.c_int32), ("metric_k", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32),
                ("ranker", C.c_int32), ("feature_sampling_rate", C.c_float), ("seed", C.c_uint64)]


RL_METRIC = dict(NDCG=0, DCG=1, MAP=2, ERR=3)
RL_RANKER = dict(MART=0, LAMBDAMART=6)


class RlTree(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("cap", C.c_int32), ("feature", C.POINTER(C.c_int32)),
                ("threshold", C.POINTER(C.c_float)), ("left", C.POINTER(C.c_int32)),
                ("right", C.POINTER(C.c_int32)), ("output", C.POINTER(C.c_float)),
                ("deviance", C.POINTER(C.c_double)), ("count", C.POINTER(C.c_int32))]


RL_FLAG_TIMING, RL_FLAG_SERIAL_CHAIN, RL_FLAG_TIMING_NODES, RL_FLAG_JAVA_ORDER, RL_FLAG_FIRST_TIE = 2, 4, 8, 16, 32
ARR = dict(LAMBDA=1, WEIGHT=2, SCORE=3, VALID_SCORE=4, NBINS=5, THRESHOLDS=6, BINS=7, ROOT_COUNT=8, ROOT_SUM=9,
           QUANT=10, ROOT_SUM_FIXED=11, NDCG_PER_QUERY=12, CHAIN_STATS=13, CHAIN_MISS=14, GROW_STATS=15, PHASE_CLOCKS=16,
           ROOT_SUM_JAVA=17, GROW_DOCS=18, SPARSE_INFO=19, STEP_LOG=20, TIE_STATS=21, BLOCK_TRACE=22, BUBBLES=23, PIECE_STATS=24)
KERNEL = dict(HIST_ROOT=0, HIST_NODE=1, LAMBDA=2)

# every symbol include/rlhip.h declares (tests/test_abi.py checks the .so exports all of them)
ABI_SYMBOLS = [
    "rl_abi_version", "rl_last_error", "rl_device_count", "rl_params_default", "rl_create", "rl_destroy",
    "rl_set_train", "rl_set_validation", "rl_set_rows", "rl_set_external_judgments", "rl_init", "rl_boost_round", "rl_boost_rounds_async", "rl_sync",
    "rl_finish", "rl_num_trees", "rl_get_tree", "rl_get_round_metrics", "rl_best_validation", "rl_predict",
    "rl_model_to_text", "rl_model_from_text", "rl_model_destroy", "rl_model_num_trees", "rl_model_features",
    "rl_model_predict", "rl_model_predict_device", "rl_dist_unique_id", "rl_dist_init", "rl_dist_init_callback", "rl_dist_stats", "rl_bin_stride", "rl_hist_features", "rl_quant_exponent", "rl_get_array", "rl_debug_exp", "rl_debug_rho", "rl_debug_float_chain",
    "rl_letor_parse", "rl_letor_info", "rl_letor_arrays", "rl_letor_rows", "rl_letor_destroy",
    "rl_get_timing", "rl_reset_timing", "rl_set_timing_flags", "rl_debug_membench", "rl_set_err_max", "rl_tree_capacity",
]

HOST_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32)
HOST_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
HOST_ALLTOALLV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64))
DT_NUMPY = {0: np.int64, 1: np.uint64, 2: np.int32, 3: np.uint32, 4: np.float64}

_lib = None


def lib():
    """Load librlhip.so; fail loudly if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RankLibError("librlhip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C ranklib_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32p = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_float)
    L.rl_abi_version.restype = C.c_int
    L.rl_last_error.restype = C.c_char_p
    L.rl_device_count.argtypes = [C.POINTER(i32)]
    L.rl_params_default.argtypes = [C.POINTER(RlParams)]
    L.rl_params_default.restype = None
    L.rl_create.argtypes = [C.POINTER(RlParams), C.POINTER(vp)]
    L.rl_destroy.argtypes = [vp]
    L.rl_destroy.restype = None
    L.rl_set_train.argtypes = [vp, vp, i64, i32, vp, vp, i32, vp, vp]
    L.rl_set_validation.argtypes = [vp, vp, i64, vp, vp, i32, vp]
    L.rl_set_rows.argtypes = [vp, i32, i64, i64, vp]
    L.rl_set_external_judgments.argtypes = [vp, i32, vp, vp]
    L.rl_init.argtypes = [vp]
    L.rl_boost_round.argtypes = [vp, C.POINTER(RlTree), f32p, f32p, C.POINTER(i32)]
    L.rl_boost_rounds_async.argtypes = [vp, i32]
    L.rl_sync.argtypes = [vp]
    L.rl_finish.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rl_num_trees.argtypes = [vp, C.POINTER(i32)]
    L.rl_tree_capacity.argtypes = [vp, C.POINTER(i32)]
    L.rl_get_tree.argtypes = [vp, i32, C.POINTER(RlTree)]
    L.rl_get_round_metrics.argtypes = [vp, i32, f32p, f32p]
    L.rl_best_validation.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_double)]
    L.rl_predict.argtypes = [vp, vp, i64, vp]
    L.rl_model_to_text.argtypes = [vp, C.c_char_p, i64, C.POINTER(i64)]
    L.rl_model_from_text.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.rl_model_destroy.argtypes = [vp]
    L.rl_model_destroy.restype = None
    L.rl_model_num_trees.argtypes = [vp, C.POINTER(i32)]
    L.rl_model_features.argtypes = [vp, vp, i32, C.POINTER(i32)]
    L.rl_model_predict.argtypes = [vp, vp, i64, i32, vp]
    L.rl_model_predict_device.argtypes = [vp, vp, i64, i32, vp, vp]
    L.rl_dist_unique_id.argtypes = [vp]
    L.rl_dist_init.argtypes = [vp, vp, i32, i32]
    L.rl_dist_stats.argtypes = [vp, vp]
    L.rl_dist_init_callback.argtypes = [vp, i32, i32, HOST_ALLREDUCE, HOST_ALLGATHER, HOST_ALLTOALLV, vp]
    L.rl_bin_stride.argtypes = [vp, C.POINTER(i32)]
    L.rl_hist_features.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), i32]
    L.rl_quant_exponent.argtypes = [vp, C.POINTER(i32)]
    L.rl_get_array.argtypes = [vp, i32, vp, i64]
    L.rl_debug_exp.argtypes = [vp, i32, vp, vp]
    if hasattr(L, "rl_debug_rho"):      # (A/B builds of older sources selected with RLHIP_LIB lack the probe; tests/test_abi.py checks the in-tree library's exports)
        L.rl_debug_rho.argtypes = [vp, vp, i32, vp, vp]
    L.rl_debug_float_chain.argtypes = [i32, vp, i64, vp, i32, vp, vp]
    L.rl_letor_parse.argtypes = [vp, i64, C.POINTER(vp)]
    L.rl_letor_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i32), C.POINTER(i64)]
    L.rl_letor_arrays.argtypes = [vp] * 10
    L.rl_letor_rows.argtypes = [vp, vp, i64]
    L.rl_letor_destroy.argtypes = [vp]
    L.rl_letor_destroy.restype = None
    L.rl_get_timing.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]
    L.rl_reset_timing.argtypes = [vp]
    L.rl_set_err_max.argtypes = [C.c_double]
    L.rl_set_timing_flags.argtypes = [vp, i32]
    L.rl_debug_membench.argtypes = [i32, i32, i64, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RankLibError("%s (rlhip status %d)" % (lib().rl_last_error().decode("utf-8", "replace"), rc))


def device_count():
    n = C.c_int32(0)
    rc = lib().rl_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class FlatTree:
    """One regression tree in pre-order (root 0, left subtree first)."""

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
        def p(a, t):
            return a.ctypes.data_as(C.POINTER(t))
        return RlTree(0, self.cap, p(self.feature, C.c_int32), p(self.threshold, C.c_float), p(self.left, C.c_int32),
                      p(self.right, C.c_int32), p(self.output, C.c_float), p(self.deviance, C.c_double),
                      p(self.count, C.c_int32))

    def trimmed(self):
        n = self.n_nodes
        return dict(feature=self.feature[:n].copy(), threshold=self.threshold[:n].copy(), left=self.left[:n].copy(),
                    right=self.right[:n].copy(), output=self.output[:n].copy(), deviance=self.deviance[:n].copy(),
                    count=self.count[:n].copy())


def debug_exp(x):
    x = np.ascontiguousarray(x, np.float64)
    a, b = np.zeros_like(x), np.zeros_like(x)
    check(lib().rl_debug_exp(x.ctypes.data, len(x), a.ctypes.data, b.ctypes.data))
    return a, b


def debug_rho(x, den=None):
    """rl_debug_rho: (fast, ref) of rho(x) = 1 / (1 + exp(x)), or of x / den when den is given"""
    x = np.ascontiguousarray(x, np.float64)
    if den is not None:
        den = np.ascontiguousarray(den, np.float64)
        assert den.shape == x.shape
    a, b = np.zeros_like(x), np.zeros_like(x)
    check(lib().rl_debug_rho(x.ctypes.data, den.ctypes.data if den is not None else None, len(x), a.ctypes.data, b.ctypes.data))
    return a, b


def letor_parse(data):
    """Native LETOR parser (rl_letor_*): `data` = the file's bytes.  Returns a dict of per-line arrays, the dense row matrix
    (NaN = not named on the line) and the flags of the lines the caller has to parse itself."""
    buf = C.create_string_buffer(data, len(data)) if not isinstance(data, C.Array) else data
    h = C.c_void_p()
    check(lib().rl_letor_parse(buf, len(data), C.byref(h)))
    try:
        n, mf, ns = C.c_int64(), C.c_int32(), C.c_int64()
        check(lib().rl_letor_info(h, C.byref(n), C.byref(mf), C.byref(ns)))
        n, mf = n.value, mf.value
        out = dict(n=n, max_fid=mf, n_slow=ns.value, labels=np.zeros(n, np.float32), last_fid=np.zeros(n, np.int32),
                   qid_off=np.zeros(n, np.int64), qid_len=np.zeros(n, np.int32), desc_off=np.zeros(n, np.int64), desc_len=np.zeros(n, np.int32),
                   line_off=np.zeros(n, np.int64), line_len=np.zeros(n, np.int32), slow=np.zeros(n, np.uint8))
        check(lib().rl_letor_arrays(h, *[out[k].ctypes.data for k in ("labels", "last_fid", "qid_off", "qid_len", "desc_off", "desc_len",
                                                                       "line_off", "line_len", "slow")]))
        X = np.empty((n, mf + 1), np.float32)
        if n:
            check(lib().rl_letor_rows(h, X.ctypes.data, mf + 1))
        out["X"] = X
        return out
    finally:
        lib().rl_letor_destroy(h)


def debug_float_chain(x, seg_start=None, device=0):
    """Java float running sums of the segments of x on the GPU (rl_chain.inc); returns (float32 sums, int32[4] stats)."""
    x = np.ascontiguousarray(x, np.float64)
    seg = np.ascontiguousarray([0, len(x)] if seg_start is None else seg_start, np.int64)
    out = np.zeros(len(seg) - 1, np.float32)
    stats = np.zeros(4, np.int32)
    check(lib().rl_debug_float_chain(device, x.ctypes.data, len(x), seg.ctypes.data, len(seg) - 1, out.ctypes.data, stats.ctypes.data))
    return out, stats


def set_err_max(max_gain):
    """ERRScorer.MAX for trainers created afterwards (rl_set_err_max)"""
    check(lib().rl_set_err_max(float(max_gain)))


def membench(mode, nbytes, stride=1, iters=10, device=0):
    """rl_debug_membench: (avg ms per launch, algorithmic bytes per launch); mode: 0 copy, 1 read, 2 write, 3 32-byte row gather;
    4..9 LDS atomics (consecutive bins / random bins / same address / random + count / three 32-bit / one 32-bit): nbytes = atomic groups per thread, returns (ms, groups per launch)"""
    ms, b = C.c_double(0), C.c_double(0)
    check(lib().rl_debug_membench(device, mode, nbytes, stride, iters, C.byref(ms), C.byref(b)))
    return ms.value, b.value


class Trainer:
    """Thin object wrapper over the rl_trainer handle (one GPU)."""

    def __init__(self, n_trees=1000, n_leaves=10, learning_rate=0.1, n_threshold=256, min_leaf_support=1,
                 early_stop_rounds=100, metric_k=10, device=0, flags=0, metric="NDCG", ranker="LAMBDAMART",
                 feature_sampling_rate=1.0, seed=0):
        L = lib()
        self.p = RlParams()
        L.rl_params_default(C.byref(self.p))
        self.p.n_trees, self.p.n_leaves, self.p.learning_rate = n_trees, n_leaves, learning_rate
        self.p.n_threshold, self.p.min_leaf_support, self.p.early_stop_rounds = n_threshold, min_leaf_support, early_stop_rounds
        self.p.metric_k, self.p.device, self.p.flags = metric_k, device, flags
        self.p.feature_sampling_rate, self.p.seed = feature_sampling_rate, seed
        self.p.metric, self.p.ranker = RL_METRIC[metric.upper()], RL_RANKER[ranker.upper()]
        self.h = C.c_void_p()
        check(L.rl_create(C.byref(self.p), C.byref(self.h)))
        self.cap = max(3, 2 * n_leaves - 1)          # the root always splits once (RegressionTree.java:62-67); -leaf -1: set_train sizes it
        self.N = self.F = self.Q = 0
        self.Nv = 0
        self.has_valid = False

    @staticmethod
    def _prep(X, labels, qoff, qkey, F=None):
        X = np.ascontiguousarray(X, dtype=np.float32)
        if X.ndim != 2:
            raise RankLibError("X must be [n_docs, n_features]")
        labels = np.ascontiguousarray(labels, dtype=np.float32)
        qoff = np.ascontiguousarray(qoff, dtype=np.int32)
        qk = None if qkey is None else np.ascontiguousarray(qkey, dtype=np.int32)
        return X, labels, qoff, qk

    def set_train(self, X, labels, qoff, feature_ids=None, qkey=None, chunk_rows=None):
        """chunk_rows: deliver the rows through rl_set_rows in blocks of that many documents (what the JNI shim does)"""
        X, labels, qoff, qk = self._prep(X, labels, qoff, qkey)
        fid = None if feature_ids is None else np.ascontiguousarray(feature_ids, dtype=np.int32)
        self.N, self.F = X.shape
        self.Q = len(qoff) - 1
        if self.p.n_leaves == -1:
            self.cap = max(3, 2 * max(1, self.N // max(1, self.p.min_leaf_support)) - 1)
        check(lib().rl_set_train(self.h, None if chunk_rows else X.ctypes.data, self.N, self.F, labels.ctypes.data, qoff.ctypes.data, self.Q,
                                 None if fid is None else fid.ctypes.data, None if qk is None else qk.ctypes.data))
        if chunk_rows:
            self.set_rows(X, False, chunk_rows)

    def set_rows(self, X, validation, chunk_rows):
        for a in range(0, X.shape[0], chunk_rows):
            blk = np.ascontiguousarray(X[a:a + chunk_rows])
            check(lib().rl_set_rows(self.h, 1 if validation else 0, a, blk.shape[0], blk.ctypes.data))

    def set_external_judgments(self, validation, ideal_dcg=None, rel_doc_count=None):
        """-qrel: per list, its qid's idealGains entry from the judgment file (NaN = none) / its relDocCount (0 = qid not in the file)"""
        idl = None if ideal_dcg is None else np.ascontiguousarray(ideal_dcg, dtype=np.float64)
        rdc = None if rel_doc_count is None else np.ascontiguousarray(rel_doc_count, dtype=np.int32)
        check(lib().rl_set_external_judgments(self.h, 1 if validation else 0, None if idl is None else idl.ctypes.data,
                                              None if rdc is None else rdc.ctypes.data))

    def set_validation(self, X, labels, qoff, qkey=None):
        X, labels, qoff, qk = self._prep(X, labels, qoff, qkey)
        if X.shape[1] != self.F:
            raise RankLibError("validation set must have the training set's feature columns")
        self.Nv = X.shape[0]
        check(lib().rl_set_validation(self.h, X.ctypes.data, self.Nv, labels.ctypes.data, qoff.ctypes.data,
                                      len(qoff) - 1, None if qk is None else qk.ctypes.data))
        self.has_valid = True

    def dist_unique_id(self):
        """128 bytes for ncclCommInitRank: call on rank 0, broadcast out of band"""
        buf = C.create_string_buffer(128)
        check(lib().rl_dist_unique_id(buf))
        return buf.raw

    def dist_init(self, uid, rank, n_ranks):
        """RCCL transport; every rank passes ITS shard of the queries to set_train (before init)"""
        check(lib().rl_dist_init(self.h, uid, rank, n_ranks))

    def dist_stats(self):
        """[all-reduce calls, all-reduce bytes, all-gather calls, all-gather bytes received, all-to-all calls, all-to-all bytes received from
        other ranks, calls / bytes received of the lazy tie-break's exchanges] of this rank so far"""
        out = np.zeros(8, np.int64)
        check(lib().rl_dist_stats(self.h, out.ctypes.data))
        return out

    def dist_init_callback(self, rank, n_ranks, allreduce, allgather, alltoallv=None):
        """host transport: allreduce(np_array, op) reduces in place, allgather(np_uint8_in) -> np_uint8 [n_ranks*len],
        alltoallv(list of n_ranks np_uint8 arrays to send, list of n_ranks byte counts to receive) -> list of n_ranks np_uint8 arrays received
        (None: emulated with all-gathers)"""
        def _ar(user, ptr, count, dtype, op):
            try:
                arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (count * np.dtype(DT_NUMPY[dtype]).itemsize,)).view(DT_NUMPY[dtype])
                allreduce(arr, op)
                return 0
            except Exception as e:          # noqa: BLE001 -- must not propagate through the C frame
                print("host all-reduce callback failed:", e)
                return 1

        def _ag(user, pin, pout, nbytes):
            try:
                src = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), (nbytes,))
                dst = np.ctypeslib.as_array(C.cast(pout, C.POINTER(C.c_uint8)), (nbytes * n_ranks,))
                dst[:] = allgather(src)
                return 0
            except Exception as e:          # noqa: BLE001
                print("host all-gather callback failed:", e)
                return 1

        def _aa(user, psend, scount, sdispl, precv, rcount, rdispl):
            try:
                sb = max([sdispl[p] + scount[p] for p in range(n_ranks)] + [1])
                rb = max([rdispl[p] + rcount[p] for p in range(n_ranks)] + [1])
                snd = np.ctypeslib.as_array(C.cast(psend, C.POINTER(C.c_uint8)), (sb,))
                rcv = np.ctypeslib.as_array(C.cast(precv, C.POINTER(C.c_uint8)), (rb,))
                got = alltoallv([snd[sdispl[p]:sdispl[p] + scount[p]] for p in range(n_ranks)], [int(rcount[p]) for p in range(n_ranks)])
                for p in range(n_ranks):
                    if len(got[p]) != rcount[p]:
                        raise ValueError("rank %d sent %d bytes, %d expected" % (p, len(got[p]), rcount[p]))
                    rcv[rdispl[p]:rdispl[p] + rcount[p]] = got[p]
                return 0
            except Exception as e:          # noqa: BLE001
                print("host all-to-all callback failed:", e)
                return 1
        self._cb = (HOST_ALLREDUCE(_ar), HOST_ALLGATHER(_ag), HOST_ALLTOALLV(_aa) if alltoallv is not None else HOST_ALLTOALLV())      # keep alive; HOST_ALLTOALLV() is a NULL pointer
        check(lib().rl_dist_init_callback(self.h, rank, n_ranks, self._cb[0], self._cb[1], self._cb[2], None))

    def init(self):
        check(lib().rl_init(self.h))
        cap = C.c_int32(0)
        check(lib().rl_tree_capacity(self.h, C.byref(cap)))      # -leaf -1: 2 * floor(N_global / mls) - 1
        self.cap = cap.value

    def boost_round(self, want_tree=True):
        t = FlatTree(self.cap) if want_tree else None
        ct = t.c() if t else None
        tm, vm, stop = C.c_float(0), C.c_float(0), C.c_int32(0)
        check(lib().rl_boost_round(self.h, C.byref(ct) if t else None, C.byref(tm), C.byref(vm), C.byref(stop)))
        if t:
            t.n_nodes = ct.n_nodes
        return t, np.float32(tm.value), (np.float32(vm.value) if self.has_valid else None), bool(stop.value)

    def boost_rounds_async(self, n):
        check(lib().rl_boost_rounds_async(self.h, n))

    def sync(self):
        check(lib().rl_sync(self.h))

    def finish(self):
        ts, vs = C.c_double(0), C.c_double(0)
        check(lib().rl_finish(self.h, C.byref(ts), C.byref(vs)))
        return ts.value, (vs.value if self.has_valid else None)

    def num_trees(self):
        n = C.c_int32(0)
        check(lib().rl_num_trees(self.h, C.byref(n)))
        return n.value

    def get_tree(self, i):
        t = FlatTree(self.cap)
        ct = t.c()
        check(lib().rl_get_tree(self.h, i, C.byref(ct)))
        t.n_nodes = ct.n_nodes
        return t

    def round_metrics(self, r):
        tm, vm = C.c_float(0), C.c_float(0)
        check(lib().rl_get_round_metrics(self.h, r, C.byref(tm), C.byref(vm)))
        return np.float32(tm.value), (np.float32(vm.value) if self.has_valid else None)

    def best_validation(self):
        b, s = C.c_int32(0), C.c_double(0)
        check(lib().rl_best_validation(self.h, C.byref(b), C.byref(s)))
        return b.value, s.value

    def predict(self, X):
        X = np.ascontiguousarray(X, dtype=np.float32)
        out = np.zeros(X.shape[0], np.float32)
        check(lib().rl_predict(self.h, X.ctypes.data, X.shape[0], out.ctypes.data))
        return out

    def model_text(self):
        need = C.c_int64(0)
        check(lib().rl_model_to_text(self.h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        check(lib().rl_model_to_text(self.h, buf, need.value, C.byref(need)))
        return buf.value.decode("ascii")

    def bin_stride(self):
        s = C.c_int32(0)
        check(lib().rl_bin_stride(self.h, C.byref(s)))
        return s.value

    def quant_exponent(self):
        e = C.c_int32(0)
        check(lib().rl_quant_exponent(self.h, C.byref(e)))
        return e.value

    def hist_features(self):
        """(number of histogram features, column behind each): the data set's features unless a threshold table has more than 4095 entries (rlhip.h)"""
        n = C.c_int32(0)
        check(lib().rl_hist_features(self.h, C.byref(n), None, 0))
        cols = np.zeros(n.value, np.int32)
        check(lib().rl_hist_features(self.h, C.byref(n), cols.ctypes.data_as(C.POINTER(C.c_int32)), n.value))
        return n.value, cols

    def array(self, name):
        which = ARR[name]
        TS = self.bin_stride()
        F_hist = self.hist_features()[0] if name in ("NBINS", "THRESHOLDS", "BINS", "ROOT_COUNT", "ROOT_SUM", "ROOT_SUM_FIXED", "ROOT_SUM_JAVA") else self.F
        shapes = {
            "LAMBDA": ((self.N,), np.float64), "WEIGHT": ((self.N,), np.float64), "SCORE": ((self.N,), np.float64),
            "VALID_SCORE": ((self.Nv,), np.float64), "NBINS": ((F_hist,), np.int32),
            "THRESHOLDS": ((F_hist, TS), np.float32), "BINS": ((F_hist, self.N), np.uint16),
            "ROOT_COUNT": ((F_hist, TS), np.int32), "ROOT_SUM": ((F_hist, TS), np.float64), "ROOT_SUM_JAVA": ((F_hist, TS), np.float64),
            "QUANT": ((self.N,), np.int64), "ROOT_SUM_FIXED": ((F_hist, TS, 2), np.int64),
            "NDCG_PER_QUERY": ((self.Q,), np.float64), "CHAIN_STATS": ((6,), np.int32), "GROW_STATS": ((4,), np.int32), "GROW_DOCS": ((4,), np.int64), "BUBBLES": ((4,), np.int64), "PIECE_STATS": ((2,), np.int64), "SPARSE_INFO": ((8,), np.int64), "PHASE_CLOCKS": ((64, 32), np.int64), "BLOCK_TRACE": ((64, 3, 2048, 8), np.int64), "STEP_LOG": ((8 + 8 * 8192,), np.int32), "TIE_STATS": ((10,), np.int64), "CHAIN_MISS": ((2, self.cap + 1), np.int32),
        }
        shape, dt = shapes[name]
        out = np.zeros(shape, dt)
        check(lib().rl_get_array(self.h, which, out.ctypes.data, out.nbytes))
        return out

    def timing(self, kernel):
        ms, n, b = C.c_double(0), C.c_int64(0), C.c_double(0)
        check(lib().rl_get_timing(self.h, KERNEL[kernel], C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value

    def reset_timing(self):
        check(lib().rl_reset_timing(self.h))

    def set_timing_flags(self, flags):
        check(lib().rl_set_timing_flags(self.h, flags))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            lib().rl_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Model:
    """A scoring-only ensemble loaded from RankLib model text (rl_model_*)."""

    def __init__(self, text, device=0):
        self.h = C.c_void_p()
        check(lib().rl_model_from_text(text.encode("ascii"), device, C.byref(self.h)))

    def num_trees(self):
        n = C.c_int32(0)
        check(lib().rl_model_num_trees(self.h, C.byref(n)))
        return n.value

    def features(self):
        n = C.c_int32(0)
        check(lib().rl_model_features(self.h, None, 0, C.byref(n)))
        ids = np.zeros(max(1, n.value), np.int32)
        check(lib().rl_model_features(self.h, ids.ctypes.data, n.value, C.byref(n)))
        return ids[:n.value]

    def predict_rows(self, rows):
        """rows[:, f] holds feature ID f (column 0 unused, like DataPoint.fVals)"""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        out = np.zeros(rows.shape[0], np.float32)
        check(lib().rl_model_predict(self.h, rows.ctypes.data, rows.shape[0], rows.shape[1], out.ctypes.data))
        return out

    def predict_device(self, dX_ptr, n_docs, row_stride, dOut_ptr, stream=None):
        """rows / scores are device pointers (e.g. torch tensors' data_ptr()); enqueued, not synchronised"""
        check(lib().rl_model_predict_device(self.h, C.c_void_p(dX_ptr), n_docs, row_stride, C.c_void_p(dOut_ptr),
                                            C.c_void_p(stream) if stream else None))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            lib().rl_model_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
