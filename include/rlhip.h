/*
 * rlhip.h -- C ABI of librlhip.so: an MI355X (gfx950) native LambdaMART trainer that is a
 * drop-in for the training / scoring path behind RankLib's `-ranker 6`.
 *
 * This is the drop-in boundary.  Everything a JNI shim, the ctypes host
 * mirror (ranklib_amd/) or a native CLI needs goes through these entry points: plain
 * pointers and sizes, no C++/torch types, int status codes, no exceptions across the ABI.
 * INTEGRATION.md shows the JNI binding and the Java host class a RankLib maintainer would add.
 *
 * Reference interfaces replaced (paths relative to
 * /root/reference/src/main/java/ciir/umass/edu/):
 *
 *   rl_create / rl_set_*        <- RankerFactory.createRanker + Ranker ctor/setters
 *                                  learning/RankerFactory.java:60-70, learning/Ranker.java:52-74,
 *                                  static parameters learning/tree/LambdaMART.java:37-42
 *   rl_init                     <- LambdaMART.init()            learning/tree/LambdaMART.java:68-166
 *   rl_boost_round(s)           <- one iteration of learn()     learning/tree/LambdaMART.java:180-251
 *   rl_finish                   <- tail of learn()              learning/tree/LambdaMART.java:253-265
 *   rl_get_tree / rl_num_trees  <- getEnsemble()                learning/tree/LambdaMART.java:327-329,
 *                                  Ensemble/RegressionTree/Split learning/tree/Ensemble.java:72-100
 *   rl_predict                  <- LambdaMART.eval -> Ensemble.eval  learning/tree/LambdaMART.java:275-277,
 *                                  learning/tree/Ensemble.java:110-116, learning/tree/Split.java:115-125
 *   rl_model_to_text/from_text  <- model() / loadFromString()   learning/tree/LambdaMART.java:290-310,
 *                                  learning/tree/Ensemble.java:45-70,119-130, learning/tree/Split.java:132-155
 *
 * Error convention: the reference throws unchecked RankLibError (utilities/RankLibError.java:25-42).
 * Here every call returns RL_OK (0) or a negative code and leaves a message retrievable with
 * rl_last_error() (thread-local); a JNI shim rethrows it as RankLibError.create(msg).
 *
 * Threading: like the reference (single caller, global statics) a handle must be used by one
 * thread at a time; distinct handles are independent.
 *
 * Ownership: the caller owns every buffer it passes; rl_set_* copy to HBM and never retain the
 * pointers.  Output buffers are caller-allocated.
 */
#ifndef RLHIP_H
#define RLHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLHIP_ABI_VERSION 5

enum {
    RL_OK = 0,
    RL_ERR_INVALID = -1,     /* bad argument / bad input data (e.g. negative label: learning/DataPoint.java:70-73) */
    RL_ERR_HIP = -2,         /* HIP runtime or kernel failure, message holds file:line */
    RL_ERR_STATE = -3,       /* call out of order (e.g. rl_boost_round before rl_init) */
    RL_ERR_UNSUPPORTED = -4, /* valid for RankLib but not built yet (documented in DESIGN.md) */
    RL_ERR_NO_DEVICE = -5,   /* no gfx950 device visible: there is NO CPU fallback */
    RL_ERR_COMM = -6         /* RCCL failure */
};

/* train / validation metric (-metric2t): metric/{NDCG,DCG,AP,ERR}Scorer.java.  P, RR and BEST are not built for
 * training (RL_ERR_UNSUPPORTED); the host side can still report them on ranked lists. */
enum { RL_METRIC_NDCG = 0, RL_METRIC_DCG = 1, RL_METRIC_MAP = 2, RL_METRIC_ERR = 3 };
/* ranker: the indices of eval/Evaluator.java:69-73 */
enum { RL_RANKER_MART = 0, RL_RANKER_LAMBDAMART = 6 };

enum {                                  /* rl_params.flags */
                                        /* bit 0 is reserved (rl_create rejects it): leaf outputs are always the Java's float running sums
                                           (learning/tree/LambdaMART.java:401-408); there is no non-parity shortcut */
    RL_FLAG_TIMING = 2,                 /* record HIP events around the root histogram and the lambda kernels (rl_get_timing) */
    RL_FLAG_TIMING_NODES = 8,           /* ... and around every growth step's node-histogram launch (30 event pairs per round) */
    RL_FLAG_SERIAL_CHAIN = 4,           /* evaluate the float running sums with the literal serial kernel instead of
                                           the exact parallel scheme (same results; for cross-checks) */
    RL_FLAG_FIRST_TIE = 32,             /* exact ties between split candidates keep the first one in scan order instead of being re-decided in the Java's
                                           summation order (the lazy tie-break, HISTORY.md 4.13: default, also for sharded runs; never with feature sampling).
                                           Faster where nodes are tiny or columns sparse; the trees differ from the reference's only in the stored threshold
                                           inside an empty-bin plateau / in which of two equivalent features is named. */
    RL_FLAG_JAVA_ORDER = 16             /* strict mode: split gains and node deviances come from the f64 histogram RankLib itself
                                           would hold -- every (feature, bin) sum accumulated sequentially in ascending sample
                                           order (FeatureHistogram.java:126-146,166-195), sequential prefix, right sibling =
                                           parent - left on the cumulative arrays (:222-234), sumResponse / sqSumResponse as
                                           :133-137,182-186,202-203 -- so the arg-max of :236-264,302-309 lands on the candidate
                                           the Java picks even where exact arithmetic ties.  Same trees, several
                                           times slower (the sums are serial chains); one GPU only. */
};

typedef struct rl_trainer rl_trainer;   /* opaque */
typedef struct rl_model rl_model;       /* opaque: a loaded ensemble for scoring only */

typedef struct {
    int32_t n_trees;            /* LambdaMART.nTrees            default 1000 */
    int32_t n_leaves;           /* LambdaMART.nTreeLeaves       default 10   */
    int32_t n_threshold;        /* LambdaMART.nThreshold        default 256 (-1: every distinct value; any size: tables beyond 4095 entries run as
                                   several histogram features, rl_hist_features) */
    int32_t min_leaf_support;   /* LambdaMART.minLeafSupport    default 1    */
    int32_t early_stop_rounds;  /* LambdaMART.nRoundToStopEarly default 100  */
    float   learning_rate;      /* LambdaMART.learningRate      default 0.1F (a Java float) */
    int32_t metric;             /* RL_METRIC_* */
    int32_t metric_k;           /* the scorer's k: default 10 for NDCG / DCG / ERR (metric/DCGScorer.java:21,
                                   metric/ERRScorer.java:28), 0 for MAP (metric/APScorer.java:37-39) */
    int32_t device;             /* HIP device ordinal */
    int32_t flags;              /* RL_FLAG_* */
    int32_t ranker;             /* RL_RANKER_LAMBDAMART (default) or RL_RANKER_MART (learning/tree/MART.java:47-65) */
    float   feature_sampling_rate;  /* FeatureHistogram.samplingRate (learning/tree/FeatureHistogram.java:34,272-287), set by
                                   RFRanker.init (learning/tree/RFRanker.java:68): < 1 = every split attempt looks at
                                   (int)(rate * n_features) features drawn without replacement.  Default 1 (0 is read as 1). */
    uint64_t seed;              /* The Java draws from an unseeded java.util.Random.  Here the draw of a node is a pure function of
                                   (seed, index of the tree in this trainer, path of the node from the root): the features sorted by
                                   a 64-bit hash key, the first (int)(rate * F) in key order (the first drawn wins a tie, as the
                                   Java's scan order does).  oracle/rl_oracle.h ro_feature_order() is the same function. */
} rl_params;

/* One regression tree: nodes in pre-order (root 0, left subtree first) == Split.leaves() order
 * (learning/tree/Split.java:100-113).  Arrays are caller-allocated with `cap` entries
 * (2*n_leaves-1 always suffices). */
typedef struct {
    int32_t  n_nodes;     /* out */
    int32_t  cap;         /* in  */
    int32_t *feature;     /* feature ID as written to the model file; -1 = leaf   (Split.featureID) */
    float   *threshold;   /* Split.threshold: go left iff value <= threshold      (Split.java:118)  */
    int32_t *left;        /* child node index, -1 for leaves */
    int32_t *right;
    float   *output;      /* leaf output (Split.avgLabel set by updateTreeOutput), 0 for internal nodes */
    double  *deviance;    /* optional (may be NULL): Split.deviance */
    int32_t *count;       /* optional (may be NULL): training samples that reached the node */
} rl_tree;

/* ---- library ----------------------------------------------------------------------------- */
int         rl_abi_version(void);
const char *rl_last_error(void);
int         rl_device_count(int32_t *n);
void        rl_params_default(rl_params *p);              /* the defaults of LambdaMART.java:37-42 */

/* ERRScorer.MAX (metric/ERRScorer.java:25), the divisor of ERR's relevance grades R = (2^label - 1) / MAX; `-gmax g` sets it to 2^g
 * (eval/Evaluator.java:241-242).  A process-wide static in the reference, and here: trainers created afterwards use it.  Default 16. */
int  rl_set_err_max(double max_gain);

/* ---- trainer ----------------------------------------------------------------------------- */
int  rl_create(const rl_params *p, rl_trainer **out);
void rl_destroy(rl_trainer *t);

/* Training set.  X is row-major [n_docs][n_features], already resolved through
 * DataPoint.getFeatureValue(feature_ids[f]) (missing / NaN -> 0: learning/DenseDataPoint.java:21-32); a NaN left in X is RL_ERR_INVALID at
 * rl_init, +-Infinity is a value like any other (binned as learning/tree/LambdaMART.java:108-149 and FeatureHistogram.java:88-107 bin it).
 * qoff[n_queries+1]: docs of query q are qoff[q]..qoff[q+1]-1 (file order, learning/RankList.java).
 * feature_ids[n_features]: the IDs written into the model (Ranker.features); NULL => 1..n_features.
 * qkey[n_queries]: optional; equal keys == equal qid strings (idealGains cache quirk,
 * metric/NDCGScorer.java:114-122,134-143); NULL => all distinct. */
int rl_set_train(rl_trainer *t, const float *X, int64_t n_docs, int32_t n_features, const float *labels,
                 const int32_t *qoff, int32_t n_queries, const int32_t *feature_ids, const int32_t *qkey);
/* Ranker.setValidationSet (learning/Ranker.java:68-70); same layout, same feature columns. */
int rl_set_validation(rl_trainer *t, const float *X, int64_t n_docs, const float *labels, const int32_t *qoff,
                      int32_t n_queries, const int32_t *qkey);

/* Chunked upload for callers that cannot hold the whole row matrix in one buffer (a Java direct ByteBuffer ends at 2 GiB; MSLR-WEB30K's
 * 3.77 M x 136 floats are 2.05 GB): pass X = NULL to rl_set_train / rl_set_validation (labels, qoff, ids as usual), then deliver the rows
 * in consecutive blocks, first_doc ascending from 0, before rl_init.  X: row-major [n_docs][n_features] of that block. */
int rl_set_rows(rl_trainer *t, int32_t validation, int64_t first_doc, int64_t n_docs, const float *X);

/* -qrel <file> (eval/Evaluator.java:243-244, :580-591): external relevance judgments, resolved per ranked list by the caller.
 *   ideal_dcg[Q]      NDCG: the entry NDCGScorer.loadExternalRelevanceJudgment (metric/NDCGScorer.java:50-96) put into idealGains for the
 *                     list's qid -- it is in the cache before any list is scored, so the list never computes its own; NaN = the qid is not
 *                     in the file.  NULL = no external ideal gains.
 *   rel_doc_count[Q]  MAP: relDocCount of the list's qid (metric/APScorer.java:45-66), 0 when the qid is not in the file (the Java then scores
 *                     the list 0 and returns all-zero swap changes, :86-94, :124-143).  NULL = every list counts its own relevant documents.
 * After rl_set_train / rl_set_validation of that data set, before rl_init.  Sharded runs: every rank passes the entries of ITS lists. */
int rl_set_external_judgments(rl_trainer *t, int32_t validation, const double *ideal_dcg, const int32_t *rel_doc_count);

int rl_init(rl_trainer *t);

/* One boosting round, synchronous.  out may be NULL.  *stop is set to 1 when the early-stop test
 * (learning/tree/LambdaMART.java:248) fires after this round.  train_metric / valid_metric are the
 * float-accumulated per-round values of :216 / :237 (valid_metric untouched without validation data). */
int rl_boost_round(rl_trainer *t, rl_tree *out, float *train_metric, float *valid_metric, int32_t *stop);

/* Enqueue n rounds without any host synchronisation (no validation set allowed: early stopping needs
 * the host).  rl_sync waits for them; trees and per-round metrics are then available through
 * rl_get_tree / rl_get_round_metrics. */
int rl_boost_rounds_async(rl_trainer *t, int32_t n);
int rl_sync(rl_trainer *t);

/* End of learn(): rollback to the best validation model, final scorer.score(rank(samples)) with
 * Ensemble.eval's float accumulation.  valid_score may be NULL. */
int rl_finish(rl_trainer *t, double *train_score, double *valid_score);

int rl_num_trees(const rl_trainer *t, int32_t *n);
/* Nodes the largest possible tree of this trainer has = the `cap` of an rl_tree that always suffices, known after rl_init: 2 * n_leaves - 1, at
 * least 3 (the root is split unconditionally, RegressionTree.java:62-67); with -leaf -1 (n_leaves == -1) 2 * floor(N / min_leaf_support) - 1. */
int rl_tree_capacity(const rl_trainer *t, int32_t *cap);
int rl_get_tree(const rl_trainer *t, int32_t i, rl_tree *out);
int rl_get_round_metrics(const rl_trainer *t, int32_t round, float *train_metric, float *valid_metric);
int rl_best_validation(const rl_trainer *t, int32_t *best_round, double *best_score);

/* Ensemble.eval over rows (float accumulation in tree order).  X row-major [n][n_features] with the
 * trainer's feature columns. */
int rl_predict(rl_trainer *t, const float *X, int64_t n_docs, float *out);

/* ---- model text (RankLib's <ensemble> format) -------------------------------------------- */
/* Writes LambdaMART.model() into buf (NUL-terminated).  Returns RL_OK and *needed = bytes required
 * incl. NUL; if cap < *needed nothing is written. */
int  rl_model_to_text(const rl_trainer *t, char *buf, int64_t cap, int64_t *needed);
int  rl_model_from_text(const char *text, int32_t device, rl_model **out);
void rl_model_destroy(rl_model *m);
int  rl_model_num_trees(const rl_model *m, int32_t *n);
/* Features used by the model: Ensemble.getFeatures (learning/tree/Ensemble.java:132-134). */
int  rl_model_features(const rl_model *m, int32_t *ids, int32_t cap, int32_t *n);
/* X row-major [n][row_stride]; feature ID f is read from column f (column 0 unused, like DataPoint.fVals);
 * IDs >= row_stride read as 0 (the -missingZero behaviour). */
int  rl_model_predict(rl_model *m, const float *X, int64_t n_docs, int32_t row_stride, float *out);
/* The same on buffers that already live in the model's device memory (rows and scores are DEVICE pointers); the work is
 * enqueued on `stream` (a hipStream_t, NULL = the default stream) and NOT synchronised.  This is the call a serving
 * loop uses: Ensemble.eval for a batch of rows without the PCIe round trip (eval/Evaluator.java:1076-1094 does the
 * same per DataPoint on the CPU). */
int  rl_model_predict_device(rl_model *m, const float *dX, int64_t n_docs, int32_t row_stride, float *dOut, void *stream);

/* ---- LETOR text (host only) ---------------------------------------------------
 * `label qid:ID fid:val ... # description` lines as learning/DataPoint.java:58-110 and features/FeatureManager.java:199-235 read
 * them: lines are trimmed, empty ones and '#' comments skipped, the description is the text from '#', the qid / the values are
 * the text after the LAST ':' of their token, values are Float.parseFloat (strtof: correctly rounded).  Every line that is not
 * plain `number qid:token (digits:number)*` -- or that the reference rejects (negative label, feature id <= 0) -- is only
 * FLAGGED (`slow`): the caller parses, or rejects, those lines itself, in file order.  `text` must outlive the handle. */
typedef struct rl_letor rl_letor;
int  rl_letor_parse(const char *text, int64_t len, rl_letor **out);
int  rl_letor_info(const rl_letor *l, int64_t *n_docs, int32_t *max_fid, int64_t *n_slow);
/* per data line (any pointer may be NULL): label, largest feature id, qid / description / whole trimmed line as (offset, length)
 * into `text`, and the slow flag */
int  rl_letor_arrays(const rl_letor *l, float *labels, int32_t *last_fid, int64_t *qid_off, int32_t *qid_len, int64_t *desc_off,
                     int32_t *desc_len, int64_t *line_off, int32_t *line_len, uint8_t *slow);
/* dense rows: X[i * row_stride + fid] = value, NaN where the line does not name the feature (DenseDataPoint's UNKNOWN);
 * row_stride >= max_fid + 1; rows of flagged lines are all NaN */
int  rl_letor_rows(const rl_letor *l, float *X, int64_t row_stride);
void rl_letor_destroy(rl_letor *l);

/* ---- multi-GPU (one process per GPU; queries sharded across ranks) ---------- */
#define RL_UNIQUE_ID_BYTES 128
int rl_dist_unique_id(void *id_out /* RL_UNIQUE_ID_BYTES */);       /* call on rank 0, broadcast out of band */
/* Must be called before rl_init.  Each rank passes ITS shard of the queries to rl_set_train (and, if there is a validation set, its
 * shard of the validation queries to rl_set_validation: every rank or none); the per-split histograms (and the few per-round scalars)
 * are summed across ranks with RCCL, per-query metric values are gathered in rank order. */
int rl_dist_init(rl_trainer *t, const void *id, int32_t rank, int32_t n_ranks);

/* Exchange volume of this rank since rl_dist_init: out[0..7] = all-reduce calls, all-reduce payload bytes, all-gather calls, all-gather
 * bytes received, all-to-all calls, all-to-all bytes received from OTHER ranks -- the per-round pattern -- and, counted apart, the calls and
 * bytes received of the lazy tie-break's exchanges (zeros for an unsharded trainer).  bench.py prints the per-round figures for N > 1. */
int rl_dist_stats(const rl_trainer *t, int64_t *out);

/* The same sharded training over a caller-supplied transport instead of RCCL (gloo, MPI, shared memory ...):
 * the library stages each exchange through host memory and calls back.  Slower (a host synchronisation per
 * exchange); used by the tests to run several ranks on ONE GPU and prove k shards == 1 shard bit for bit.
 * dtype: RL_DT_*, op: RL_OP_*; all-reduce is in place on `host_buf`; all-gather writes n_ranks*bytes to `out`.
 * Callbacks return 0 on success. */
enum { RL_DT_I64 = 0, RL_DT_U64 = 1, RL_DT_I32 = 2, RL_DT_U32 = 3, RL_DT_F64 = 4 };
enum { RL_OP_SUM = 0, RL_OP_MAX = 1, RL_OP_MIN = 2 };
typedef int (*rl_host_allreduce_fn)(void *user, void *host_buf, int64_t count, int32_t dtype, int32_t op);
typedef int (*rl_host_allgather_fn)(void *user, const void *in, void *out, int64_t bytes_per_rank);
/* variable all-to-all in BYTES: rank p receives send[sdispl[p] .. +scount[p]) of this rank; what rank p sends to this rank lands at
 * recv[rdispl[p] .. +rcount[p]) (p == own rank included: a plain copy).  All four arrays have n_ranks entries. */
typedef int (*rl_host_alltoallv_fn)(void *user, const void *send, const int64_t *scount, const int64_t *sdispl, void *recv,
                                    const int64_t *rcount, const int64_t *rdispl);
/* alltoallv may be NULL: the leaf exchange is then emulated with all-gathers of whole send buffers (correct, R times the bytes). */
int rl_dist_init_callback(rl_trainer *t, int32_t rank, int32_t n_ranks, rl_host_allreduce_fn allreduce,
                          rl_host_allgather_fn allgather, rl_host_alltoallv_fn alltoallv, void *user);

/* ---- introspection for parity tests and the roofline report ------------------------------- */
enum {
    RL_ARR_LAMBDA = 1,       /* double[n_docs]  pseudoResponses of the last round */
    RL_ARR_WEIGHT = 2,       /* double[n_docs] */
    RL_ARR_SCORE = 3,        /* double[n_docs]  modelScores */
    RL_ARR_VALID_SCORE = 4,  /* double[n_valid_docs] */
    RL_ARR_NBINS = 5,        /* int32[n_features]           thresholds[f].length */
    RL_ARR_THRESHOLDS = 6,   /* float[n_features*stride]    rows padded to `stride` = max nbins */
    RL_ARR_BINS = 7,         /* uint16[n_features*n_docs]   sampleToThresholdMap, feature-major */
    RL_ARR_ROOT_COUNT = 8,   /* int32[n_features*stride]    cumulative root counts */
    RL_ARR_ROOT_SUM = 9,     /* double[n_features*stride]   cumulative root sums of the last round */
    RL_ARR_QUANT = 10,       /* int64[n_docs]               fixed-point lambdas of the last round */
    RL_ARR_ROOT_SUM_FIXED = 11, /* int64[2*n_features*stride] (hi,lo) 128-bit cumulative fixed-point sums */
    RL_ARR_NDCG_PER_QUERY = 12, /* double[n_queries] of the last round */
    RL_ARR_CHAIN_STATS = 13,    /* int32[6]: leaf float chains {evaluated, candidate-window misses repaired, finished
                                   by the serial kernel}; the same three for the per-round metric chain */
    RL_ARR_CHAIN_MISS = 14,     /* int32[2*(2*n_leaves)]: per (value array, leaf slot) window misses of the last round */
    RL_ARR_GROW_STATS = 15,     /* int32[4] cumulative: growth steps run, nodes prepared (partition + child histograms),
                                   splits committed to trees, trees grown -- speculative best-first growth */
    RL_ARR_ROOT_SUM_JAVA = 17,  /* double[n_features*stride]   with RL_FLAG_JAVA_ORDER: cumulative root sums of the last round in the
                                   Java's own accumulation order (== FeatureHistogram.sum of the root, bit for bit) */
    RL_ARR_GROW_DOCS = 18,      /* int64[4] cumulative documents: accumulated into child histograms (the smaller child of every prepared
                                   node), partitioned, left children of committed splits (= what the Java accumulates: the rho of
                                   SURVEY.md 8d times N), committed split nodes (nu times N) */
    RL_ARR_SPARSE_INFO = 19,    /* int64[8]: 16-feature groups whose root histogram comes from sparse-column entry lists (rl_csc.inc), entries,
                                   groups read as dense rows, live columns in the sparse groups; groups whose child passes read compact rows (0 = off),
                                   their entries (cells outside the mode bins), rows with more than eight entries (dense fallback), row stride */
    RL_ARR_TIE_STATS = 21,      /* int64[10] cumulative, the lazy Java-order tie-break (HISTORY.md 4.13): resolutions run by the host (stalled trees + batches), nodes
                                   whose tied best split was re-decided in the Java's summation order, nodes and documents of the derivation chains that were summed,
                                   host microseconds spent resolving, chain segments evaluated speculatively, candidate-window misses, segments run serially,
                                   [8] of the resolutions the batches at the end of a tree (deferred ties), [9] trees grown a second time (a deferred tie over
                                   several features did not cut the node one way) */
    RL_ARR_STEP_LOG = 20,       /* int32[8 + 8 * 8192], only with RLHIP_STEPLOG=1 in the environment of rl_init (else zeros): [0] = entries written; entry e at
                                   8 + 8 e: {tree, 0, growth step, slot, documents of the split node, documents of the accumulated child, tie flag, slots of the
                                   step} or {tree, 1, tie kind (1 = thresholds of one feature, 2 = several features), right child?, documents, largest node of
                                   the Java-order derivation chain, nodes in the chain, documents in the chain} for a committed split whose best candidate was tied */
    RL_ARR_PHASE_CLOCKS = 16,   /* int64[64][32] device wall-clock stamps (10 ns) inside the last 64 growth steps; all zero unless the
                                   library was built with -DRL_PHASE_CLOCKS (tools/phase_clocks.py) */
    RL_ARR_BLOCK_TRACE = 22,    /* int64[64][3][2048][8] entry / phase / exit stamps (10 ns; [0] entry, [7] exit) of every working block of the partition / child-histogram /
                                   finish kernels in the growth steps of the tree named by RLHIP_TRACE_TREE (-DRL_PHASE_CLOCKS builds,
                                   tools/step_trace.py); RL_ERR_STATE when no trace was requested */
    RL_ARR_PIECE_STATS = 24,    /* int64[2] cumulative, sharded runs with distributed float chains (rl_dist.inc): rounds of the repair loop, pieces re-evaluated from an
                                   exact start state after a detected window miss */
    RL_ARR_BUBBLES = 23         /* int64[4] cumulative device wall-clock time (10 ns units) the main stream idled behind host decisions: [0] from the bookkeeping that ended a
                                   tree to the leaf table's first instruction, [1] trees, [2] from a leaf chain's last stitch to the leaf outputs, [3] rounds (round 6) */
};
/* The device's two exp implementations (rho of learning/tree/LambdaMART.java:383) on n arguments: the branch-free one the
 * lambda kernels use and the literal fdlibm e_exp transcription; both must equal StrictMath.exp bit for bit. */
int rl_debug_exp(const double *x, int32_t n, double *out_fast, double *out_ref);
/* The lambda kernels' divisions (rl_device.h: div_by_rcp / rcp_newton2 = the compiler's own IEEE expansion without the scaling steps that do
 * nothing on the operand ranges of learning/tree/LambdaMART.java:383 and metric/NDCGScorer.java:154) against the compiler's division.
 * den == NULL: out_fast[i] = rho(x[i]) = 1 / (1 + exp(x[i])) as the kernels evaluate it, out_ref[i] = the same through `/` and the literal e_exp.
 * den != NULL: out_fast[i] = x[i] / den[i] through the reciprocal form, out_ref[i] = x[i] / den[i].  Both pairs must be equal bit for bit. */
int rl_debug_rho(const double *x, const double *den, int32_t n, double *out_fast, double *out_ref);
/* The exact parallel evaluation of Java float running sums (`float s = 0; for (k) s += x[k];`, learning/tree/LambdaMART.java:401-408,
 * :474-483) on arbitrary data: n doubles cut into n_seg segments (seg_start[0] = 0 ... seg_start[n_seg] = n), out[s] = the float
 * sum of segment s.  stats (may be null): int32[4] = segments evaluated, window misses repaired, segments finished serially, 0. */
int rl_debug_float_chain(int32_t device, const double *x, int64_t n, const int64_t *seg_start, int32_t n_seg, float *out, int32_t *stats);
int rl_bin_stride(const rl_trainer *t, int32_t *stride);
/* Histogram features of an initialised trainer and the column (0-based position in feature_ids) behind each.  Equal to the data set's features
 * unless a threshold table has more than 4095 entries (-tc -1 on a column with that many distinct values, or -tc N > 4095: learning/tree/
 * LambdaMART.java:135-149): such a feature is split into runs of 4094 thresholds (HISTORY.md 10.4) and RL_ARR_NBINS / THRESHOLDS / BINS /
 * ROOT_* are shaped by THIS count.  columns may be NULL; at most cap entries are written. */
int rl_hist_features(const rl_trainer *t, int32_t *n, int32_t *columns, int32_t cap);
int rl_quant_exponent(const rl_trainer *t, int32_t *e);   /* q = rint(lambda * 2^e) in the last round */
int rl_get_array(rl_trainer *t, int32_t which, void *out, int64_t cap_bytes);

enum { RL_KERNEL_HIST_ROOT = 0, RL_KERNEL_HIST_NODE = 1, RL_KERNEL_LAMBDA = 2, RL_KERNEL_COUNT_ = 3 };
/* With RL_FLAG_TIMING: accumulated HIP-event time (ms), launch count and algorithmic bytes of a kernel
 * since the last rl_reset_timing. */
int rl_get_timing(rl_trainer *t, int32_t kernel, double *total_ms, int64_t *launches, double *alg_bytes);
/* Switch the RL_FLAG_TIMING / RL_FLAG_TIMING_NODES bits of a live trainer (bench.py times the headline region without the 30
 * per-step event pairs, then a few more rounds with them). */
int rl_set_timing_flags(rl_trainer *t, int32_t flags);
/* Memory micro-benchmarks with this library's own access patterns ("re-measure with the build's own copy kernel"),
 * also the known byte counts that calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE (tools/calib_fetch.sh):
 * mode 0 = copy (16 B per lane; reads + writes `bytes`), 1 = streaming read of `bytes`, 2 = streaming write, 3 = 32-byte row
 * gathers through an ascending index list that takes one row in `stride` (the child-node histogram pattern; 32 B row + 4 B
 * index per entry).  avg_ms = mean HIP-event time of `iters` launches, alg_bytes = algorithmic bytes of one launch.
 * Modes 4..9 = LDS atomic throughput in the histogram kernels' own LDS layout (three 256-thread blocks per CU, 16 rows of 264 int64
 * accumulators): 4 consecutive bins per wavefront, 5 a random bin of 256 per lane, 6 one bin per wavefront (same address), 7 = 5 plus the
 * 32-bit count atomic of the child passes, 8 three 32-bit atomics instead of one 64-bit one, 9 one 32-bit atomic; `bytes` = atomic groups per
 * thread, alg_bytes returns the groups of one launch (the calibration behind bench.py's lds_atomics_frac_of_measured_peak). */
int rl_debug_membench(int32_t device, int32_t mode, int64_t bytes, int32_t stride, int32_t iters, double *avg_ms, double *alg_bytes);
int rl_reset_timing(rl_trainer *t);

#ifdef __cplusplus
}
#endif
#endif
