/*
 * rl_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of RankLib's LambdaMART training path
 * (-ranker 6, -metric2t NDCG@k).  It exists to CHECK the HIP path; nothing in
 * the product (ranklib_amd/, include/) may include, link or call it.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY PINNING: the reference (Java) cannot be compiled or run in this
 * environment (no JDK) and its own tests hold no numeric golden vectors for
 * this path => "parity unpinned" by reference artefacts.  The
 * oracle is pinned instead by hand-derived known answers (tests/test_oracle_kat.py)
 * and by an independent numpy restatement (tests/np_restatement.py).
 *
 * All citations are relative to
 *   /root/reference/src/main/java/ciir/umass/edu/
 */
#ifndef RL_ORACLE_H
#define RL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ro_trainer ro_trainer;

typedef struct {
    int32_t n_trees;          /* LambdaMART.nTrees           learning/tree/LambdaMART.java:37 */
    int32_t n_leaves;         /* LambdaMART.nTreeLeaves      :41 (-1 = unlimited)             */
    int32_t n_threshold;      /* LambdaMART.nThreshold       :39 (-1 = all distinct values)    */
    int32_t min_leaf_support; /* LambdaMART.minLeafSupport   :42 */
    int32_t early_stop;       /* LambdaMART.nRoundToStopEarly:40 */
    float   learning_rate;    /* LambdaMART.learningRate     :38 (float!) */
    int32_t metric_k;         /* NDCG@k, DCGScorer.k         metric/DCGScorer.java:21 */
    int32_t n_threads;        /* MyThreadPool size           utilities/MyThreadPool.java:42 */
    int32_t ranker;           /* RO_RANKER_*: 6 = LambdaMART (learning/tree/LambdaMART.java), 0 = MART (learning/tree/MART.java) */
    int32_t metric;           /* RO_METRIC_*: the train / validation metric (-metric2t), metric/MetricScorerFactory.java:24-30 */
    float   feature_sampling_rate;  /* FeatureHistogram.samplingRate (learning/tree/FeatureHistogram.java:34,272-287): < 1 = every split
                                       looks at (int)(rate * F) features drawn without replacement (Random Forests).  0 is read as 1. */
    uint64_t seed;            /* the Java draws from an UNSEEDED java.util.Random; here the draw of a node is a pure function of
                                 (seed, tree index, path of the node from the root), see ro_feature_order() */
} ro_params;

/* The feature draw of one split attempt.  `node_hash`: ro_root_hash(seed, tree) at the root, ro_child_hash(parent, side) below.
 * out_order[0..size) = the drawn feature INDICES in draw order (the scan visits them in this order, so the first drawn wins a tie,
 * FeatureHistogram.java:289-309): the features sorted by (ro_feature_key(node_hash, f), f), first `size`. */
void ro_set_err_max(double max_gain);     /* ERRScorer.MAX (metric/ERRScorer.java:25): process-wide, default 16 */
uint64_t ro_root_hash(uint64_t seed, int32_t tree);
uint64_t ro_child_hash(uint64_t parent, int32_t side /* 0 = left, 1 = right */);
uint64_t ro_feature_key(uint64_t node_hash, int32_t f);
int32_t ro_feature_order(uint64_t node_hash, int32_t n_features, float rate, int32_t *out_order);
enum { RO_RANKER_MART = 0, RO_RANKER_LAMBDAMART = 6 };
enum { RO_METRIC_NDCG = 0, RO_METRIC_DCG = 1, RO_METRIC_MAP = 2, RO_METRIC_ERR = 3 };

/* One regression tree, nodes in pre-order (root = 0, left subtree first), which
 * is also the order of Split.leaves() (learning/tree/Split.java:100-113). */
typedef struct {
    int32_t  n_nodes;
    int32_t  cap;          /* capacity of the arrays below (caller-allocated) */
    int32_t *feature;      /* feature ID (not index); -1 = leaf   Split.java:24 */
    float   *threshold;    /* Split.java:25 */
    int32_t *left, *right; /* child node indices, -1 for leaves */
    float   *output;       /* leaf output, float-valued  LambdaMART.java:412 */
    double  *deviance;     /* Split.deviance (diagnostic) */
    int32_t *count;        /* #samples that reached the node (diagnostic) */
} ro_tree;

/* X: row-major n_docs x n_features, already resolved through
 * DataPoint.getFeatureValue (NaN/missing -> 0, learning/DenseDataPoint.java:21-32).
 * qoff: n_queries+1 offsets into the doc arrays (consecutive docs of one query).
 * qkey: optional (may be NULL): integer key per query; equal keys model equal qid
 *       strings for the idealGains cache quirk (metric/NDCGScorer.java:114-122,134-143).
 * feature_ids: n_features feature IDs written into the model (Ranker.features). */
ro_trainer *ro_create(const ro_params *p,
                      const float *X, int64_t n_docs, int32_t n_features,
                      const float *labels, const int32_t *qoff, int32_t n_queries,
                      const int32_t *feature_ids, const int32_t *qkey);
/* optional validation set (Ranker.setValidationSet, learning/Ranker.java:68-70) */
void ro_set_validation(ro_trainer *t, const float *X, int64_t n_docs,
                       const float *labels, const int32_t *qoff, int32_t n_queries,
                       const int32_t *qkey);
/* -qrel (eval/Evaluator.java:580-591): per query, the idealGains entry its qid has in the judgment file (NaN = none; NDCG) and its
 * relDocCount (0 when the qid is absent; MAP).  NULL = that scorer has no external judgments.  Before ro_init, after the data set. */
void ro_set_external(ro_trainer *t, int validation, const double *ideal, const int32_t *rel_count);
void ro_destroy(ro_trainer *t);

/* LambdaMART.init()  learning/tree/LambdaMART.java:68-166 */
void ro_init(ro_trainer *t);

/* One iteration m of the loop at LambdaMART.java:180-251.  Returns 1 if the
 * early-stop condition (:248) fired after this round, else 0.  train_metric /
 * valid_metric are the float-accumulated per-round values (:216, :237). */
int ro_round(ro_trainer *t, ro_tree *out, float *train_metric, float *valid_metric);

/* Steps of one round, exposed separately for kernel-level parity tests and for
 * timing: ro_round == lambdas; hist_update; fit; leaf outputs; score update; eval */
void ro_compute_lambdas(ro_trainer *t);            /* LambdaMART.java:331-396 */

/* Accessors (pointers stay owned by the trainer) */
int32_t        ro_n_bins(const ro_trainer *t, int32_t f);          /* thresholds[f].length */
const float   *ro_thresholds(const ro_trainer *t, int32_t f);      /* LambdaMART.java:108-150 */
const int32_t *ro_bins(const ro_trainer *t, int32_t f);            /* sampleToThresholdMap[f]  FeatureHistogram.java:102 */
const int32_t *ro_root_count(const ro_trainer *t, int32_t f);      /* cumulative count[f][] */
const double  *ro_root_sum(const ro_trainer *t, int32_t f);        /* cumulative sum[f][] after last update() */
const double  *ro_lambdas(const ro_trainer *t);                    /* pseudoResponses */
const double  *ro_weights(const ro_trainer *t);
const double  *ro_scores(const ro_trainer *t);                     /* modelScores */
const double  *ro_valid_scores(const ro_trainer *t);               /* modelScoresOnValidation, flattened */
int32_t        ro_trees_kept(const ro_trainer *t);                 /* ensemble.treeCount() */
int32_t        ro_best_valid_round(const ro_trainer *t);
double         ro_best_valid_score(const ro_trainer *t);

/* Split trace of the last fitted tree, in the order the splits were made:
 * (feature index, threshold index, S, n_node, n_left) -- diagnostics for
 * classifying a mismatch as near-tie vs bug.  Returns number of splits. */
int32_t ro_last_split_trace(const ro_trainer *t, int32_t cap, int32_t *fidx, int32_t *tidx,
                            double *S, int32_t *n_node, int32_t *n_left);

/* End of learn(): rollback to best validation model (LambdaMART.java:254-256)
 * and compute scorer.score(rank(samples)) with Ensemble.eval float accumulation
 * (:259, learning/tree/Ensemble.java:110-116).  valid_score may be NULL.
 * The training rows are passed again because the trainer does not retain them. */
void ro_finish_with_rows(ro_trainer *t, const float *Xtrain, double *train_score, double *valid_score);

/* kept tree i (after rollback) copied into caller arrays; returns n_nodes or -1 */
int32_t ro_get_tree(const ro_trainer *t, int32_t i, ro_tree *out);
/* FeatureHistogram.update(pseudoResponses) alone (FeatureHistogram.java:114-146) */
void ro_hist_update_only(ro_trainer *t);

/* Ensemble.eval on arbitrary rows (float accumulation in tree order). */
void ro_predict(const ro_trainer *t, const float *X, int64_t n_docs, float *out);

/* Ensemble.eval of a flat model (n_trees x maxn node arrays, feature < 0 = leaf, feature f read from column f, columns
 * >= stride read 0) on n rows with n_threads threads: the CPU baseline of the inference configuration (c4). */
void ro_eval_flat_model(int32_t n_trees, int32_t maxn, const int32_t *feature, const float *thr, const int32_t *left,
                        const int32_t *right, const float *outv, const float *weight, const float *X, int64_t n,
                        int32_t stride, int32_t n_threads, float *res);

/* Stand-alone pieces for known-answer tests */
double ro_exp(double x);                       /* the exp used for rho (fdlibm e_exp restatement) */
double ro_discount(int32_t i);                 /* metric/DCGScorer.java:26 */
/* stable descending index sort of scores[0..n) -> idx (utilities/MergeSorter.java:134-189) */
void   ro_sort_desc(const double *scores, int32_t n, int32_t *idx);
/* lambdas/weights of one query: NDCGScorer.swapChange + LambdaMART.java:361-396 */
void   ro_query_lambdas(const double *scores, const float *labels, int32_t n, int32_t k,
                        double ideal_override /* <0: compute */, double *lambda, double *weight);
/* the same for any metric (RO_METRIC_*): swapChange of metric/{NDCG,DCG,AP,ERR}Scorer.java + LambdaMART.java:361-396 */
void   ro_query_lambdas_metric(int32_t metric, const double *scores, const float *labels, int32_t n, int32_t k,
                               double *lambda, double *weight);
/* MetricScorer.score(RankList) of the list ranked by `scores` (stable descending) */
double ro_query_score(int32_t metric, const double *scores, const float *labels, int32_t n, int32_t k);
double ro_query_ndcg(const double *scores, const float *labels, int32_t n, int32_t k,
                     double ideal_override);
/* float running sum  LambdaMART.java:401-408 */
float  ro_float_chain(const double *x, const int32_t *idx, int32_t n);

#ifdef __cplusplus
}
#endif
#endif
