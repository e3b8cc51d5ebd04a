/*
 * rl_oracle.c -- TEST INFRASTRUCTURE ONLY (see rl_oracle.h).
 *
 * A from-scratch C restatement of what RankLib's Java does for
 *   -ranker 6 -metric2t NDCG@k
 * with the same evaluation order and the same float/double casts, so that
 * its results stand in for the (un-runnable here) Java reference.
 * Single-threaded it is the semantic oracle; with n_threads > 1 it splits
 * work exactly like MyThreadPool.partition and serves as the CPU baseline.
 *
 * PARITY UNPINNED by reference artefacts (no JDK, no golden vectors in the
 * reference's tests); pinned by hand-derived KATs + an independent numpy
 * restatement in tests/.
 *
 * Citations are relative to /root/reference/src/main/java/ciir/umass/edu/.
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: Java never fuses).
 */
#include "rl_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* exp: restatement of the published fdlibm e_exp algorithm (what Java's
 * StrictMath.exp is specified as; Math.exp is allowed 1 ulp around it).
 * Used for rho at learning/tree/LambdaMART.java:383.                          */
/* ------------------------------------------------------------------------- */
static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

double ro_exp(double x)
{
    const double ln2HI = u2d(0x3fe62e42fee00000ULL), ln2LO = u2d(0x3dea39ef35793c76ULL);
    const double invln2 = u2d(0x3ff71547652b82feULL);
    const double P1 = u2d(0x3FC555555555553EULL), P2 = u2d(0xBF66C16C16BEBD93ULL),
                 P3 = u2d(0x3F11566AAF25DE2CULL), P4 = u2d(0xBEBBBD41C5D26BF1ULL),
                 P5 = u2d(0x3E66376972BEA4D0ULL);
    const double o_threshold = u2d(0x40862E42FEFA39EFULL), u_threshold = u2d(0xc0874910D52D3051ULL);
    const double huge = 1.0e+300, twom1000 = u2d(0x0170000000000000ULL); /* 2^-1000 */
    double hi = 0.0, lo = 0.0, c, t, y;
    int32_t k = 0;
    uint64_t bits = d2u(x);
    uint32_t hx = (uint32_t)(bits >> 32);
    int xsb = (int)((hx >> 31) & 1);
    hx &= 0x7fffffff;

    if (hx >= 0x40862E42) {                 /* |x| >= 709.78... */
        if (hx >= 0x7ff00000) {
            uint32_t lx = (uint32_t)bits;
            if (((hx & 0xfffff) | lx) != 0) return x + x; /* NaN */
            return xsb == 0 ? x : 0.0;                  /* exp(+-inf) */
        }
        if (x > o_threshold) return huge * huge;
        if (x < u_threshold) return twom1000 * twom1000;
    }
    if (hx > 0x3fd62e42) {                  /* |x| > 0.5 ln2 */
        if (hx < 0x3FF0A2B2) {              /* and |x| < 1.5 ln2 */
            hi = x - (xsb ? -ln2HI : ln2HI);
            lo = xsb ? -ln2LO : ln2LO;
            k = 1 - xsb - xsb;
        } else {
            k = (int32_t)(invln2 * x + (xsb ? -0.5 : 0.5));
            t = k;
            hi = x - t * ln2HI;
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000) {           /* |x| < 2^-28 */
        if (huge + x > 1.0) return 1.0 + x;
    } else {
        k = 0;
    }
    t = x * x;
    c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    if (k >= -1021) {
        return u2d(d2u(y) + ((uint64_t)(uint32_t)k << 52));
    }
    y = u2d(d2u(y) + ((uint64_t)(uint32_t)(k + 1000) << 52));
    return y * twom1000;
}

/* metric/DCGScorer.java:26 with utilities/SimpleMath.java:17,24-26:
 * discount(i) = 1.0 / (Math.log(i + 2) / Math.log(2)) */
double ro_discount(int32_t i) { return 1.0 / (log((double)(i + 2)) / log(2.0)); }

/* metric/DCGScorer.java:28-31: gain(l) = (1 << l) - 1 (int arithmetic) */
static inline int32_t java_pow2m1(int32_t rel) { return (int32_t)(((uint32_t)1 << (rel & 31)) - 1u); }      /* Java: shift count mod 32, the subtraction wraps */
static inline double gain_of(int rel) { return (double)java_pow2m1(rel); }

/* ------------------------------------------------------------------------- */
/* Stable index merge sort over natural runs -- utilities/MergeSorter.java:134-217.
 * idx[] holds absolute indices begin..begin+n-1; ties keep the lower index first
 * (merge takes the left element on >= / <=, :198,:204).                        */
/* ------------------------------------------------------------------------- */
static void sort_idx(const double *list, int32_t begin, int32_t n, int asc, int32_t *idx, int32_t *tmp,
                     int32_t *runs /* n+1 */)
{
    for (int32_t i = 0; i < n; i++) idx[i] = begin + i;
    if (n < 2) return;
    /* natural runs (:150-170) */
    int32_t nr = 0;
    runs[nr++] = 0;
    for (int32_t i = 1; i < n; i++) {
        double a = list[begin + i - 1], b = list[begin + i];
        int ok = asc ? (b >= a) : (b <= a);
        if (!ok) runs[nr++] = i;
    }
    runs[nr] = n;
    int32_t *src = idx, *dst = tmp;
    while (nr > 1) {
        int32_t w = 0, out = 0;
        for (int32_t r = 0; r + 1 < nr; r += 2) {
            int32_t i = runs[r], e1 = runs[r + 1], j = e1, e2 = runs[r + 2], k = i;
            while (i < e1 && j < e2) {
                double a = list[src[i]], b = list[src[j]];
                int takeLeft = asc ? (a <= b) : (a >= b);
                dst[k++] = takeLeft ? src[i++] : src[j++];
            }
            while (i < e1) dst[k++] = src[i++];
            while (j < e2) dst[k++] = src[j++];
            runs[w++] = runs[r];
            out = e2;
        }
        if (nr & 1) {
            int32_t s = runs[nr - 1];
            memcpy(dst + s, src + s, (size_t)(n - s) * sizeof(int32_t));
            runs[w++] = s;
            out = n;
        }
        (void)out;
        runs[w] = n;
        nr = w;
        int32_t *sw = src; src = dst; dst = sw;
    }
    if (src != idx) memcpy(idx, src, (size_t)n * sizeof(int32_t));
}

void ro_sort_desc(const double *scores, int32_t n, int32_t *idx)
{
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * n + 2));
    sort_idx(scores, 0, n, 0, idx, tmp, tmp + n);
    free(tmp);
}

/* ------------------------------------------------------------------------- */
/* Thread pool with RankLib's work split -- utilities/MyThreadPool.java:50-87   */
/* ------------------------------------------------------------------------- */
typedef void (*chunk_fn)(void *ctx, int32_t start, int32_t end /* inclusive */, int32_t worker);

typedef struct {
    pthread_t *th;
    int32_t size;
    pthread_mutex_t mu;
    pthread_cond_t cv_go, cv_done;
    uint64_t gen;
    int32_t pending;
    int stop;
    chunk_fn fn;
    void *ctx;
    int32_t part[1026];
    int32_t nchunks;
} pool_t;

typedef struct { pool_t *p; int32_t id; } pool_arg;

static void *pool_main(void *a_)
{
    pool_arg *a = (pool_arg *)a_;
    pool_t *p = a->p;
    uint64_t seen = 0;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (p->gen == seen && !p->stop) pthread_cond_wait(&p->cv_go, &p->mu);
        if (p->stop) { pthread_mutex_unlock(&p->mu); break; }
        seen = p->gen;
        pthread_mutex_unlock(&p->mu);
        if (a->id < p->nchunks) p->fn(p->ctx, p->part[a->id], p->part[a->id + 1] - 1, a->id);
        pthread_mutex_lock(&p->mu);
        if (--p->pending == 0) pthread_cond_signal(&p->cv_done);
        pthread_mutex_unlock(&p->mu);
    }
    free(a);
    return NULL;
}

static void pool_init(pool_t *p, int32_t size)
{
    memset(p, 0, sizeof(*p));
    if (size > 1024) size = 1024;
    p->size = size < 1 ? 1 : size;
    if (p->size == 1) return;
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->cv_go, NULL);
    pthread_cond_init(&p->cv_done, NULL);
    p->th = (pthread_t *)calloc((size_t)p->size, sizeof(pthread_t));
    for (int32_t i = 0; i < p->size; i++) {
        pool_arg *a = (pool_arg *)malloc(sizeof(pool_arg));
        a->p = p; a->id = i;
        pthread_create(&p->th[i], NULL, pool_main, a);
    }
}

static void pool_free(pool_t *p)
{
    if (p->size <= 1) return;
    pthread_mutex_lock(&p->mu);
    p->stop = 1;
    pthread_cond_broadcast(&p->cv_go);
    pthread_mutex_unlock(&p->mu);
    for (int32_t i = 0; i < p->size; i++) pthread_join(p->th[i], NULL);
    free(p->th);
}

/* MyThreadPool.partition (:77-87): min(n,size) contiguous chunks, the first
 * n % chunks get one extra element. */
static int32_t pool_partition(int32_t size, int32_t n, int32_t *part)
{
    int32_t nChunks = n < size ? n : size;
    if (nChunks <= 0) { part[0] = 0; return 0; }
    int32_t chunk = n / nChunks, mod = n % nChunks;
    part[0] = 0;
    for (int32_t i = 1; i <= nChunks; i++) part[i] = part[i - 1] + chunk + (i <= mod ? 1 : 0);
    return nChunks;
}

/* MyThreadPool.execute(worker, nTasks) (:50-62); returns number of chunks used */
static int32_t pool_run(pool_t *p, int32_t nTasks, chunk_fn fn, void *ctx)
{
    if (p->size == 1) { /* callers take the "p.size() == 1" branch of the Java */
        fn(ctx, 0, nTasks - 1, 0);
        return 1;
    }
    pthread_mutex_lock(&p->mu);
    p->nchunks = pool_partition(p->size, nTasks, p->part);
    p->fn = fn; p->ctx = ctx;
    p->pending = p->size;
    p->gen++;
    pthread_cond_broadcast(&p->cv_go);
    while (p->pending > 0) pthread_cond_wait(&p->cv_done, &p->mu);
    int32_t used = p->nchunks;
    pthread_mutex_unlock(&p->mu);
    return used;
}

/* ------------------------------------------------------------------------- */
/* Data structures                                                             */
/* ------------------------------------------------------------------------- */
typedef struct {            /* learning/tree/FeatureHistogram.java:36-44 */
    double  *sum;           /* [F][TS] cumulative label sums  */
    int32_t *count;         /* [F][TS] cumulative counts      */
    double   sumResponse, sqSumResponse;
    int      owns_arrays;
} hist_t;

typedef struct node_s {     /* learning/tree/Split.java:22-38 */
    int32_t featureID;      /* -1 = leaf */
    float   threshold;
    double  avgLabel;       /* output */
    int     isRoot;
    double  deviance;
    int32_t *samples;
    int32_t n;
    hist_t  *hist;
    struct node_s *left, *right;
    uint64_t ph;            /* path hash: identifies the node for the seeded feature draw (ro_feature_order) */
} node_t;

typedef struct {            /* one data set (train or validation) */
    int64_t n;
    int32_t q;
    const float *X;         /* borrowed until ro_init copies what it needs */
    float   *Xown;          /* validation keeps a private copy for rt.eval  */
    float   *labels;
    int32_t *qoff;
    int32_t *qkey;          /* may be NULL */
    /* -qrel (eval/Evaluator.java:580-591): external relevance judgments resolved per query by the caller */
    double  *ext_ideal;     /* NDCGScorer.loadExternalRelevanceJudgment: idealGains entry of the list's qid, NaN = none (may be NULL) */
    int32_t *ext_rd;        /* APScorer.loadExternalRelevanceJudgment: relDocCount of the list's qid, 0 when the qid is not in the file (NULL = no -qrel) */
} dataset_t;

typedef struct {            /* kept model: learning/tree/Ensemble.java:33-35 */
    int32_t n_nodes;
    int32_t *feature; float *threshold; int32_t *left, *right; float *output;
} kept_tree;

struct ro_trainer {
    ro_params p;
    int32_t F, TS;                  /* features, histogram row stride (max bins) */
    int32_t *feature_ids;
    dataset_t tr, va;
    int has_valid;
    pool_t pool;

    float   **thr;  int32_t *nthr;  /* thresholds[f][]  LambdaMART.java:45 */
    int32_t **bins;                 /* sampleToThresholdMap[f][k] */
    hist_t  root;                   /* LambdaMART.hist */
    double *modelScores, *pseudoResponses, *weights;
    double *validScores;            /* flattened modelScoresOnValidation */
    float  *Xcol;                   /* column-major copy of train X, freed after init */

    /* idealGains cache  metric/NDCGScorer.java:32 */
    int32_t *ck; double *cv; uint8_t *cu; int32_t ccap;
    int32_t next_anon_key;

    /* ensemble */
    kept_tree *trees; int32_t n_trees, cap_trees;
    int32_t bestModelOnValidation;  /* LambdaMART.java:50 */
    double  bestScoreOnValidationData;
    int32_t round;

    /* trace of last tree */
    int32_t trace_n; int32_t *trace_f, *trace_t, *trace_nn, *trace_nl; double *trace_S; int32_t trace_cap;

    /* scratch for the serial per-query work */
    int32_t maxq;
};

/* ---- idealGains cache (HashMap<String,Double>) --------------------------- */
static void cache_init(ro_trainer *t, int32_t nkeys)
{
    int32_t cap = 16;
    while (cap < 2 * nkeys + 2) cap <<= 1;
    t->ccap = cap;
    t->ck = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    t->cv = (double *)malloc(sizeof(double) * (size_t)cap);
    t->cu = (uint8_t *)calloc((size_t)cap, 1);
}
static int cache_get(const ro_trainer *t, int32_t key, double *out)
{
    uint32_t h = ((uint32_t)key * 2654435761u) & (uint32_t)(t->ccap - 1);
    while (t->cu[h]) {
        if (t->ck[h] == key) { *out = t->cv[h]; return 1; }
        h = (h + 1) & (uint32_t)(t->ccap - 1);
    }
    return 0;
}
static void cache_put(ro_trainer *t, int32_t key, double v)
{
    uint32_t h = ((uint32_t)key * 2654435761u) & (uint32_t)(t->ccap - 1);
    while (t->cu[h]) {
        if (t->ck[h] == key) { t->cv[h] = v; return; }
        h = (h + 1) & (uint32_t)(t->ccap - 1);
    }
    t->cu[h] = 1; t->ck[h] = key; t->cv[h] = v;
}

/* ------------------------------------------------------------------------- */
/* NDCG pieces -- metric/NDCGScorer.java:103-174, metric/DCGScorer.java:97-103 */
/* ------------------------------------------------------------------------- */
/* getIdealDCG (:167-174): labels sorted descending (Sorter.sort), top `topK`.
 * Equal labels have equal gains, so any correct descending sort gives the same
 * sequence of addends. */
static int cmp_int_desc(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x < y) - (x > y);
}
static double ideal_dcg(const int32_t *rel, int32_t n, int32_t topK, int32_t *scratch)
{
    memcpy(scratch, rel, sizeof(int32_t) * (size_t)n);
    qsort(scratch, (size_t)n, sizeof(int32_t), cmp_int_desc);
    double dcg = 0;
    for (int32_t i = 0; i < topK; i++) dcg += gain_of(scratch[i]) * ro_discount(i);
    return dcg;
}

/* discount table shared by all callers (DCGScorer's static cache :17,:106-123) */
static double *g_disc = NULL; static int32_t g_disc_n = 0;
static pthread_mutex_t g_disc_mu = PTHREAD_MUTEX_INITIALIZER;
static void disc_reserve(int32_t n)
{
    pthread_mutex_lock(&g_disc_mu);
    if (n > g_disc_n) {
        int32_t m = n + 1000;
        double *d = (double *)malloc(sizeof(double) * (size_t)m);
        for (int32_t i = 0; i < m; i++) d[i] = ro_discount(i);
        /* old table intentionally leaked if another thread may still read it */
        g_disc = d; g_disc_n = m;
    }
    pthread_mutex_unlock(&g_disc_mu);
}

/* NDCGScorer.score(RankList) (:103-129) on labels already in ranked order.
 * key < 0: no cache interaction (stand-alone use). */
static double ndcg_score_ranked(ro_trainer *t, const int32_t *rel, int32_t n, int32_t k, int32_t key,
                                int32_t *scratch)
{
    if (n == 0) return 0;
    int32_t size = k;
    if (k > n || k <= 0) size = n;
    double ideal;
    if (!(t && key >= 0 && cache_get(t, key, &ideal))) {
        ideal = ideal_dcg(rel, n, size, scratch);
        if (t && key >= 0) cache_put(t, key, ideal);
    }
    if (ideal <= 0.0) return 0.0;
    double dcg = 0;                                   /* getDCG, DCGScorer.java:97-103 */
    for (int32_t i = 0; i < size; i++) dcg += gain_of(rel[i]) * g_disc[i];
    return dcg / ideal;
}

/* rdCount of the list being processed when an external judgment file is loaded (APScorer.java:124-133, :86-94); -1 = relDocCount == null.
 * Set by the per-query callers (they run inside pool threads). */
static __thread int32_t g_ext_rd = -1;

/* APScorer.swapChange (metric/APScorer.java:108-162) on float labels in ranked order; changes = n*n row-major.
 * K is ignored ("consider the entire ranked list"). */
static void ap_swap_change(const float *lab, int32_t n, double *changes)
{
    int32_t *relCount = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * n + 2)), *labels = relCount + n + 1;
    int32_t count = 0;
    for (int32_t i = 0; i < n; i++) {
        if (lab[i] > 0) { labels[i] = 1; count++; } else labels[i] = 0;       /* :113-121 (float compare) */
        relCount[i] = count;
    }
    const int32_t rdCount = g_ext_rd >= 0 ? g_ext_rd : count;                  /* :124-133 */
    memset(changes, 0, sizeof(double) * (size_t)n * (size_t)n);
    if (rdCount == 0 || count == 0) { free(relCount); return; }               /* :141-143 */
    for (int32_t i = 0; i < n - 1; i++)
        for (int32_t j = i + 1; j < n; j++) {
            double change = 0;
            if (labels[i] != labels[j]) {
                const int32_t diff = labels[j] - labels[i];
                change += ((double)((relCount[i] + diff) * labels[j] - relCount[i] * labels[i])) / (i + 1);   /* :150 */
                for (int32_t k = i + 1; k <= j - 1; k++) if (labels[k] > 0) change += ((double)diff) / (k + 1);
                change += ((double)(-relCount[j] * diff)) / (j + 1);                                           /* :156 */
            }
            changes[(size_t)j * n + i] = changes[(size_t)i * n + j] = change / rdCount;                        /* :159 */
        }
    free(relCount);
}

static double g_err_max = 16.0;                                                   /* ERRScorer.MAX (static, :25); -gmax sets 2^gmax */
void ro_set_err_max(double m) { g_err_max = m; }
static double err_R(int32_t rel) { return java_pow2m1(rel) / g_err_max; }        /* ERRScorer.java:71-73 */

/* ERRScorer.swapChange (metric/ERRScorer.java:76-115): labels, R and np are only filled for the top `size`
 * positions (the rest stay 0), and np is the running product as written (p *= np[i]). */
static void err_swap_change(const float *lab, int32_t n, int32_t k, double *changes)
{
    const int32_t size = (n > k) ? k : n;
    int32_t *labels = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    double *R = (double *)calloc((size_t)(2 * n + 2), sizeof(double)), *np = R + n + 1;
    double p = 1.0;
    for (int32_t i = 0; i < size; i++) {
        labels[i] = (int32_t)lab[i];
        R[i] = err_R(labels[i]);
        np[i] = p * (1.0 - R[i]);
        p *= np[i];
    }
    memset(changes, 0, sizeof(double) * (size_t)n * (size_t)n);
    for (int32_t i = 0; i < size; i++) {
        const double v1 = 1.0 / (i + 1) * (i == 0 ? 1 : np[i - 1]);
        double change = 0;
        for (int32_t j = i + 1; j < n; j++) {
            if (labels[i] == labels[j]) change = 0;
            else {
                change = v1 * (R[j] - R[i]);
                p = (i == 0 ? 1 : np[i - 1]) * (R[i] - R[j]);
                for (int32_t kk = i + 1; kk < j; kk++) { change += p * R[kk] / (1 + kk); p *= 1.0 - R[kk]; }
                change += (np[j - 1] * (1.0 - R[j]) * R[i] / (1.0 - R[i]) - np[j - 1] * R[j]) / (j + 1);
            }
            changes[(size_t)j * n + i] = changes[(size_t)i * n + j] = change;
        }
    }
    free(labels); free(R);
}

/* MetricScorer.score(RankList) on float labels in ranked order: DCGScorer.score (metric/DCGScorer.java:58-71),
 * APScorer.score (metric/APScorer.java:73-100), ERRScorer.score (metric/ERRScorer.java:45-64). */
static double dcg_score_ranked(const float *lab, int32_t n, int32_t k)
{
    if (n == 0) return 0;
    int32_t size = k;
    if (k > n || k <= 0) size = n;
    double dcg = 0;
    for (int32_t i = 0; i < size; i++) dcg += gain_of((int32_t)lab[i]) * g_disc[i];
    return dcg;
}
static double ap_score_ranked(const float *lab, int32_t n)
{
    double ap = 0.0; int32_t count = 0;
    for (int32_t i = 0; i < n; i++) if (lab[i] > 0.0) { count++; ap += ((double)count) / (i + 1); }
    const int32_t rdCount = g_ext_rd >= 0 ? g_ext_rd : count;                  /* APScorer.java:86-94 */
    if (rdCount == 0) return 0.0;
    return ap / rdCount;
}
static double err_score_ranked(const float *lab, int32_t n, int32_t k)
{
    int32_t size = k;
    if (k > n || k <= 0) size = n;
    double s = 0.0, p = 1.0;
    for (int32_t i = 1; i <= size; i++) {
        const double R = err_R((int32_t)lab[i - 1]);
        s += p * R / i;
        p *= (1.0 - R);
    }
    return s;
}

/* scorer.getK(): APScorer's constructor sets k = 0 (metric/APScorer.java:37-39) and the factory only overrides it
 * for "MAP@k" (metric/MetricScorerFactory.java:43-57); callers pass that k. */

/* One query of computePseudoResponses (LambdaMART.java:364-394) with
 * NDCGScorer.swapChange (:132-160) evaluated on the fly instead of
 * materialising the n x n matrix (same values, same order of use); the other
 * metrics materialise their matrix like the Java does.
 * scores/labels/lambda/weight are the GLOBAL arrays; docs cur..cur+n-1. */
static void query_lambdas(const ro_trainer *t, const double *modelScores, const float *labels, int32_t cur,
                          int32_t n, int32_t k, int32_t key, double ideal_override, double *pseudo,
                          double *weights, int32_t *idx, int32_t *tmp, int32_t *rel)
{
    const int32_t metric = t ? t->p.metric : (ideal_override <= -2.0 ? (int32_t)(-ideal_override - 2.0) : RO_METRIC_NDCG);
    sort_idx(modelScores, cur, n, 0, idx, tmp, tmp + n);      /* :366 */
    for (int32_t i = 0; i < n; i++) rel[i] = (int32_t)labels[idx[i]]; /* MetricScorer.java:54-60 */
    int32_t size = (n > k) ? k : n;                           /* NDCGScorer.java:133 */
    double ideal = 1.0;
    double *changes = NULL;
    if (metric == RO_METRIC_NDCG) {
        if (ideal_override >= 0) ideal = ideal_override;
        else if (!(t && key >= 0 && cache_get(t, key, &ideal))) ideal = ideal_dcg(rel, n, size, tmp); /* :137-143 */
    } else if (metric == RO_METRIC_MAP || metric == RO_METRIC_ERR) {
        float *lr = (float *)malloc(sizeof(float) * (size_t)(n + 1));
        for (int32_t i = 0; i < n; i++) lr[i] = labels[idx[i]];
        changes = (double *)malloc(sizeof(double) * (size_t)n * (size_t)n + 8);
        if (metric == RO_METRIC_MAP) ap_swap_change(lr, n, changes); else err_swap_change(lr, n, k, changes);
        free(lr);
    }
    const int32_t cutoff = k;                                 /* LambdaMART.java:362 */
    for (int32_t j = 0; j < n; j++) {
        const int32_t mj = idx[j];
        for (int32_t kk = 0; kk < n; kk++) {
            if (j > cutoff && kk > cutoff) break;             /* :375-377 */
            const int32_t mk = idx[kk];
            if (labels[mj] > labels[mk]) {                    /* :380 float compare */
                double change = 0.0;
                if (changes) change = changes[(size_t)j * n + kk];
                else if (j != kk) {
                    int32_t a = j < kk ? j : kk, b = j < kk ? kk : j;
                    if (metric == RO_METRIC_NDCG) {           /* changes[j][kk], NDCGScorer.java:151-157 */
                        if (ideal > 0 && a < size) change = (g_disc[a] - g_disc[b]) * (gain_of(rel[a]) - gain_of(rel[b])) / ideal;
                    } else if (a < size)                      /* DCGScorer.java:84-88 */
                        change = (g_disc[a] - g_disc[b]) * (gain_of(rel[a]) - gain_of(rel[b]));
                }
                const double deltaNDCG = fabs(change);        /* :381 */
                if (deltaNDCG > 0) {
                    const double rho = 1.0 / (1 + ro_exp(modelScores[mj] - modelScores[mk])); /* :383 */
                    const double lambda = rho * deltaNDCG;
                    pseudo[mj] += lambda;
                    pseudo[mk] -= lambda;
                    const double delta = rho * (1.0 - rho) * deltaNDCG;
                    weights[mj] += delta;
                    weights[mk] += delta;
                }
            }
        }
    }
    free(changes);
}

/* scorer.score(rl) of one query, labels given through the ranking idx (MetricScorer subclasses) */
static double query_score(ro_trainer *t, int32_t metric, const float *labels, const int32_t *idx, int32_t n, int32_t k,
                          int32_t key, int32_t *rel, int32_t *scratch)
{
    if (metric == RO_METRIC_NDCG) {
        for (int32_t i = 0; i < n; i++) rel[i] = (int32_t)labels[idx[i]];
        return ndcg_score_ranked(t, rel, n, k, key, scratch);
    }
    float *lr = (float *)malloc(sizeof(float) * (size_t)(n + 1));
    for (int32_t i = 0; i < n; i++) lr[i] = labels[idx[i]];
    double r;
    if (metric == RO_METRIC_DCG) r = dcg_score_ranked(lr, n, k);
    else if (metric == RO_METRIC_MAP) r = ap_score_ranked(lr, n);
    else r = err_score_ranked(lr, n, k);
    free(lr);
    return r;
}

void ro_query_lambdas_metric(int32_t metric, const double *scores, const float *labels, int32_t n, int32_t k,
                             double *lambda, double *weight)
{
    disc_reserve(n + 2);
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(4 * n + 4));
    memset(lambda, 0, sizeof(double) * (size_t)n);
    memset(weight, 0, sizeof(double) * (size_t)n);
    g_ext_rd = -1;
    query_lambdas(NULL, scores, labels, 0, n, k, -1, metric == RO_METRIC_NDCG ? -1.0 : -2.0 - metric, lambda, weight,
                  buf, buf + n, buf + 3 * n + 2);
    free(buf);
}

double ro_query_score(int32_t metric, const double *scores, const float *labels, int32_t n, int32_t k)
{
    disc_reserve(n + 2);
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(4 * n + 4));
    int32_t *idx = buf, *tmp = buf + n, *rel = buf + 3 * n + 2;
    sort_idx(scores, 0, n, 0, idx, tmp, tmp + n);
    g_ext_rd = -1;
    const double r = query_score(NULL, metric, labels, idx, n, k, -1, rel, tmp);
    free(buf);
    return r;
}

void ro_query_lambdas(const double *scores, const float *labels, int32_t n, int32_t k, double ideal_override,
                      double *lambda, double *weight)
{
    disc_reserve(n + 2);
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(4 * n + 4));
    memset(lambda, 0, sizeof(double) * (size_t)n);
    memset(weight, 0, sizeof(double) * (size_t)n);
    query_lambdas(NULL, scores, labels, 0, n, k, -1, ideal_override, lambda, weight, buf, buf + n, buf + 3 * n + 2);
    free(buf);
}

double ro_query_ndcg(const double *scores, const float *labels, int32_t n, int32_t k, double ideal_override)
{
    disc_reserve(n + 2);
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(4 * n + 4));
    int32_t *idx = buf, *tmp = buf + n, *rel = buf + 3 * n + 2;
    sort_idx(scores, 0, n, 0, idx, tmp, tmp + n);
    for (int32_t i = 0; i < n; i++) rel[i] = (int32_t)labels[idx[i]];
    double r;
    if (ideal_override >= 0) {
        int32_t size = (k > n || k <= 0) ? n : k;
        double dcg = 0;
        for (int32_t i = 0; i < size; i++) dcg += gain_of(rel[i]) * g_disc[i];
        r = ideal_override <= 0.0 ? 0.0 : dcg / ideal_override;
    } else {
        r = ndcg_score_ranked(NULL, rel, n, k, -1, tmp);
    }
    free(buf);
    return r;
}

float ro_float_chain(const double *x, const int32_t *idx, int32_t n)
{
    float s = 0.0f;
    for (int32_t i = 0; i < n; i++) s = (float)((double)s + x[idx ? idx[i] : i]);
    return s;
}

/* ------------------------------------------------------------------------- */
/* create / destroy                                                            */
/* ------------------------------------------------------------------------- */
static int32_t query_key(const dataset_t *d, int32_t q, int32_t base);
static void dataset_set(dataset_t *d, const float *X, int64_t n, int32_t F, const float *labels,
                        const int32_t *qoff, int32_t q, const int32_t *qkey, int copyX)
{
    d->n = n; d->q = q; d->X = X;
    d->labels = (float *)malloc(sizeof(float) * (size_t)(n ? n : 1));
    memcpy(d->labels, labels, sizeof(float) * (size_t)n);
    d->qoff = (int32_t *)malloc(sizeof(int32_t) * (size_t)(q + 1));
    memcpy(d->qoff, qoff, sizeof(int32_t) * (size_t)(q + 1));
    d->qkey = NULL; d->ext_ideal = NULL; d->ext_rd = NULL;
    if (qkey) {
        d->qkey = (int32_t *)malloc(sizeof(int32_t) * (size_t)(q ? q : 1));
        memcpy(d->qkey, qkey, sizeof(int32_t) * (size_t)q);
    }
    d->Xown = NULL;
    if (copyX) {
        d->Xown = (float *)malloc(sizeof(float) * (size_t)((n > 0 && F > 0) ? n * F : 1));
        memcpy(d->Xown, X, sizeof(float) * (size_t)(n * F));
        d->X = d->Xown;
    }
}

ro_trainer *ro_create(const ro_params *p, const float *X, int64_t n_docs, int32_t n_features,
                      const float *labels, const int32_t *qoff, int32_t n_queries, const int32_t *feature_ids,
                      const int32_t *qkey)
{
    ro_trainer *t = (ro_trainer *)calloc(1, sizeof(ro_trainer));
    t->p = *p;
    t->F = n_features;
    t->feature_ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_features);
    for (int32_t f = 0; f < n_features; f++) t->feature_ids[f] = feature_ids ? feature_ids[f] : f + 1;
    dataset_set(&t->tr, X, n_docs, n_features, labels, qoff, n_queries, qkey, 0);
    /* column-major copy so that later stages never touch the caller's X */
    t->Xcol = (float *)malloc(sizeof(float) * (size_t)((n_docs > 0 && n_features > 0) ? n_docs * n_features : 1));
    for (int64_t k = 0; k < n_docs; k++)
        for (int32_t f = 0; f < n_features; f++) t->Xcol[(int64_t)f * n_docs + k] = X[k * n_features + f];
    t->tr.X = NULL;
    pool_init(&t->pool, p->n_threads);
    t->bestModelOnValidation = 2147483647 - 2;          /* LambdaMART.java:50 */
    t->bestScoreOnValidationData = 0.0;                 /* learning/Ranker.java:43 */
    return t;
}

void ro_set_validation(ro_trainer *t, const float *X, int64_t n_docs, const float *labels, const int32_t *qoff,
                       int32_t n_queries, const int32_t *qkey)
{
    dataset_set(&t->va, X, n_docs, t->F, labels, qoff, n_queries, qkey, 1);
    t->has_valid = 1;
}

void ro_set_external(ro_trainer *t, int validation, const double *ideal, const int32_t *rel_count)
{
    dataset_t *d = validation ? &t->va : &t->tr;
    free(d->ext_ideal); free(d->ext_rd); d->ext_ideal = NULL; d->ext_rd = NULL;
    if (ideal) { d->ext_ideal = (double *)malloc(sizeof(double) * (size_t)(d->q + 1)); memcpy(d->ext_ideal, ideal, sizeof(double) * (size_t)d->q); }
    if (rel_count) { d->ext_rd = (int32_t *)malloc(sizeof(int32_t) * (size_t)(d->q + 1)); memcpy(d->ext_rd, rel_count, sizeof(int32_t) * (size_t)d->q); }
}

static void free_kept(kept_tree *k)
{
    free(k->feature); free(k->threshold); free(k->left); free(k->right); free(k->output);
}

void ro_destroy(ro_trainer *t)
{
    if (!t) return;
    pool_free(&t->pool);
    for (int32_t f = 0; f < t->F; f++) {
        if (t->thr) free(t->thr[f]);
        if (t->bins) free(t->bins[f]);
    }
    free(t->thr); free(t->nthr); free(t->bins);
    free(t->root.sum); free(t->root.count);
    free(t->modelScores); free(t->pseudoResponses); free(t->weights); free(t->validScores);
    free(t->Xcol);
    free(t->tr.labels); free(t->tr.qoff); free(t->tr.qkey); free(t->tr.ext_ideal); free(t->tr.ext_rd);
    free(t->va.labels); free(t->va.qoff); free(t->va.qkey); free(t->va.Xown); free(t->va.ext_ideal); free(t->va.ext_rd);
    free(t->ck); free(t->cv); free(t->cu);
    for (int32_t i = 0; i < t->n_trees; i++) free_kept(&t->trees[i]);
    free(t->trees);
    free(t->trace_f); free(t->trace_t); free(t->trace_nn); free(t->trace_nl); free(t->trace_S);
    free(t->feature_ids);
    free(t);
}

/* ------------------------------------------------------------------------- */
/* init()  learning/tree/LambdaMART.java:68-166                                */
/* ------------------------------------------------------------------------- */
typedef struct { ro_trainer *t; int32_t **sortedIdx; } init_ctx;

/* sortSamplesByFeature (:417-424, :520-524): stable ascending sort of the float
 * feature values widened to double. */
static void init_sort_chunk(void *c_, int32_t fs, int32_t fe, int32_t worker)
{
    (void)worker;
    init_ctx *c = (init_ctx *)c_;
    ro_trainer *t = c->t;
    const int32_t N = (int32_t)t->tr.n;
    double *score = (double *)malloc(sizeof(double) * (size_t)(N ? N : 1));
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * N + 2));
    for (int32_t f = fs; f <= fe; f++) {
        const float *col = t->Xcol + (int64_t)f * N;
        for (int32_t i = 0; i < N; i++) score[i] = col[i];
        c->sortedIdx[f] = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
        sort_idx(score, 0, N, 1, c->sortedIdx[f], tmp, tmp + N);
    }
    free(score); free(tmp);
}

/* FeatureHistogram.construct(samples, labels, sortedIdx, thresholds, start, end)
 * (learning/tree/FeatureHistogram.java:75-112). labels are all 0 at init. */
static void init_hist_chunk(void *c_, int32_t fs, int32_t fe, int32_t worker)
{
    (void)worker;
    init_ctx *c = (init_ctx *)c_;
    ro_trainer *t = c->t;
    const int32_t N = (int32_t)t->tr.n;
    for (int32_t i = fs; i <= fe; i++) {
        const int32_t *idx = c->sortedIdx[i];
        const float *col = t->Xcol + (int64_t)i * N;
        const float *threshold = t->thr[i];
        const int32_t T = t->nthr[i];
        double sumLeft = 0;
        double *sumLabel = t->root.sum + (int64_t)i * t->TS;
        int32_t *cnt = t->root.count + (int64_t)i * t->TS;
        int32_t *stMap = (int32_t *)calloc((size_t)(N ? N : 1), sizeof(int32_t));
        int32_t last = -1;
        for (int32_t tt = 0; tt < T; tt++) {
            int32_t j = last + 1;
            for (; j < N; j++) {
                const int32_t k = idx[j];
                if (col[k] > threshold[tt]) break;            /* :94 float compare */
                sumLeft += t->pseudoResponses[k];
                if (i == 0) {
                    t->root.sumResponse += t->pseudoResponses[k];
                    t->root.sqSumResponse += t->pseudoResponses[k] * t->pseudoResponses[k];
                }
                stMap[k] = tt;
            }
            last = j - 1;
            sumLabel[tt] = sumLeft;
            cnt[tt] = last + 1;
        }
        t->bins[i] = stMap;
    }
}

void ro_init(ro_trainer *t)
{
    const int32_t N = (int32_t)t->tr.n, F = t->F;
    t->modelScores = (double *)calloc((size_t)(N ? N : 1), sizeof(double));       /* :78,:86 */
    t->pseudoResponses = (double *)calloc((size_t)(N ? N : 1), sizeof(double));
    t->weights = (double *)calloc((size_t)(N ? N : 1), sizeof(double));

    int32_t maxq = 1;
    for (int32_t q = 0; q < t->tr.q; q++) { int32_t n = t->tr.qoff[q + 1] - t->tr.qoff[q]; if (n > maxq) maxq = n; }
    for (int32_t q = 0; q < t->va.q; q++) { int32_t n = t->va.qoff[q + 1] - t->va.qoff[q]; if (n > maxq) maxq = n; }
    t->maxq = maxq;
    disc_reserve(maxq + 2);
    cache_init(t, t->tr.q + t->va.q);
    t->next_anon_key = 0;
    /* -qrel: NDCGScorer.loadExternalRelevanceJudgment puts the file's ideal DCGs into idealGains BEFORE any list is scored (:50-96), so
     * lists with such a qid never compute their own (the scorer object is shared by training and validation, Evaluator.java:580-591) */
    for (int32_t q = 0; q < t->tr.q; q++) if (t->tr.ext_ideal && t->tr.ext_ideal[q] == t->tr.ext_ideal[q]) cache_put(t, query_key(&t->tr, q, 0), t->tr.ext_ideal[q]);
    for (int32_t q = 0; q < t->va.q; q++) if (t->va.ext_ideal && t->va.ext_ideal[q] == t->va.ext_ideal[q]) cache_put(t, query_key(&t->va, q, t->tr.q), t->va.ext_ideal[q]);

    /* sort samples by each feature (:94-105) */
    init_ctx c; c.t = t;
    c.sortedIdx = (int32_t **)calloc((size_t)F, sizeof(int32_t *));
    pool_run(&t->pool, F, init_sort_chunk, &c);

    /* candidate thresholds (:108-150), serial */
    t->thr = (float **)calloc((size_t)F, sizeof(float *));
    t->nthr = (int32_t *)calloc((size_t)F, sizeof(int32_t));
    float *values = (float *)malloc(sizeof(float) * (size_t)(N ? N : 1));
    const int32_t nThreshold = t->p.n_threshold;
    int32_t TS = 1;
    for (int32_t f = 0; f < F; f++) {
        const float *col = t->Xcol + (int64_t)f * N;
        const int32_t *sidx = c.sortedIdx[f];
        int32_t nv = 0;
        float fmax = -INFINITY, fmin = FLT_MAX;
        for (int32_t i = 0; i < N; i++) {
            const float fv = col[sidx[i]];
            values[nv++] = fv;
            if (fmax < fv) fmax = fv;
            if (fmin > fv) fmin = fv;
            int32_t j = i + 1;
            while (j < N) { if (col[sidx[j]] > fv) break; j++; }   /* :126-131 */
            i = j - 1;
        }
        if (nv <= nThreshold || nThreshold == -1) {                /* :135-140 */
            t->thr[f] = (float *)malloc(sizeof(float) * (size_t)(nv + 1));
            memcpy(t->thr[f], values, sizeof(float) * (size_t)nv);
            t->thr[f][nv] = FLT_MAX;
            t->nthr[f] = nv + 1;
        } else {                                                   /* :141-149 */
            const float step = fabsf(fmax - fmin) / (float)nThreshold;
            t->thr[f] = (float *)malloc(sizeof(float) * (size_t)(nThreshold + 1));
            t->thr[f][0] = fmin;
            for (int32_t j = 1; j < nThreshold; j++) t->thr[f][j] = t->thr[f][j - 1] + step;
            t->thr[f][nThreshold] = FLT_MAX;
            t->nthr[f] = nThreshold + 1;
        }
        if (t->nthr[f] > TS) TS = t->nthr[f];
    }
    free(values);
    t->TS = TS;

    if (t->has_valid) t->validScores = (double *)calloc((size_t)(t->va.n ? t->va.n : 1), sizeof(double)); /* :152-158 */

    /* feature histogram (:161-162) */
    t->root.sum = (double *)calloc((size_t)F * (size_t)TS, sizeof(double));
    t->root.count = (int32_t *)calloc((size_t)F * (size_t)TS, sizeof(int32_t));
    t->root.owns_arrays = 1;
    t->root.sumResponse = 0; t->root.sqSumResponse = 0;
    t->bins = (int32_t **)calloc((size_t)F, sizeof(int32_t *));
    pool_run(&t->pool, F, init_hist_chunk, &c);

    for (int32_t f = 0; f < F; f++) free(c.sortedIdx[f]);          /* :164 */
    free(c.sortedIdx);
    free(t->Xcol); t->Xcol = NULL;
}

/* ------------------------------------------------------------------------- */
/* computePseudoResponses  LambdaMART.java:331-396                             */
/* ------------------------------------------------------------------------- */
static int32_t query_key(const dataset_t *d, int32_t q, int32_t base) { return d->qkey ? d->qkey[q] : base + q; }

typedef struct { ro_trainer *t; } lam_ctx;
static void lambda_chunk(void *c_, int32_t qs, int32_t qe, int32_t worker)
{
    (void)worker;
    ro_trainer *t = ((lam_ctx *)c_)->t;
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(4 * t->maxq + 4));
    for (int32_t q = qs; q <= qe; q++) {
        const int32_t cur = t->tr.qoff[q], n = t->tr.qoff[q + 1] - cur;
        g_ext_rd = t->tr.ext_rd ? t->tr.ext_rd[q] : -1;
        query_lambdas(t, t->modelScores, t->tr.labels, cur, n, t->p.metric_k, query_key(&t->tr, q, 0), -1.0,
                      t->pseudoResponses, t->weights, buf, buf + n, buf + 3 * n + 2);
    }
    free(buf);
}

void ro_compute_lambdas(ro_trainer *t)
{
    const int32_t N = (int32_t)t->tr.n;
    if (t->p.ranker == RO_RANKER_MART) {                          /* MART.computePseudoResponses, learning/tree/MART.java:47-51 */
        for (int32_t i = 0; i < N; i++) t->pseudoResponses[i] = t->tr.labels[i] - t->modelScores[i];
        return;
    }
    for (int32_t i = 0; i < N; i++) { t->pseudoResponses[i] = 0.0F; t->weights[i] = 0; }  /* :332-333 */
    lam_ctx c = { t };
    pool_run(&t->pool, t->tr.q, lambda_chunk, &c);
}

/* ------------------------------------------------------------------------- */
/* FeatureHistogram.update  FeatureHistogram.java:114-146                      */
/* ------------------------------------------------------------------------- */
static void hist_update_chunk(void *c_, int32_t fs, int32_t fe, int32_t worker)
{
    (void)worker;
    ro_trainer *t = (ro_trainer *)c_;
    const int32_t N = (int32_t)t->tr.n, TS = t->TS;
    const double *labels = t->pseudoResponses;
    hist_t *h = &t->root;
    for (int32_t f = fs; f <= fe; f++) memset(h->sum + (int64_t)f * TS, 0, sizeof(double) * (size_t)t->nthr[f]);
    for (int32_t k = 0; k < N; k++) {
        for (int32_t f = fs; f <= fe; f++) {
            const int32_t tt = t->bins[f][k];
            h->sum[(int64_t)f * TS + tt] += labels[k];
            if (f == 0) {
                h->sumResponse += labels[k];
                h->sqSumResponse += labels[k] * labels[k];
            }
        }
    }
    for (int32_t f = fs; f <= fe; f++) {
        double *s = h->sum + (int64_t)f * TS;
        for (int32_t tt = 1; tt < t->nthr[f]; tt++) s[tt] += s[tt - 1];
    }
}

static void hist_update(ro_trainer *t)
{
    t->root.sumResponse = 0;
    t->root.sqSumResponse = 0;
    pool_run(&t->pool, t->F, hist_update_chunk, t);
}

/* ------------------------------------------------------------------------- */
/* FeatureHistogram.construct(parent, soi, labels)  :148-195  (left child)     */
/* ------------------------------------------------------------------------- */
typedef struct { ro_trainer *t; hist_t *h; const int32_t *soi; int32_t n; } cons_ctx;
static void hist_construct_chunk(void *c_, int32_t fs, int32_t fe, int32_t worker)
{
    (void)worker;
    cons_ctx *c = (cons_ctx *)c_;
    ro_trainer *t = c->t;
    hist_t *h = c->h;
    const int32_t TS = t->TS;
    const double *labels = t->pseudoResponses;
    for (int32_t i = 0; i < c->n; i++) {
        const int32_t k = c->soi[i];
        for (int32_t f = fs; f <= fe; f++) {
            const int32_t tt = t->bins[f][k];
            h->sum[(int64_t)f * TS + tt] += labels[k];
            h->count[(int64_t)f * TS + tt]++;
            if (f == 0) {
                h->sumResponse += labels[k];
                h->sqSumResponse += labels[k] * labels[k];
            }
        }
    }
    for (int32_t f = fs; f <= fe; f++) {
        double *s = h->sum + (int64_t)f * TS;
        int32_t *cn = h->count + (int64_t)f * TS;
        for (int32_t tt = 1; tt < t->nthr[f]; tt++) { s[tt] += s[tt - 1]; cn[tt] += cn[tt - 1]; }
    }
}

/* FeatureHistogram.construct(parent, leftSibling, reuseParent)  :197-234 */
typedef struct { ro_trainer *t; hist_t *h; const hist_t *parent, *left; } sub_ctx;
static void hist_subtract_chunk(void *c_, int32_t fs, int32_t fe, int32_t worker)
{
    (void)worker;
    sub_ctx *c = (sub_ctx *)c_;
    const int32_t TS = c->t->TS;
    for (int32_t f = fs; f <= fe; f++) {
        for (int32_t tt = 0; tt < c->t->nthr[f]; tt++) {
            const int64_t o = (int64_t)f * TS + tt;
            c->h->sum[o] = c->parent->sum[o] - c->left->sum[o];
            c->h->count[o] = c->parent->count[o] - c->left->count[o];
        }
    }
}

/* findBestSplit(usedFeatures, minLeafSupport, start, end)  :236-264 */
/* ---- seeded stand-in for the unseeded `new Random()` of FeatureHistogram.java:283-287 ------------------------------------
 * Drawing `size` features without replacement == taking the `size` smallest of F independent random keys, in key order. */
static inline uint64_t mix64(uint64_t x)
{   /* splitmix64 finaliser */
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
uint64_t ro_root_hash(uint64_t seed, int32_t tree) { return mix64(seed ^ mix64((uint64_t)(uint32_t)tree)); }
uint64_t ro_child_hash(uint64_t parent, int32_t side) { return mix64(parent + 1u + (uint64_t)(side != 0)); }
uint64_t ro_feature_key(uint64_t node_hash, int32_t f) { return mix64(node_hash ^ ((uint64_t)(uint32_t)(f + 1) * 0xA24BAED4963EE407ULL)); }
int32_t ro_feature_order(uint64_t node_hash, int32_t F, float rate, int32_t *out)
{
    const int32_t size = (int32_t)(rate * (float)F);                    /* (int) (samplingRate * features.length)  :274 */
    for (int32_t f = 0; f < F; f++) {
        const uint64_t kf = ro_feature_key(node_hash, f);
        int32_t r = 0;
        for (int32_t g = 0; g < F; g++) { const uint64_t kg = ro_feature_key(node_hash, g); r += (kg < kf) || (kg == kf && g < f); }
        if (r < size) out[r] = f;
    }
    return size;
}

typedef struct { int32_t featureIdx, thresholdIdx; double S; } cfg_t;
typedef struct { ro_trainer *t; const hist_t *h; const int32_t *used; cfg_t cfg[1024]; } scan_ctx;
static void scan_chunk(void *c_, int32_t fs, int32_t fe, int32_t worker)
{
    scan_ctx *c = (scan_ctx *)c_;
    const ro_trainer *t = c->t;
    const hist_t *h = c->h;
    const int32_t TS = t->TS, mls = t->p.min_leaf_support;
    cfg_t cfg = { -1, -1, -1.0 };
    const int32_t f0 = c->used ? c->used[fs] : fs;
    const int32_t totalCount = h->count[(int64_t)f0 * TS + t->nthr[f0] - 1];
    for (int32_t f = fs; f <= fe; f++) {
        const int32_t i = c->used ? c->used[f] : f;                /* usedFeatures[f]  :289-300 */
        for (int32_t tt = 0; tt < t->nthr[i]; tt++) {
            const int32_t countLeft = h->count[(int64_t)i * TS + tt];
            const int32_t countRight = totalCount - countLeft;
            if (countLeft < mls || countRight < mls) continue;
            const double sumLeft = h->sum[(int64_t)i * TS + tt];
            const double sumRight = h->sumResponse - sumLeft;
            const double S = sumLeft * sumLeft / countLeft + sumRight * sumRight / countRight;
            if (cfg.S < S) { cfg.S = S; cfg.featureIdx = i; cfg.thresholdIdx = tt; }
        }
    }
    c->cfg[worker] = cfg;
}

static hist_t *hist_new(const ro_trainer *t, int alloc)
{
    hist_t *h = (hist_t *)calloc(1, sizeof(hist_t));
    if (alloc) {
        h->sum = (double *)calloc((size_t)t->F * (size_t)t->TS, sizeof(double));
        h->count = (int32_t *)calloc((size_t)t->F * (size_t)t->TS, sizeof(int32_t));
        h->owns_arrays = 1;
    }
    return h;
}
static void hist_release(hist_t *h, const hist_t *root)
{
    if (!h || h == root) return;
    if (h->owns_arrays) { free(h->sum); free(h->count); }
    free(h);
}

static node_t *node_new(int32_t *samples, int32_t n, hist_t *hist, double deviance)
{
    node_t *s = (node_t *)calloc(1, sizeof(node_t));       /* Split(int[],hist,deviance,sumLabel) Split.java:58-64 */
    s->featureID = -1; s->samples = samples; s->n = n; s->hist = hist; s->deviance = deviance;
    return s;
}

static void trace_push(ro_trainer *t, int32_t f, int32_t tt, double S, int32_t nn, int32_t nl)
{
    if (t->trace_n == t->trace_cap) {
        t->trace_cap = t->trace_cap ? 2 * t->trace_cap : 64;
        t->trace_f = (int32_t *)realloc(t->trace_f, sizeof(int32_t) * (size_t)t->trace_cap);
        t->trace_t = (int32_t *)realloc(t->trace_t, sizeof(int32_t) * (size_t)t->trace_cap);
        t->trace_nn = (int32_t *)realloc(t->trace_nn, sizeof(int32_t) * (size_t)t->trace_cap);
        t->trace_nl = (int32_t *)realloc(t->trace_nl, sizeof(int32_t) * (size_t)t->trace_cap);
        t->trace_S = (double *)realloc(t->trace_S, sizeof(double) * (size_t)t->trace_cap);
    }
    t->trace_f[t->trace_n] = f; t->trace_t[t->trace_n] = tt; t->trace_S[t->trace_n] = S;
    t->trace_nn[t->trace_n] = nn; t->trace_nl[t->trace_n] = nl; t->trace_n++;
}

/* FeatureHistogram.findBestSplit(Split, labels, minLeafSupport)  :266-359 */
static int node_split(ro_trainer *t, node_t *sp)
{
    hist_t *h = sp->hist;
    if (sp->deviance >= 0.0 && sp->deviance <= 0.0) return 0;        /* :267-269 */

    scan_ctx *sc = (scan_ctx *)malloc(sizeof(scan_ctx));
    sc->t = t; sc->h = h; sc->used = NULL;
    int32_t nused = t->F;
    int32_t *order = NULL;
    if (t->p.feature_sampling_rate > 0.0f && t->p.feature_sampling_rate < 1.0f) {     /* :272-287, seeded (see ro_feature_order) */
        order = (int32_t *)malloc(sizeof(int32_t) * (size_t)t->F);
        nused = ro_feature_order(sp->ph, t->F, t->p.feature_sampling_rate, order);
        sc->used = order;
    }
    cfg_t best = { -1, -1, -1.0 };
    if (nused > 0) {
        int32_t used = pool_run(&t->pool, nused, scan_chunk, sc);
        for (int32_t w = 0; w < used; w++)                              /* :302-309 */
            if (best.S < sc->cfg[w].S) best = sc->cfg[w];
    }
    free(order);
    free(sc);
    if (best.S == -1) return 0;                                         /* :311-313 */

    const int32_t TS = t->TS;
    const double *bestHist = h->sum + (int64_t)best.featureIdx * TS;
    const int32_t *sampleCount = h->count + (int64_t)best.featureIdx * TS;
    const int32_t lastT = t->nthr[best.featureIdx] - 1;
    const int32_t cAll = sampleCount[lastT];
    const int32_t countLeft = sampleCount[best.thresholdIdx];
    const int32_t countRight = cAll - countLeft;
    (void)bestHist;

    int32_t *left = (int32_t *)malloc(sizeof(int32_t) * (size_t)(countLeft ? countLeft : 1));
    int32_t *right = (int32_t *)malloc(sizeof(int32_t) * (size_t)(countRight ? countRight : 1));
    int32_t l = 0, r = 0;
    const int32_t *bmap = t->bins[best.featureIdx];
    for (int32_t i = 0; i < sp->n; i++) {                               /* :334-341 */
        const int32_t k = sp->samples[i];
        if (bmap[k] <= best.thresholdIdx) left[l++] = k; else right[r++] = k;
    }

    hist_t *lh = hist_new(t, 1);                                        /* :343-344 */
    cons_ctx cc = { t, lh, left, l };
    lh->sumResponse = 0; lh->sqSumResponse = 0;
    pool_run(&t->pool, t->F, hist_construct_chunk, &cc);

    const int reuseParent = !sp->isRoot;                                /* :345-346 */
    hist_t *rh = hist_new(t, !reuseParent);
    rh->sumResponse = h->sumResponse - lh->sumResponse;                 /* :202-203 */
    rh->sqSumResponse = h->sqSumResponse - lh->sqSumResponse;
    if (reuseParent) { rh->sum = h->sum; rh->count = h->count; rh->owns_arrays = h->owns_arrays; h->owns_arrays = 0; }
    sub_ctx sb = { t, rh, h, lh };
    pool_run(&t->pool, t->F, hist_subtract_chunk, &sb);

    const double var = h->sqSumResponse - h->sumResponse * h->sumResponse / sp->n;          /* :348 */
    const double varLeft = lh->sqSumResponse - lh->sumResponse * lh->sumResponse / l;      /* :349 */
    const double varRight = rh->sqSumResponse - rh->sumResponse * rh->sumResponse / r;     /* :350 */

    trace_push(t, best.featureIdx, best.thresholdIdx, best.S, sp->n, l);

    sp->featureID = t->feature_ids[best.featureIdx];                    /* :352 */
    sp->threshold = t->thr[best.featureIdx][best.thresholdIdx];
    sp->deviance = var;
    sp->left = node_new(left, l, lh, varLeft);                          /* :353-354 */
    sp->right = node_new(right, r, rh, varRight);
    sp->left->ph = ro_child_hash(sp->ph, 0); sp->right->ph = ro_child_hash(sp->ph, 1);
    sp->n = (int32_t)sp->n;
    /* sp.clearSamples()  :356 */
    if (!sp->isRoot) free(sp->samples);
    sp->samples = NULL;
    hist_release(h, &t->root);
    sp->hist = NULL;
    return 1;
}

/* RegressionTree.insert  learning/tree/RegressionTree.java:147-157 */
typedef struct { node_t **a; int32_t n, cap; } queue_t;
static void queue_insert(queue_t *q, node_t *s)
{
    int32_t i = 0;
    while (i < q->n) { if (q->a[i]->deviance > s->deviance) i++; else break; }
    if (q->n == q->cap) { q->cap = q->cap ? 2 * q->cap : 64; q->a = (node_t **)realloc(q->a, sizeof(node_t *) * (size_t)q->cap); }
    memmove(q->a + i + 1, q->a + i, sizeof(node_t *) * (size_t)(q->n - i));
    q->a[i] = s; q->n++;
}

/* RegressionTree.fit  :58-87 */
static node_t *tree_fit(ro_trainer *t, int32_t *index)
{
    const int32_t N = (int32_t)t->tr.n;
    const int32_t nodes = t->p.n_leaves, mls = t->p.min_leaf_support;
    queue_t q = { NULL, 0, 0 };
    node_t *root = node_new(index, N, &t->root, (double)FLT_MAX);  /* :60 */
    root->isRoot = 1;
    root->ph = ro_root_hash(t->p.seed, t->round);
    t->trace_n = 0;
    if (node_split(t, root)) { queue_insert(&q, root->left); queue_insert(&q, root->right); }
    int32_t taken = 0;
    while ((nodes == -1 || taken + q.n < nodes) && q.n > 0) {
        node_t *leaf = q.a[0];
        memmove(q.a, q.a + 1, sizeof(node_t *) * (size_t)(q.n - 1));
        q.n--;
        if (leaf->n < 2 * mls) { taken++; continue; }
        if (!node_split(t, leaf)) taken++;
        else { queue_insert(&q, leaf->left); queue_insert(&q, leaf->right); }
    }
    free(q.a);
    return root;
}

static void collect_leaves(node_t *n, node_t **out, int32_t *cnt)
{
    if (n->featureID == -1) out[(*cnt)++] = n;                     /* Split.java:106-113 */
    else { collect_leaves(n->left, out, cnt); collect_leaves(n->right, out, cnt); }
}
static int32_t count_nodes(const node_t *n) { return n->featureID == -1 ? 1 : 1 + count_nodes(n->left) + count_nodes(n->right); }

static void tree_free(ro_trainer *t, node_t *n, int is_root)
{
    if (!n) return;
    tree_free(t, n->left, 0); tree_free(t, n->right, 0);
    if (!is_root) free(n->samples);
    hist_release(n->hist, &t->root);
    free(n);
}

/* Split.eval  learning/tree/Split.java:115-125 on a flattened tree */
static inline float flat_eval(const kept_tree *k, const int32_t *fid2col, const float *row)
{
    int32_t n = 0;
    while (k->feature[n] != -1) {
        const float v = row[fid2col[k->feature[n]]];
        n = (v <= k->threshold[n]) ? k->left[n] : k->right[n];
    }
    return k->output[n];
}

static int32_t flatten(const node_t *n, kept_tree *k, ro_tree *out, int32_t *pos)
{
    const int32_t me = (*pos)++;
    k->feature[me] = n->featureID; k->threshold[me] = n->threshold; k->output[me] = (float)n->avgLabel;
    k->left[me] = k->right[me] = -1;
    if (out && me < out->cap) { if (out->deviance) out->deviance[me] = n->deviance; if (out->count) out->count[me] = n->n; }
    if (n->featureID != -1) {
        k->left[me] = flatten(n->left, k, out, pos);
        k->right[me] = flatten(n->right, k, out, pos);
    }
    return me;
}

/* per-round train/validation metric with float accumulation
 * (LambdaMART.java:442-483, :485-518) */
static float model_score(ro_trainer *t, const dataset_t *d, const double *scores, int32_t keybase)
{
    float s = 0;
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(4 * t->maxq + 4));
    for (int32_t q = 0; q < d->q; q++) {
        const int32_t cur = d->qoff[q], n = d->qoff[q + 1] - cur;
        int32_t *idx = buf, *tmp = buf + n, *rel = buf + 3 * n + 2;
        sort_idx(scores, cur, n, 0, idx, tmp, tmp + n);           /* rank(), :432-440 */
        g_ext_rd = d->ext_rd ? d->ext_rd[q] : -1;
        const double sc = query_score(t, t->p.metric, d->labels, idx, n, t->p.metric_k, query_key(d, q, keybase), rel, tmp);
        s = (float)((double)s + sc);                              /* float s; s += double  :479 */
    }
    free(buf);
    s = s / (float)d->q;                                          /* :470 */
    return s;
}

static int32_t *build_fid2col(const ro_trainer *t, int32_t *maxfid_out)
{
    int32_t maxfid = 0;
    for (int32_t f = 0; f < t->F; f++) if (t->feature_ids[f] > maxfid) maxfid = t->feature_ids[f];
    int32_t *m = (int32_t *)calloc((size_t)maxfid + 1, sizeof(int32_t));
    for (int32_t f = 0; f < t->F; f++) m[t->feature_ids[f]] = f;
    if (maxfid_out) *maxfid_out = maxfid;
    return m;
}

int ro_round(ro_trainer *t, ro_tree *out, float *train_metric, float *valid_metric)
{
    const int32_t N = (int32_t)t->tr.n;
    const int32_t m = t->round;

    ro_compute_lambdas(t);                                        /* :187 */
    hist_update(t);                                               /* :190 */

    int32_t *index = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));  /* RegressionTree ctor :49-52 */
    for (int32_t i = 0; i < N; i++) index[i] = i;
    node_t *root = tree_fit(t, index);                            /* :193-194 */

    const int32_t nn = count_nodes(root);
    node_t **leaves = (node_t **)malloc(sizeof(node_t *) * (size_t)nn);
    int32_t nl = 0;
    collect_leaves(root, leaves, &nl);

    /* updateTreeOutput  :398-415 */
    for (int32_t i = 0; i < nl; i++) {
        float s1 = 0.0F, s2 = 0.0F;
        node_t *s = leaves[i];
        const int32_t *idx = (s == root) ? index : s->samples;
        for (int32_t j = 0; j < s->n; j++) {
            const int32_t k = idx[j];
            s1 = (float)((double)s1 + t->pseudoResponses[k]);
            s2 = (float)((double)s2 + t->weights[k]);
        }
        if (t->p.ranker == RO_RANKER_MART) s->avgLabel = (double)(s1 / s->n);   /* MART.updateTreeOutput :54-65: float / int */
        else if (s2 == 0) s->avgLabel = 0; else s->avgLabel = (double)(s1 / s2);
    }
    /* score update  :203-210 */
    for (int32_t i = 0; i < nl; i++) {
        node_t *s = leaves[i];
        const int32_t *idx = (s == root) ? index : s->samples;
        for (int32_t j = 0; j < s->n; j++) t->modelScores[idx[j]] += (double)t->p.learning_rate * s->avgLabel;
    }

    /* ensemble.add(rt, learningRate)  :197 */
    if (t->n_trees == t->cap_trees) {
        t->cap_trees = t->cap_trees ? 2 * t->cap_trees : 64;
        t->trees = (kept_tree *)realloc(t->trees, sizeof(kept_tree) * (size_t)t->cap_trees);
    }
    kept_tree *k = &t->trees[t->n_trees++];
    k->n_nodes = nn;
    k->feature = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    k->threshold = (float *)malloc(sizeof(float) * (size_t)nn);
    k->left = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    k->right = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    k->output = (float *)malloc(sizeof(float) * (size_t)nn);
    int32_t pos = 0;
    flatten(root, k, out, &pos);
    if (out) {
        out->n_nodes = nn;
        for (int32_t i = 0; i < nn && i < out->cap; i++) {
            out->feature[i] = k->feature[i]; out->threshold[i] = k->threshold[i];
            out->left[i] = k->left[i]; out->right[i] = k->right[i]; out->output[i] = k->output[i];
        }
    }
    free(leaves);
    tree_free(t, root, 1);
    free(index);

    const float tm = model_score(t, &t->tr, t->modelScores, 0);   /* :216 */
    if (train_metric) *train_metric = tm;

    if (t->has_valid) {                                           /* :228-244 */
        int32_t *fid2col = build_fid2col(t, NULL);
        for (int64_t i = 0; i < t->va.n; i++)
            t->validScores[i] += (double)t->p.learning_rate * (double)flat_eval(k, fid2col, t->va.X + i * t->F);
        free(fid2col);
        const float vs = model_score(t, &t->va, t->validScores, t->tr.q);
        if (valid_metric) *valid_metric = vs;
        const double score = vs;
        if (score > t->bestScoreOnValidationData) {
            t->bestScoreOnValidationData = score;
            t->bestModelOnValidation = t->n_trees - 1;
        }
    }
    t->round = m + 1;
    return (m - t->bestModelOnValidation > t->p.early_stop) ? 1 : 0;    /* :248 */
}

void ro_predict(const ro_trainer *t, const float *X, int64_t n_docs, float *out)
{
    int32_t *fid2col = build_fid2col(t, NULL);
    for (int64_t i = 0; i < n_docs; i++) {
        float s = 0;                                              /* Ensemble.eval  Ensemble.java:110-116 */
        for (int32_t j = 0; j < t->n_trees; j++)
            s = (float)((double)s + (double)flat_eval(&t->trees[j], fid2col, X + i * t->F) * (double)t->p.learning_rate);
        out[i] = s;
    }
    free(fid2col);
}

/* Ensemble.eval (learning/tree/Ensemble.java:110-116, Split.eval learning/tree/Split.java:115-125) of a flat model
 * (n_trees x maxn nodes, feature < 0 = leaf; feature f is read from column f of the row, columns >= stride read 0) on
 * n rows, documents split over n_threads like a caller scoring lists in parallel would.  CPU baseline of config c4. */
typedef struct { int32_t nt, maxn; const int32_t *feature, *left, *right; const float *thr, *outv, *w; const float *X; int64_t n;
                 int32_t stride; float *res; int32_t n_threads, id; } evm_arg;
static void *evm_main(void *a_)
{
    evm_arg *a = (evm_arg *)a_;
    const int64_t per = (a->n + a->n_threads - 1) / a->n_threads, i0 = per * a->id, i1 = (i0 + per < a->n) ? i0 + per : a->n;
    for (int64_t i = i0; i < i1; i++) {
        const float *row = a->X + i * a->stride;
        float s = 0;
        for (int32_t t = 0; t < a->nt; t++) {
            const size_t o = (size_t)t * a->maxn;
            int32_t nd = 0;
            while (a->feature[o + nd] >= 0) {
                const int32_t f = a->feature[o + nd];
                const float v = f < a->stride ? row[f] : 0.0f;
                nd = (v <= a->thr[o + nd]) ? a->left[o + nd] : a->right[o + nd];
            }
            s = (float)((double)s + (double)a->outv[o + nd] * (double)a->w[t]);
        }
        a->res[i] = s;
    }
    return NULL;
}
void ro_eval_flat_model(int32_t n_trees, int32_t maxn, const int32_t *feature, const float *thr, const int32_t *left,
                        const int32_t *right, const float *outv, const float *weight, const float *X, int64_t n,
                        int32_t stride, int32_t n_threads, float *res)
{
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    evm_arg *args = (evm_arg *)malloc(sizeof(evm_arg) * (size_t)n_threads);
    for (int32_t k = 0; k < n_threads; k++) {
        args[k] = (evm_arg){ n_trees, maxn, feature, left, right, thr, outv, weight, X, n, stride, res, n_threads, k };
        pthread_create(&th[k], NULL, evm_main, &args[k]);
    }
    for (int32_t k = 0; k < n_threads; k++) pthread_join(th[k], NULL);
    free(th); free(args);
}

/* scorer.score(rank(samples))  LambdaMART.java:259, Ranker.java:88-103,
 * MetricScorer.java:46-52 (double mean). Needs the raw rows again. */
static double final_score(ro_trainer *t, const dataset_t *d, const float *X, int32_t keybase)
{
    float *ev = (float *)malloc(sizeof(float) * (size_t)(d->n ? d->n : 1));
    ro_predict(t, X, d->n, ev);
    double *sc = (double *)malloc(sizeof(double) * (size_t)(d->n ? d->n : 1));
    for (int64_t i = 0; i < d->n; i++) sc[i] = ev[i];
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(4 * t->maxq + 4));
    double score = 0.0;
    for (int32_t q = 0; q < d->q; q++) {
        const int32_t cur = d->qoff[q], n = d->qoff[q + 1] - cur;
        int32_t *idx = buf, *tmp = buf + n, *rel = buf + 3 * n + 2;
        sort_idx(sc, cur, n, 0, idx, tmp, tmp + n);
        g_ext_rd = d->ext_rd ? d->ext_rd[q] : -1;
        score += query_score(t, t->p.metric, d->labels, idx, n, t->p.metric_k, query_key(d, q, keybase), rel, tmp);
    }
    free(buf); free(sc); free(ev);
    return score / d->q;
}

void ro_finish_with_rows(ro_trainer *t, const float *Xtrain, double *train_score, double *valid_score)
{
    while (t->n_trees > t->bestModelOnValidation + 1) {           /* :254-256 */
        free_kept(&t->trees[t->n_trees - 1]);
        t->n_trees--;
    }
    if (train_score) *train_score = final_score(t, &t->tr, Xtrain, 0);
    if (t->has_valid) {
        const double v = final_score(t, &t->va, t->va.X, t->tr.q);
        t->bestScoreOnValidationData = v;                         /* :263 */
        if (valid_score) *valid_score = v;
    }
}

/* ------------------------------------------------------------------------- */
/* accessors                                                                   */
/* ------------------------------------------------------------------------- */
int32_t ro_n_bins(const ro_trainer *t, int32_t f) { return t->nthr[f]; }
const float *ro_thresholds(const ro_trainer *t, int32_t f) { return t->thr[f]; }
const int32_t *ro_bins(const ro_trainer *t, int32_t f) { return t->bins[f]; }
const int32_t *ro_root_count(const ro_trainer *t, int32_t f) { return t->root.count + (int64_t)f * t->TS; }
const double *ro_root_sum(const ro_trainer *t, int32_t f) { return t->root.sum + (int64_t)f * t->TS; }
const double *ro_lambdas(const ro_trainer *t) { return t->pseudoResponses; }
const double *ro_weights(const ro_trainer *t) { return t->weights; }
const double *ro_scores(const ro_trainer *t) { return t->modelScores; }
const double *ro_valid_scores(const ro_trainer *t) { return t->validScores; }
int32_t ro_trees_kept(const ro_trainer *t) { return t->n_trees; }
int32_t ro_best_valid_round(const ro_trainer *t) { return t->bestModelOnValidation; }
double ro_best_valid_score(const ro_trainer *t) { return t->bestScoreOnValidationData; }
void ro_hist_update_only(ro_trainer *t) { hist_update(t); }

int32_t ro_last_split_trace(const ro_trainer *t, int32_t cap, int32_t *fidx, int32_t *tidx, double *S,
                            int32_t *n_node, int32_t *n_left)
{
    for (int32_t i = 0; i < t->trace_n && i < cap; i++) {
        if (fidx) fidx[i] = t->trace_f[i];
        if (tidx) tidx[i] = t->trace_t[i];
        if (S) S[i] = t->trace_S[i];
        if (n_node) n_node[i] = t->trace_nn[i];
        if (n_left) n_left[i] = t->trace_nl[i];
    }
    return t->trace_n;
}

/* kept tree i, copied into caller arrays (for model comparison after finish) */
int32_t ro_get_tree(const ro_trainer *t, int32_t i, ro_tree *out)
{
    if (i < 0 || i >= t->n_trees) return -1;
    const kept_tree *k = &t->trees[i];
    out->n_nodes = k->n_nodes;
    for (int32_t j = 0; j < k->n_nodes && j < out->cap; j++) {
        out->feature[j] = k->feature[j]; out->threshold[j] = k->threshold[j];
        out->left[j] = k->left[j]; out->right[j] = k->right[j]; out->output[j] = k->output[j];
    }
    return k->n_nodes;
}
