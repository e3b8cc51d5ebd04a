// rl_trainer.hip -- host orchestration + C ABI of librlhip.so (see include/rlhip.h).
//
// One handle = one GPU = one HIP stream.  After rl_init everything a boosting round needs is resident in
// HBM; a round is a fixed sequence of kernel launches with NO host synchronisation inside it: the tree is
// grown by device-side state (TreeState / NodeRec), the host only enqueues "one more split step" L-1 times.
//
// There is no CPU fallback anywhere in this file: without a gfx950 device rl_create fails with
// RL_ERR_NO_DEVICE.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>

#include "rl_internal.h"
#include "rl_device.h"
#include "rl_wave.h"
#include "rl_kernels_init.inc"
#include "rl_chain.inc"
#include "rl_kernels_round.inc"
#include "rl_step2.inc"
#include "rl_java_order.inc"
#include "rl_csc.inc"
#include "rl_tie.inc"
#include "rl_membench.inc"
#include "rl_dist.inc"
#include "rl_model.h"

namespace rl {

static thread_local std::string g_err;
static double g_err_max = 16.0;      // ERRScorer.MAX: a process-wide static in the reference as well (metric/ERRScorer.java:25)
void set_error(const std::string &msg) { g_err = msg; }
int fail(int code, const std::string &msg) { g_err = msg; return code; }

struct DevPool {
    std::vector<void *> ptrs;
    template <typename T> hipError_t alloc(T **p, size_t n)
    {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) { ptrs.push_back(q); *p = (T *)q; }
        return e;
    }
    void release(void *p)
    {
        for (auto &q : ptrs) if (q == p) { (void)hipFree(q); q = nullptr; }
    }
    ~DevPool() { for (void *q : ptrs) if (q) (void)hipFree(q); }
};

struct DataSet {
    int64_t N = 0; int32_t Q = 0;
    std::vector<float> labels; std::vector<int32_t> qoff, qkey; bool has_key = false;
    float *d_X = nullptr;          // row-major rows [N][F]
    int64_t rows_next = -1;        // chunked upload (rl_set_rows): next row expected; -1 = the rows came with rl_set_train / rl_set_validation
    float *d_labels = nullptr; int32_t *d_qoff = nullptr; double *d_ideal0 = nullptr, *d_ideal1 = nullptr;
    double *d_scores = nullptr, *d_ndcg = nullptr;
    double *d_ss = nullptr; float *d_sl = nullptr; int32_t *d_srel = nullptr, *d_sidx = nullptr, *d_docq = nullptr;   // ranked order (training set only)
    int32_t *d_aux_i = nullptr; double *d_aux_a = nullptr, *d_aux_b = nullptr;   // swapChange tables of MAP / ERR in ranked order
    // -qrel (rl_set_external_judgments): per list, the idealGains entry of its qid in the judgment file (NaN = none) and its relDocCount
    std::vector<double> ext_ideal; std::vector<int32_t> ext_rd; int32_t *d_ext_rd = nullptr;
    int32_t *d_qsmall = nullptr, *d_qbig = nullptr, *d_qtiny = nullptr; int32_t n_small = 0, n_big = 0, n_tiny = 0, max_big = 0, n_small_long = 0; bool all_small = false;
    int32_t *d_qhuge = nullptr, *d_relscratch = nullptr; int32_t n_huge = 0;      // lists beyond kLambdaBlockCap documents (k_rank_huge)
    // queries by length class for the fused lambda kernel: <= 64, <= 128, <= 192 documents, longer (tiled by 256); a block is as
    // wide as its class, so short lists do not leave most of a block idle
    int32_t *d_qcls[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; int32_t n_qcls[5] = {0, 0, 0, 0, 0};     // [4]: <= 16 documents (k_lambda_tiny)
    int32_t maxq = 0;
};

struct TimingSlot { double ms = 0; int64_t launches = 0; double bytes = 0; };

}  // namespace rl

using namespace rl;

struct rl_trainer {
    rl_params p;
    int32_t F = 0;
    std::vector<int32_t> feature_ids;
    std::vector<int32_t> vcol;         // histogram (virtual) feature -> column of the row matrix; empty = identity (rl_init: tables beyond 4095 entries)
    DataSet tr, va;
    bool has_train = false, has_valid = false, inited = false, finished = false;
    hipStream_t stream = nullptr;
    // the per-round training metric (a float chain over the queries) is off the critical path of the next round: it runs
    // on a side stream between two events (single-GPU runs without a validation set)
    hipStream_t side = nullptr; hipEvent_t ev_ranked = nullptr, ev_metric = nullptr; bool side_pending = false;
    // the lambda kernels of the list-length classes are independent: three of them run on streams of their own beside the main one, so that the tail
    // of one class (its last blocks) overlaps the next class instead of idling the chip (RLHIP_LAMBDA_STREAMS=0: one after the other)
    hipStream_t lam_s[3] = {nullptr, nullptr, nullptr}; hipEvent_t ev_lam_fork = nullptr, ev_lam_join[3] = {nullptr, nullptr, nullptr}; bool lam_streams = false;
    int32_t lam_side = 1; bool lam_compact = false;      // RLHIP_LAMBDA_SIDE / RLHIP_LAMBDA_COMPACT, read when the trainer is created (they were process-wide statics)
    DevPool pool;
    Ctx ctx;
    EnsTree ens;
    int32_t L_eff = 0;          // leaf budget in force: n_leaves, or floor(N / min_leaf_support) for -leaf -1
    int32_t *d_tie_flag = nullptr;                    // sharded tie-break: the ranks' common out-of-memory decision (resolve_ties)
    int64_t sp_entries = 0; int32_t sp_cols = 0;      // sparse-column path of the root pass (rl_csc.inc)
    int32_t cr_groups = 0; double cr_entries = 0, cr_overflow = 0;          // compact rows (groups that use them; of the child passes (k_compact_rows): entries outside the mode bins, rows that need the dense fallback
    double err_max = 16.0;      // ERRScorer.MAX when the trainer was created (rl_set_err_max)
    int32_t round = 0;          // rounds enqueued so far
    // growth progress reported by the device (Ctx::progress): the host keeps at most `step_ahead` growth steps in flight and
    // stops enqueuing steps of a finished tree; 0 = enqueue all L-1 steps blindly
    unsigned long long *h_progress = nullptr; uint32_t tree_seq = 0; int32_t step_ahead = 1;       // (1: c2 409.5 -> 411.7 rounds/s against 3, profiles/r05g_ab_step_ahead_c2.txt -- fewer empty steps behind a finished tree)
    int32_t dist_ahead = 1;     // sharded runs: growth steps enqueued beyond the last one whose bookkeeping the host has seen -- 0: every enqueued step has work (an empty
                                // step still costs its all-reduce on every rank); RLHIP_DIST_STEP_AHEAD
    unsigned long long chain_seq = 0; std::vector<void *> pinned;     // chain pass tags; pinned words of the chains (freed in rl_destroy)
    int32_t synced_rounds = 0;
    long long tie_stalls = 0, tie_nodes = 0, tie_chain_nodes = 0, tie_chain_docs = 0;      // lazy tie-break (rl_tie.inc): resolutions run, nodes resolved, chain nodes / documents summed
    long long tie_batches = 0;      // of the resolutions, the batched ones at the end of a tree (deferred ties)
    bool fin_split = false;      // wide data: k_hist_finish_wide + k_select instead of the fused finish (rl_init)
    bool sel2_wide = true;       // k_select2<true> on data with 161 .. 768 histogram features (RLHIP_SELECT2_WIDE=0: k_select)
    bool step2 = true;           // k_fin2 (+ k_select2) instead of the fused finish / bookkeeping kernel (rl_step2.inc; RLHIP_STEP2=0: the round-4 kernels)
    long long tie_phase_us[6] = {0, 0, 0, 0, 0, 0};   // RLHIP_TIE_PROF: host microseconds per phase of resolve_ties (printed by rl_destroy)
    long long chain_calls[2] = {0, 0}, chain_repairs[2] = {0, 0}, chain_timeouts = 0, chain_wait_us = 0;      // RLHIP_CHAIN_PROF: float-chain evaluations [hinted, blind], repair passes enqueued, progress-word time-outs, host microseconds spent waiting for a stitch
    long long tie_regrown = 0;      // trees grown a second time because a deferred tie over several features hid two different cuts (k_tie_verify)
    long long tie_us = 0, tie_spec_segs = 0, tie_spec_miss = 0, tie_spec_serial = 0, tie_spec_repairs = 0;      // host time in resolve_ties; segments evaluated, window misses, serial segments, repair passes
    std::vector<int32_t> h_nthr; std::vector<char> tie_blob;
    void *tie_pin = nullptr; size_t tie_pin_cap = 0;                                          // pinned staging of its small device-to-host reads
    void *tie_buf = nullptr; size_t tie_cap = 0, tie_hint = 0;                                              // scratch arena of resolve_ties (only ever grows)
    int32_t n_kept = 0;         // trees kept after rollback (== round until rl_finish)
    int32_t best_round = 2147483647 - 2;     // LambdaMART.bestModelOnValidation  LambdaMART.java:50
    double best_score = 0.0;                 // Ranker.bestScoreOnValidationData  Ranker.java:43
    std::vector<float> h_metrics;            // [round][2]
    std::vector<HostTree> trees;             // host copies (pre-order), fetched lazily
    float *d_final_f = nullptr; double *d_final_d = nullptr, *d_mean = nullptr;
    ChainBufs leaf_chain, metric_chain;      // exact parallel float chains (rl_chain.inc)
    int32_t *d_seg_buf = nullptr;
    double2 *d_T = nullptr;                  // pair terms of the lambda computation [N][k]
    double *d_wmax = nullptr;                // per-block max |lambda| of the lambda launches
    float *d_vmetric = nullptr;
    // timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending[RL_KERNEL_COUNT_];
    std::vector<double> ev_bytes[RL_KERNEL_COUNT_];
    std::vector<hipEvent_t> ev_free;
    TimingSlot timing[RL_KERNEL_COUNT_];
    // distributed (rl_dist.inc)
    int32_t rank = 0, n_ranks = 1;
    std::unique_ptr<DistBackend> dist;
    std::vector<int32_t> all_N, all_Q;        // local sizes of every rank
    int64_t Nglobal = 0; int32_t Qglobal = 0, Qmax = 0;
    ChainBufs gchain;                          // float chains over the all-gathered leaf values
    double *d_gx = nullptr, *d_send = nullptr; int32_t *d_gls = nullptr; int32_t lsstride = 0; float *d_gres = nullptr;     // leaf-owner exchange: receive / send buffers
    int32_t *d_own = nullptr; long long *d_xtab = nullptr;     // owner of every leaf; pack / assemble offsets (rl_dist.inc LeafExchange)
    std::vector<int32_t> h_gls, h_own; std::vector<long long> h_xtab;
    // round 6, distributed float chains (rl_dist.inc "piece mode"; RLHIP_DIST_OWNER_CHAINS=1: the leaf-owner exchange instead)
    bool piece_chains = false; double *d_pc_loc = nullptr, *d_pc_all = nullptr, *d_pc_base = nullptr; uint32_t *d_ptab = nullptr, *d_gtab = nullptr, *d_res_loc = nullptr, *d_res_all = nullptr;
    CrossState xstate{nullptr, nullptr, nullptr}; long long piece_rounds = 0, piece_misses = 0; int32_t piece_force = 0;      // (RLHIP_PIECE_FORCE_MISS=1, a test aid: every piece behind a rank's first is re-evaluated)
    long long *h_xmail = nullptr, *d_xmail = nullptr, xmail_tag = 0;     // pinned mailbox of k_plan_exchange (transfer sizes of the leaf-owner exchange): no stream synchronisation in a round
    double *d_qsend = nullptr, *d_qgath = nullptr, *d_qcat = nullptr; int32_t *d_allQ = nullptr;
    // the same for the validation set (sharded by query like the training set)
    int32_t vQglobal = 0, vQmax = 0; double *d_vqsend = nullptr, *d_vqgath = nullptr, *d_vqcat = nullptr; int32_t *d_vallQ = nullptr;
    double dist_timeout_s = 300.0;             // a rank that waits this long for its own device to report a growth step gives up (RLHIP_DIST_TIMEOUT_S)
};

namespace rl {

static int check_trainer(const rl_trainer *t) { return t ? RL_OK : fail(RL_ERR_INVALID, "null trainer handle"); }

// ---- host-side NDCG constants ------------------------------------------------------------------
static double discount_of(int i) { return 1.0 / (std::log((double)(i + 2)) / std::log(2.0)); }   // DCGScorer.java:26

static double ideal_dcg(const float *labels, int n, int topk, const std::vector<double> &disc)
{   // NDCGScorer.getIdealDCG (:167-174)
    std::vector<int> rel(n);
    for (int i = 0; i < n; i++) rel[i] = (int)labels[i];
    std::sort(rel.begin(), rel.end(), [](int a, int b) { return a > b; });
    double dcg = 0;
    for (int i = 0; i < topk; i++) dcg += (double)(int32_t)(((uint32_t)1 << (rel[i] & 31)) - 1u) * disc[i];      // Java int arithmetic (DCGScorer.java:137-139)
    return dcg;
}

static int validate_dataset(const float *X, int64_t n, int32_t F, const float *labels, const int32_t *qoff, int32_t Q)
{
    if (!labels || !qoff) return fail(RL_ERR_INVALID, "null data pointer");       // X == NULL: the rows follow through rl_set_rows
    if (n <= 0 || Q <= 0 || F <= 0) return fail(RL_ERR_INVALID, "There are no training samples / features");
    if (n >= (int64_t)2147483647 - 4096) return fail(RL_ERR_UNSUPPORTED, "more than 2^31 documents per GPU");
    if (qoff[0] != 0 || (int64_t)qoff[Q] != n) return fail(RL_ERR_INVALID, "qoff must start at 0 and end at n_docs");
    for (int32_t q = 0; q < Q; q++)
        if (qoff[q + 1] <= qoff[q]) return fail(RL_ERR_INVALID, "qoff must be strictly increasing (empty ranked list)");
    for (int64_t i = 0; i < n; i++) {
        if (!(labels[i] >= 0)) return fail(RL_ERR_INVALID, "Relevance label cannot be negative. System will now exit.");  // DataPoint.java:71-73
        // labels above 30 are legal: the gain (1 << l) - 1 wraps as Java ints do (gain_of).  The Java keeps a gain cache of l + 10 doubles
        // (DCGScorer.java:131-140), so a label near 2^31 ends in an OutOfMemoryError there; the line is drawn where (int)label is exact
        if (labels[i] >= 16777216.f) return fail(RL_ERR_UNSUPPORTED, "relevance label of 2^24 or more");
    }
    return RL_OK;
}

static int load_dataset(rl_trainer *t, DataSet &d, const float *X, int64_t n, const float *labels, const int32_t *qoff,
                        int32_t Q, const int32_t *qkey)
{
    d.N = n; d.Q = Q;
    d.labels.assign(labels, labels + n);
    d.qoff.assign(qoff, qoff + Q + 1);
    d.has_key = qkey != nullptr;
    if (qkey) d.qkey.assign(qkey, qkey + Q); else d.qkey.clear();
    d.maxq = 0;
    for (int32_t q = 0; q < Q; q++) d.maxq = std::max(d.maxq, qoff[q + 1] - qoff[q]);
    RL_HIP(t->pool.alloc(&d.d_X, (size_t)n * t->F));
    if (X) { RL_HIP(hipMemcpy(d.d_X, X, (size_t)n * t->F * sizeof(float), hipMemcpyHostToDevice)); d.rows_next = -1; }
    else d.rows_next = 0;
    return RL_OK;
}

static int upload_query_side(rl_trainer *t, DataSet &d, const std::vector<double> &ideal0, const std::vector<double> &ideal1)
{
    RL_HIP(t->pool.alloc(&d.d_labels, (size_t)d.N));
    RL_HIP(hipMemcpy(d.d_labels, d.labels.data(), d.N * sizeof(float), hipMemcpyHostToDevice));
    RL_HIP(t->pool.alloc(&d.d_qoff, (size_t)d.Q + 1));
    RL_HIP(hipMemcpy(d.d_qoff, d.qoff.data(), ((size_t)d.Q + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    RL_HIP(t->pool.alloc(&d.d_ideal0, (size_t)d.Q));
    RL_HIP(t->pool.alloc(&d.d_ideal1, (size_t)d.Q));
    RL_HIP(hipMemcpy(d.d_ideal0, ideal0.data(), d.Q * sizeof(double), hipMemcpyHostToDevice));
    RL_HIP(hipMemcpy(d.d_ideal1, ideal1.data(), d.Q * sizeof(double), hipMemcpyHostToDevice));
    RL_HIP(t->pool.alloc(&d.d_scores, (size_t)d.N));
    RL_HIP(hipMemset(d.d_scores, 0, d.N * sizeof(double)));                // modelScores = 0  LambdaMART.java:86
    RL_HIP(t->pool.alloc(&d.d_ndcg, (size_t)d.Q));
    size_t tiny_min = 4096;           // lists of <= 16 documents get kernels of their own when there are enough of them
    if (const char *e = getenv("RLHIP_TINY_MIN")) tiny_min = (size_t)std::max(0, atoi(e));      // tests / tuning
    std::vector<int32_t> small, big, tiny, huge;
    for (int32_t q = 0; q < d.Q; q++) {
        const int n = d.qoff[q + 1] - d.qoff[q];
        (n <= kRankTinyDocs ? tiny : n <= kLambdaWaveCap ? small : n <= kLambdaBlockCap ? big : huge).push_back(q);
    }
    d.n_huge = (int32_t)huge.size();
    if (!huge.empty()) {
        RL_HIP(t->pool.alloc(&d.d_qhuge, huge.size()));
        RL_HIP(hipMemcpy(d.d_qhuge, huge.data(), huge.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        RL_HIP(t->pool.alloc(&d.d_relscratch, (size_t)d.N));
    }
    if (tiny.size() < tiny_min) { small.insert(small.end(), tiny.begin(), tiny.end()); tiny.clear(); }
    d.n_tiny = (int32_t)tiny.size();
    RL_HIP(t->pool.alloc(&d.d_qtiny, tiny.size()));
    if (!tiny.empty()) RL_HIP(hipMemcpy(d.d_qtiny, tiny.data(), tiny.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    // longest first: a block's work grows with n (n^2 for the rank), so late long lists would leave a tail
    auto by_len = [&](int32_t a, int32_t b) { return (d.qoff[a + 1] - d.qoff[a]) > (d.qoff[b + 1] - d.qoff[b]); };
    std::stable_sort(small.begin(), small.end(), by_len);
    std::stable_sort(big.begin(), big.end(), by_len);
    d.n_small = (int32_t)small.size(); d.n_big = (int32_t)big.size();
    d.n_small_long = 0;
    for (int32_t q : small) d.n_small_long += (d.qoff[q + 1] - d.qoff[q] > kRankShort) ? 1 : 0;
    d.max_big = big.empty() ? 0 : (((d.qoff[big[0] + 1] - d.qoff[big[0]]) + 63) & ~63);      // (longest first) what k_rank_block's LDS is sized for
    d.all_small = false;      // the lists are permutations now: always index through them
    RL_HIP(t->pool.alloc(&d.d_qsmall, small.size()));
    RL_HIP(t->pool.alloc(&d.d_qbig, big.size()));
    if (!small.empty()) RL_HIP(hipMemcpy(d.d_qsmall, small.data(), small.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!big.empty()) RL_HIP(hipMemcpy(d.d_qbig, big.data(), big.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    std::vector<int32_t> qcls[5];
    for (int32_t q = 0; q < d.Q; q++) {
        const int n = d.qoff[q + 1] - d.qoff[q];
        qcls[n <= kLambdaTinyDocs ? 4 : n <= 64 ? 0 : n <= 128 ? 1 : n <= 192 ? 2 : 3].push_back(q);
    }
    if (qcls[4].size() < tiny_min) {   // a handful of tiny lists is not worth a launch of its own: they join the 64-wide class
        qcls[0].insert(qcls[0].end(), qcls[4].begin(), qcls[4].end());
        qcls[4].clear();
    }
    for (int cI = 0; cI < 5; cI++) {
        std::stable_sort(qcls[cI].begin(), qcls[cI].end(), by_len);
        d.n_qcls[cI] = (int32_t)qcls[cI].size();
        RL_HIP(t->pool.alloc(&d.d_qcls[cI], qcls[cI].size()));
        if (!qcls[cI].empty()) RL_HIP(hipMemcpy(d.d_qcls[cI], qcls[cI].data(), qcls[cI].size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return RL_OK;
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// ---- timing helpers ----------------------------------------------------------------------------
static hipEvent_t take_event(rl_trainer *t)
{
    if (!t->ev_free.empty()) { hipEvent_t e = t->ev_free.back(); t->ev_free.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
struct ScopedTiming {
    rl_trainer *t; int which; hipEvent_t a = nullptr, b = nullptr; bool on;
    ScopedTiming(rl_trainer *t_, int which_, double bytes)
        : t(t_), which(which_), on((t_->p.flags & (which_ == RL_KERNEL_HIST_NODE ? RL_FLAG_TIMING_NODES : RL_FLAG_TIMING)) != 0)
    {
        if (!on) return;
        a = take_event(t); b = take_event(t);
        (void)hipEventRecord(a, t->stream);
        t->ev_bytes[which].push_back(bytes);
    }
    ~ScopedTiming() { if (on) { (void)hipEventRecord(b, t->stream); t->ev_pending[which].push_back({a, b}); } }
};
static void collect_timing(rl_trainer *t)
{
    for (int w = 0; w < RL_KERNEL_COUNT_; w++) {
        for (size_t i = 0; i < t->ev_pending[w].size(); i++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, t->ev_pending[w][i].first, t->ev_pending[w][i].second) == hipSuccess) {
                t->timing[w].ms += ms; t->timing[w].launches++; t->timing[w].bytes += t->ev_bytes[w][i];
            }
            t->ev_free.push_back(t->ev_pending[w][i].first); t->ev_free.push_back(t->ev_pending[w][i].second);
        }
        t->ev_pending[w].clear(); t->ev_bytes[w].clear();
    }
}


// ---- exact parallel float chains (rl_chain.inc) ------------------------------------------------------
static size_t chain_stitch_lds(const ChainBufs &b)
{
    const size_t ng1 = (size_t)(b.cap_chunks / b.group + 2), ng2 = ng1 / kChainSuper + 2;
    return (ng1 * (kChainW + 1) + ng2 * kChainW) * sizeof(uint32_t);
}

static int alloc_chain(rl_trainer *t, ChainBufs &b, int maxseg, int A, int64_t n, bool hint = false)
{
    memset(&b, 0, sizeof(b));
    b.maxseg = maxseg; b.A = A;
    b.cap_tiles = n / kChainTile + maxseg + 2;
    b.cap_chunks = b.cap_tiles + maxseg + 2;
    b.cap_n = std::max<int64_t>(n, b.cap_chunks);
    // stitch tables in LDS: first-level groups of `group` chunks (the smaller the group, the shorter the chain of dependent loads)
    b.group = kChainBlock;
    while (chain_stitch_lds(b) > 152 * 1024) {
        b.group *= 2;
        if (b.group > 65536) return fail(RL_ERR_UNSUPPORTED, "data set too large for the float-chain stitch kernel");
    }
    RL_HIP(t->pool.alloc(&b.plan, (size_t)1));
    RL_HIP(t->pool.alloc(&b.seg_start, (size_t)maxseg + 2)); RL_HIP(t->pool.alloc(&b.seg_tile0, (size_t)maxseg + 2));
    RL_HIP(t->pool.alloc(&b.xs, (size_t)A * b.cap_n)); RL_HIP(t->pool.alloc(&b.pre, (size_t)A * b.cap_n));
    RL_HIP(t->pool.alloc(&b.tile_tot, (size_t)A * b.cap_tiles)); RL_HIP(t->pool.alloc(&b.tile_base, (size_t)A * b.cap_tiles));
    RL_HIP(t->pool.alloc(&b.bnd, (size_t)A * b.cap_tiles));
    RL_HIP(t->pool.alloc(&b.cbase, (size_t)A * b.cap_chunks)); RL_HIP(t->pool.alloc(&b.drift, (size_t)A * b.cap_chunks));
    RL_HIP(t->pool.alloc(&b.drift2, (size_t)A * b.cap_chunks));
    RL_HIP(t->pool.alloc(&b.gkey, (size_t)A * b.cap_chunks)); RL_HIP(t->pool.alloc(&b.gkey2, (size_t)A * b.cap_chunks));
    RL_HIP(t->pool.alloc(&b.R, (size_t)A * b.cap_chunks * kChainW));
    RL_HIP(t->pool.alloc(&b.comp0, (size_t)A * (b.cap_chunks / kChainBlock + 1) * kChainW));
    RL_HIP(t->pool.alloc(&b.result, (size_t)A * maxseg)); RL_HIP(t->pool.alloc(&b.miss, (size_t)A * maxseg));
    RL_HIP(t->pool.alloc(&b.st_status, (size_t)A * maxseg)); RL_HIP(t->pool.alloc(&b.st_chunk, (size_t)A * maxseg));
    RL_HIP(t->pool.alloc(&b.st_key, (size_t)A * maxseg)); RL_HIP(t->pool.alloc(&b.st_delta, (size_t)A * maxseg));
    RL_HIP(t->pool.alloc(&b.st_win, (size_t)A * maxseg));
    b.lds_words = (int32_t)(chain_stitch_lds(b) / 4);
    RL_HIP(hipMemset(b.st_status, 0, (size_t)A * maxseg * sizeof(int32_t)));
    RL_HIP(hipMemset(b.miss, 0, (size_t)A * maxseg * sizeof(int32_t)));
    RL_HIP(t->pool.alloc(&b.arrive, (size_t)1)); RL_HIP(hipMemset(b.arrive, 0, sizeof(unsigned long long)));
    if (hint) {      // optional, like the trainer's progress word
        if (hipHostMalloc((void **)&b.h_progress, sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            *b.h_progress = 0;
            t->pinned.push_back(b.h_progress);
            if (hipHostGetDevicePointer((void **)&b.progress, b.h_progress, 0) != hipSuccess) { b.progress = nullptr; b.h_progress = nullptr; (void)hipGetLastError(); }
        } else { b.h_progress = nullptr; (void)hipGetLastError(); }
    }
    RL_HIP(t->pool.alloc(&b.stats, (size_t)4));
    RL_HIP(hipMemset(b.stats, 0, 4 * sizeof(int32_t)));
    RL_HIP(hipMemset(b.plan, 0, sizeof(ChainPlan)));
    return RL_OK;
}

// the plan must already be on the device (k_leaf_table / k_plan_single); grids are sized by capacity
// bounded wait on a pinned progress word written by the device (a scheduling hint, never a correctness dependency)
template <class Pred>
static bool spin_until(const unsigned long long *word, Pred ok, unsigned long long &w)
{
    w = __atomic_load_n(word, __ATOMIC_ACQUIRE);
    if (ok(w)) return true;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; spins++) {
        w = __atomic_load_n(word, __ATOMIC_ACQUIRE);
        if (ok(w)) return true;
        if ((spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) return false;
    }
}

static void enqueue_chain(rl_trainer *t, const ChainBufs &b_in, const ChainSource &src, hipStream_t s = nullptr)
{
    if (!s) s = t->stream;
    // The device cuts its repair passes into windows (kChainWinMax chunks) only when the host watches them and keeps repairing (the progress word);
    // a host that enqueues its passes blindly (no pinned word, RLHIP_STEP_AHEAD=0) gets passes that rebuild everything that remains -- otherwise a
    // chain longer than kChainRepairs windows ended in the serial fallback (exact, but ~100 ms for a 40 M-document leaf).
    ChainBufs b = b_in;
    if (!(b.h_progress != nullptr && t->step_ahead > 0)) { b.progress = nullptr; b.h_progress = nullptr; }
    const unsigned tb = (unsigned)((b.cap_tiles + 3) / 4);
    hipLaunchKernelGGL(k_chain_prefix, dim3(tb), dim3(kThreads), 0, s, b, src);
    hipLaunchKernelGGL(k_chain_scan_tiles, dim3(b.A), dim3(kScanThreads), 0, s, b);
    hipLaunchKernelGGL(k_chain_bounds, dim3(tb, b.A), dim3(kThreads), 0, s, b);
    const dim3 p1grid((unsigned)((b.cap_chunks * 16 + kThreads - 1) / kThreads), b.A);
    hipLaunchKernelGGL(k_chain_pass1<false>, p1grid, dim3(kThreads), 0, s, b);
    hipLaunchKernelGGL(k_chain_guess, dim3(b.A), dim3(kScanThreads), 0, s, b);
    const dim3 tgrid((unsigned)((b.cap_chunks + kThreads / 64 - 1) / (kThreads / 64)), b.A);       // one wavefront per chunk
    const size_t lds = chain_stitch_lds(b);
    // every stitch pass carries a tag; with a progress word (ChainBufs::h_progress) the host looks at the result of the pass
    // before the last one it enqueued and leaves the remaining repair passes (near-empty launches) away once nothing is open
    const unsigned long long seq = ++t->chain_seq;
    t->chain_calls[b.h_progress ? 0 : 1]++;
    bool hint = b.h_progress != nullptr && t->step_ahead > 0, clean = false;
    const dim3 cgrid((unsigned)((b.cap_chunks / kChainBlock + kThreads / 64) / (kThreads / 64)), b.A);       // one wavefront per block of kChainBlock chunks
    hipLaunchKernelGGL(k_chain_tables<false>, tgrid, dim3(kThreads), 0, s, b, 0);
    hipLaunchKernelGGL(k_chain_compose, cgrid, dim3(kThreads), 0, s, b, 0);
    hipLaunchKernelGGL(k_chain_stitch, dim3(b.maxseg, b.A), dim3(kScanThreads), lds, s, b, 0, seq << 16);
    // With the hint the host sees every stitch's outcome before it enqueues the next repair pass, so it repairs for as long as a segment is open
    // (a chain of tens of millions of elements needs a pass per window of 8192 chunks and one per window miss; the serial finish of a chain
    // that long costs 100 ms) and enqueues nothing on a clean round; without it exactly kChainRepairs passes are enqueued blindly.
    for (int rep = 0; rep < (hint ? kChainRepairsMax : kChainRepairs); rep++) {
        if (hint) {      // (a repair pass is four launches: none is enqueued before the stitch it would repair has reported)
            const unsigned long long want = (seq << 16) | (unsigned)rep;            // the stitch before repair pass rep (0 = the first stitch)
            unsigned long long w;
            const auto tw0 = std::chrono::steady_clock::now();
            const bool seen = spin_until(b.h_progress, [&](unsigned long long v) { return (v >> 1) >= want; }, w);
            t->chain_wait_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tw0).count();
            if (!seen) { hint = false; t->chain_timeouts++; }
            else if ((w >> 17) == seq && !(w & 1)) { clean = true; break; }
        }
        t->chain_repairs[b.h_progress ? 0 : 1]++;
        hipLaunchKernelGGL(k_chain_pass1<true>, p1grid, dim3(kThreads), 0, s, b);
        hipLaunchKernelGGL(k_chain_recentre, dim3(b.maxseg, b.A), dim3(kScanThreads), 0, s, b);
        hipLaunchKernelGGL(k_chain_tables<true>, tgrid, dim3(kThreads), 0, s, b, 1);
        hipLaunchKernelGGL(k_chain_compose, cgrid, dim3(kThreads), 0, s, b, 1);
        hipLaunchKernelGGL(k_chain_stitch, dim3(b.maxseg, b.A), dim3(kScanThreads), lds, s, b, 1, (seq << 16) | (unsigned)(rep + 1));
    }
    if (!clean) hipLaunchKernelGGL(k_chain_fallback, dim3(b.maxseg, b.A), dim3(64), 0, s, b);
}

// float s = 0; for (q) s += ndcg_q; s / Q   -- serial for short lists, exact parallel chain otherwise
// the ordinary repair passes on the segments k_chain_arm has opened (exact start states known), until the host has seen them closed
static void chain_repair_rounds(rl_trainer *t, const ChainBufs &b, hipStream_t s)
{
    const dim3 p1grid((unsigned)((b.cap_chunks * 16 + kThreads - 1) / kThreads), b.A);
    const dim3 tgrid((unsigned)((b.cap_chunks + kThreads / 64 - 1) / (kThreads / 64)), b.A);
    const dim3 cgrid((unsigned)((b.cap_chunks / kChainBlock + kThreads / 64) / (kThreads / 64)), b.A);
    const size_t lds = chain_stitch_lds(b);
    const unsigned long long seq = ++t->chain_seq;
    bool hint = b.h_progress != nullptr && t->step_ahead > 0, clean = false;
    for (int rep = 0; rep < (hint ? kChainRepairsMax : kChainRepairs); rep++) {
        t->chain_repairs[b.h_progress ? 0 : 1]++;
        hipLaunchKernelGGL(k_chain_pass1<true>, p1grid, dim3(kThreads), 0, s, b);
        hipLaunchKernelGGL(k_chain_recentre, dim3(b.maxseg, b.A), dim3(kScanThreads), 0, s, b);
        hipLaunchKernelGGL(k_chain_tables<true>, tgrid, dim3(kThreads), 0, s, b, 1);
        hipLaunchKernelGGL(k_chain_compose, cgrid, dim3(kThreads), 0, s, b, 1);
        hipLaunchKernelGGL(k_chain_stitch, dim3(b.maxseg, b.A), dim3(kScanThreads), lds, s, b, 1, (seq << 16) | (unsigned)(rep + 1));
        if (hint) {
            const unsigned long long want = (seq << 16) | (unsigned)(rep + 1);
            unsigned long long w;
            const bool seen = spin_until(b.h_progress, [&](unsigned long long v) { return (v >> 1) >= want; }, w);
            if (!seen) { hint = false; t->chain_timeouts++; }
            else if ((w >> 17) == seq && !(w & 1)) { clean = true; break; }
        }
    }
    if (!clean) hipLaunchKernelGGL(k_chain_fallback, dim3(b.maxseg, b.A), dim3(64), 0, s, b);
}

// Sharded runs, round 6: the leaves' float sums from this rank's own pieces (rl_dist.inc, "piece mode"): three small all-gathers, no document leaves its rank.
static int enqueue_leaf_chains_pieces(rl_trainer *t, const ChainSource &src)
{
    Ctx &c = t->ctx;
    hipStream_t s = t->stream;
    ChainBufs b = t->leaf_chain;
    if (!(b.h_progress != nullptr && t->step_ahead > 0)) { b.progress = nullptr; b.h_progress = nullptr; }
    const int R = t->n_ranks, me = t->dist->rank, A = b.A, MS = b.maxseg, nseg = std::max(c.L, 2), n = A * MS;
    int rcd = t->dist->allgather(c.leaf_start, t->d_gls, (size_t)t->lsstride * sizeof(int32_t), s);       // the pieces' lengths on every rank
    if (rcd) return rcd;
    const unsigned tb = (unsigned)((b.cap_tiles + 3) / 4), nb = (unsigned)((n + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(k_chain_prefix, dim3(tb), dim3(kThreads), 0, s, b, src);
    hipLaunchKernelGGL(k_chain_scan_tiles, dim3(b.A), dim3(kScanThreads), 0, s, b);
    hipLaunchKernelGGL(k_piece_totals, dim3(nb), dim3(kThreads), 0, s, b, t->d_pc_loc);
    rcd = t->dist->allgather(t->d_pc_loc, t->d_pc_all, (size_t)n * sizeof(double), s);
    if (rcd) return rcd;
    hipLaunchKernelGGL(k_piece_base, dim3(nb), dim3(kThreads), 0, s, (const double *)t->d_pc_all, n, me, t->d_pc_base, 0);
    b.seg_base = t->d_pc_base; b.ptab = t->d_ptab;
    hipLaunchKernelGGL(k_chain_bounds, dim3(tb, b.A), dim3(kThreads), 0, s, b);
    const dim3 p1grid((unsigned)((b.cap_chunks * 16 + kThreads - 1) / kThreads), b.A);
    hipLaunchKernelGGL(k_chain_pass1<false>, p1grid, dim3(kThreads), 0, s, b);
    hipLaunchKernelGGL(k_piece_drifts, dim3(MS, b.A), dim3(64), 0, s, b, t->d_pc_loc);
    rcd = t->dist->allgather(t->d_pc_loc, t->d_pc_all, (size_t)n * sizeof(double), s);
    if (rcd) return rcd;
    hipLaunchKernelGGL(k_piece_base, dim3(nb), dim3(kThreads), 0, s, (const double *)t->d_pc_all, n, me, t->d_pc_base, 1);
    hipLaunchKernelGGL(k_chain_guess, dim3(b.A), dim3(kScanThreads), 0, s, b);
    const dim3 tgrid((unsigned)((b.cap_chunks + kThreads / 64 - 1) / (kThreads / 64)), b.A);
    const dim3 cgrid((unsigned)((b.cap_chunks / kChainBlock + kThreads / 64) / (kThreads / 64)), b.A);
    const size_t lds = chain_stitch_lds(b);
    const unsigned long long seq = ++t->chain_seq;
    t->chain_calls[b.h_progress ? 0 : 1]++;
    hipLaunchKernelGGL(k_chain_tables<false>, tgrid, dim3(kThreads), 0, s, b, 0);
    hipLaunchKernelGGL(k_chain_compose, cgrid, dim3(kThreads), 0, s, b, 0);
    hipLaunchKernelGGL(k_chain_stitch, dim3(b.maxseg, b.A), dim3(kScanThreads), lds, s, b, 0, seq << 16);
    rcd = t->dist->allgather(t->d_ptab, t->d_gtab, (size_t)n * (kChainW + 1) * sizeof(uint32_t), s);
    if (rcd) return rcd;
    hipLaunchKernelGGL(k_chain_cross, dim3(1), dim3(kThreads), 0, s, (const uint32_t *)t->d_gtab, (const int32_t *)t->d_gls, t->lsstride, R, nseg, A, MS, me, t->xstate,
                       (const uint32_t *)t->d_res_all, 0, t->piece_force, b.result, t->d_xmail, ++t->xmail_tag);
    ChainBufs br = b; br.ptab = nullptr;            // (the repair passes stitch from ONE known state)
    for (int round = 0;; round++) {
        // every rank sees the same gathered tables, hence the same pending pieces: the rounds of this loop -- and their collectives -- are the same everywhere
        const auto t0w = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(&t->h_xmail[2], __ATOMIC_ACQUIRE) != t->xmail_tag) {
            if ((++spins & 0xffff) == 0) {
                const hipError_t q = hipStreamQuery(s);
                if (q != hipSuccess && q != hipErrorNotReady) return fail(RL_ERR_HIP, std::string("device error in the leaves' float chains: ") + hipGetErrorString(q));
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0w).count() > t->dist_timeout_s)
                    return fail(RL_ERR_COMM, "timed out waiting for the leaves' float chains (a rank of the job is missing from a collective?)");
            }
        }
        const long long pend = t->h_xmail[0], mine = t->h_xmail[1];
        if (pend == 0) break;
        if (round > R * n + 8) return fail(RL_ERR_STATE, "the leaves' float chains did not close (internal error)");
        t->piece_rounds++; t->piece_misses += pend;
        hipLaunchKernelGGL(k_chain_arm, dim3(1), dim3(kThreads), 0, s, br, t->xstate, me, br.progress ? 1 : 0);
        if (mine > 0) chain_repair_rounds(t, br, s);
        hipLaunchKernelGGL(k_chain_resolved, dim3(1), dim3(kThreads), 0, s, br, t->xstate, me, t->d_res_loc);
        rcd = t->dist->allgather(t->d_res_loc, t->d_res_all, (size_t)n * sizeof(uint32_t), s);
        if (rcd) return rcd;
        hipLaunchKernelGGL(k_chain_cross, dim3(1), dim3(kThreads), 0, s, (const uint32_t *)t->d_gtab, (const int32_t *)t->d_gls, t->lsstride, R, nseg, A, MS, me, t->xstate,
                           (const uint32_t *)t->d_res_all, 1, t->piece_force, b.result, t->d_xmail, ++t->xmail_tag);
    }
    return RL_OK;
}

static void enqueue_metric_mean(rl_trainer *t, const double *ndcg_q, int Q, float *out, hipStream_t s = nullptr)
{
    if (!s) s = t->stream;
    if (Q <= 4096 || (t->p.flags & RL_FLAG_SERIAL_CHAIN)) { hipLaunchKernelGGL(k_float_mean, dim3(1), dim3(64), 0, s, ndcg_q, Q, out); return; }
    hipLaunchKernelGGL(k_plan_single, dim3(1), dim3(64), 0, s, t->metric_chain, Q);
    ChainSource src{ndcg_q, nullptr, nullptr, nullptr, nullptr, nullptr};
    enqueue_chain(t, t->metric_chain, src, s);
    hipLaunchKernelGGL(k_metric_finish, dim3(1), dim3(64), 0, s, t->metric_chain, Q, out);
}

// ---- per-query kernels on a data set -------------------------------------------------------------
// rank every query by `scores` (stable, descending) and leave NDCG@k per query in `out`; with `ranked` the
// ranked-order arrays of the data set are refreshed too (they feed the next round's lambdas)
static int launch_rank(rl_trainer *t, DataSet &d, const double *scores, double *out, bool ranked)
{
    RankArgs a{scores, d.d_labels, d.d_qoff, d.d_ideal1, t->ctx.disc,
               ranked ? d.d_ss : nullptr, ranked ? d.d_sl : nullptr, ranked ? d.d_srel : nullptr, ranked ? d.d_sidx : nullptr,
               out, t->p.metric_k, t->p.metric, ranked ? d.d_aux_i : nullptr, ranked ? d.d_aux_a : nullptr, ranked ? d.d_aux_b : nullptr, t->err_max,
               d.d_ext_rd};
    if (d.n_tiny > 0)
        hipLaunchKernelGGL(k_rank_tiny, dim3((d.n_tiny + kRankTinyGroups - 1) / kRankTinyGroups), dim3(kRankTinyDocs * kRankTinyGroups), 0, t->stream, a,
                           (const int *)d.d_qtiny, d.n_tiny);
    static const bool rank_mixed = getenv("RLHIP_RANK_SPLIT") == nullptr;
    if (rank_mixed && d.n_big > 0 && d.n_small > 0) {
        const int wpb = kRankBlockThreads / 64;
        const size_t lds = std::max((size_t)d.max_big, (size_t)wpb * kLambdaWaveCap) * kRankLdsPerDoc;
        hipLaunchKernelGGL(k_rank_mixed, dim3(d.n_big + (d.n_small + wpb - 1) / wpb), dim3(kRankBlockThreads), lds, t->stream, a, (const int *)d.d_qbig, d.n_big, d.max_big,
                           (const int *)d.d_qsmall, d.n_small, kLambdaWaveCap);
        if (d.n_huge > 0)
            hipLaunchKernelGGL(k_rank_huge, dim3(d.n_huge), dim3(kRankBlockThreads), 0, t->stream, a, (const int *)d.d_qhuge, d.n_huge, d.d_relscratch);
        RL_HIP(hipGetLastError());
        return RL_OK;
    }
    if (d.n_small_long > 0)       // (d_qsmall: longest first)
        hipLaunchKernelGGL(k_rank_wave, dim3((d.n_small_long + 3) / 4), dim3(kThreads), 4 * kLambdaWaveCap * kRankLdsPerDoc, t->stream, a,
                           (const int *)d.d_qsmall, d.n_small_long, kLambdaWaveCap);
    if (d.n_small > d.n_small_long)
        hipLaunchKernelGGL(k_rank_wave, dim3((d.n_small - d.n_small_long + 3) / 4), dim3(kThreads), 4 * kRankShort * kRankLdsPerDoc, t->stream, a,
                           (const int *)d.d_qsmall + d.n_small_long, d.n_small - d.n_small_long, kRankShort);
    if (d.n_big > 0)
        hipLaunchKernelGGL(k_rank_block, dim3(d.n_big), dim3(kRankBlockThreads), (size_t)d.max_big * kRankLdsPerDoc, t->stream, a, d.d_qbig, d.n_big, d.max_big);
    if (d.n_huge > 0)
        hipLaunchKernelGGL(k_rank_huge, dim3(d.n_huge), dim3(kRankBlockThreads), 0, t->stream, a, (const int *)d.d_qhuge, d.n_huge, d.d_relscratch);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

template <bool ROOT>
static void launch_hist(const Ctx &c, int gx, int gy, size_t lds, hipStream_t s, bool fq = false)
{
    // the XCD-aware block map of k_hist wants a multiple of 8 chunks (the extra blocks exit); child passes: a bounded grid whose blocks walk the
    // step's chunks (k_hist), about one resident set of blocks (3 per CU)
    static const int grid_blocks = getenv("RLHIP_HIST_GRID") ? atoi(getenv("RLHIP_HIST_GRID")) : 1024;
    // (balanced steps -- balance_slots -- want exactly balance_target rows: block row r then works through the chunks r, r + balance_target, ..)
    static const bool grid_env = getenv("RLHIP_HIST_GRID") != nullptr;        // (cached: this runs thousands of times a second)
    const auto bounded = [&](int gxx) { return ROOT ? ((gy + 7) & ~7) : (c.balance && (!c.sharded || c.cum_cnt_loc) && !grid_env) ? std::min((gy + 7) & ~7, c.balance_target)
                                                                      : std::min((gy + 7) & ~7, std::max(8, ((grid_blocks + gxx - 1) / gxx + 7) & ~7)); };
    // (RLHIP_HIST_LDSPAD: extra dynamic LDS per child-pass block -- 28 KB caps a CU at two blocks; a measuring aid)
    static const size_t lds_pad = getenv("RLHIP_HIST_LDSPAD") ? (size_t)atoi(getenv("RLHIP_HIST_LDSPAD")) : 0;
    if (!ROOT) lds += lds_pad;
    const dim3 g(gx, bounded(gx)), b(kThreads);
    if (!ROOT && c.crows && c.sub == 16 && c.TS <= kHistLdsStride && !c.any_runs) {      // sparse data: compact rows (k_compact_rows)
        hipLaunchKernelGGL((k_hist<false, 16, kHistLdsStride, false, false, kThreads, true>), g, b, lds, s, c);
        return;
    }
    if (!ROOT && c.sub == 16 && c.TS <= kHistLdsStride && !c.any_runs && !(c.p8 > 1) && c.sub_child < 16) {
        // child passes, features of a group spread over 16 / sub_child blocks: a step of a few chunks leaves most CUs idle while every block is bound by
        // the LDS atomics of ITS CU -- the same atomics on more CUs (the rows are read once per sub-block, from L2)
        const dim3 g2(gx * (16 / c.sub_child), bounded(gx * (16 / c.sub_child)));
        const size_t lds2 = (size_t)c.sub_child * kHistLdsStride * 12;
        if (c.sub_child == 8) hipLaunchKernelGGL((k_hist<false, 8, kHistLdsStride, false, false, 256>), g2, dim3(256), lds2, s, c);
        else hipLaunchKernelGGL((k_hist<false, 4, kHistLdsStride, false, false, 256>), g2, dim3(256), lds2, s, c);
        return;
    }
    if (!ROOT && c.sub == 16 && c.TS <= kHistLdsStride && !c.any_runs && !(c.p8 > 1) && c.hist_nt > kThreads) {
        // child passes: a step has few chunks (~12 per node), so the chip is mostly idle and every block is a chain of dependent row gathers --
        // larger blocks keep more of them in flight per chunk
        if (c.hist_nt >= 1024) hipLaunchKernelGGL((k_hist<false, 16, kHistLdsStride, false, false, 1024>), g, dim3(1024), lds, s, c);
        else hipLaunchKernelGGL((k_hist<false, 16, kHistLdsStride, false, false, 512>), g, dim3(512), lds, s, c);
        return;
    }
    if constexpr (ROOT) {
        if (fq && c.sub == 16 && c.TS <= kHistLdsStride && !c.any_runs) {      // root pass that quantises the lambdas itself (root_quant_fused)
            if (c.p8) hipLaunchKernelGGL((k_hist<true, 16, kHistLdsStride, false, true, kThreads, false, true>), g, b, lds, s, c);
            else hipLaunchKernelGGL((k_hist<true, 16, kHistLdsStride, false, false, kThreads, false, true>), g, b, lds, s, c);
            return;
        }
    }
    if (c.sub == 16 && c.TS <= kHistLdsStride) {
        // packed rows for the root pass only (measured at c2: 43 % fewer bytes buy the root pass 9 % -- it is bound by LDS atomics, not by HBM --
        // and the child passes nothing: their extra bit-field work costs what the two 128-byte lines per document instead of three save)
        if (c.p8 && (ROOT || c.p8 > 1)) {
            if (c.any_runs) hipLaunchKernelGGL((k_hist<ROOT, 16, kHistLdsStride, true, true>), g, b, lds, s, c);
            else hipLaunchKernelGGL((k_hist<ROOT, 16, kHistLdsStride, false, true>), g, b, lds, s, c);
        } else if (c.any_runs) hipLaunchKernelGGL((k_hist<ROOT, 16, kHistLdsStride, true>), g, b, lds, s, c);
        else hipLaunchKernelGGL((k_hist<ROOT, 16, kHistLdsStride>), g, b, lds, s, c);
        return;
    }
    switch (c.sub) {
    case 16: hipLaunchKernelGGL((k_hist<ROOT, 16, 0>), g, b, lds, s, c); break;
    case 8: hipLaunchKernelGGL((k_hist<ROOT, 8, 0>), g, b, lds, s, c); break;
    case 4: hipLaunchKernelGGL((k_hist<ROOT, 4, 0>), g, b, lds, s, c); break;
    case 2: hipLaunchKernelGGL((k_hist<ROOT, 2, 0>), g, b, lds, s, c); break;
    default: hipLaunchKernelGGL((k_hist<ROOT, 1, 0>), g, b, lds, s, c); break;
    }
}

// multi-GPU: per-query values of all ranks in global query order (ranks hold ascending contiguous query ranges)
static int gather_queries(rl_trainer *t, const double *local, const double **out, bool valid = false)
{
    hipStream_t s = t->stream;
    const int Q = valid ? t->va.Q : t->tr.Q, Qmax = valid ? t->vQmax : t->Qmax;
    double *snd = valid ? t->d_vqsend : t->d_qsend, *gat = valid ? t->d_vqgath : t->d_qgath, *cat = valid ? t->d_vqcat : t->d_qcat;
    hipLaunchKernelGGL(k_copy_f64, dim3(std::max(1, std::min(1024, (Q + kThreads - 1) / kThreads))), dim3(kThreads), 0, s, local, snd, Q);
    int rc = t->dist->allgather(snd, gat, (size_t)Qmax * sizeof(double), s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_concat_ranks, dim3(64), dim3(kThreads), 0, s, (const double *)gat, (const int32_t *)(valid ? t->d_vallQ : t->d_allQ), t->n_ranks, Qmax, cat);
    *out = cat;
    return RL_OK;
}

// Lazy Java-order tie-break (rl_tie.inc): the device stalled the tree on nodes whose exactly tied best split the Java's rounding noise decides.
// The stream is idle when this runs (the caller synchronised it).  Reads the node records, lays out the derivation chains -- a node the Java
// accumulates (root / left child) is summed from its members; a right child is parent - left sibling, recursively -- and runs the kernels that
// put the Java's choice into the node records and resume the growth bookkeeping.  Device scratch comes from one arena that only ever grows.
struct TieArena {
    char *base = nullptr; size_t cap = 0, used = 0;
    template <class T> T *take(size_t n) { used = (used + 255) & ~(size_t)255; T *p = (T *)(base + used); used += n * sizeof(T); return p; }
};
static int tie_arena_reserve(rl_trainer *t, size_t bytes)
{
    if (bytes <= t->tie_cap) return RL_OK;
    if (t->tie_buf) { (void)hipFree(t->tie_buf); t->tie_buf = nullptr; t->tie_cap = 0; }
    const size_t want = bytes + bytes / 4 + (1 << 20);
    if (hipMalloc(&t->tie_buf, want) != hipSuccess) { (void)hipGetLastError(); t->tie_buf = nullptr; return RL_ERR_HIP; }
    t->tie_cap = want;
    return RL_OK;
}

// deferred = false: the tree is stalled on TreeState::stall_node (ties whose candidates may cut the node differently); afterwards the growth resumes.
// deferred = true: the tree is grown; the committed nodes flagged 0x40 (plateau ties of right children: the partition was known, the stored
// threshold was not) get the Java's threshold, all of them in one batch, before the tree is exported.
static int resolve_ties(rl_trainer *t, size_t fin_lds, int nodes_in_lds, bool deferred = false, bool *other_cut = nullptr)
{
    Ctx &c = t->ctx;
    hipStream_t s = t->stream;
    const auto t_begin = std::chrono::steady_clock::now();
    struct TieScope { DistBackend *d; TieScope(DistBackend *d_) : d(d_) { if (d) d->tie_scope = true; } ~TieScope() { if (d) d->tie_scope = false; } } tie_scope(t->dist.get());
    // small reads come back through one pinned buffer (a pageable copy costs tens of microseconds each)
    // The pinned buffer grows with what a resolution needs (nothing in flight reads it when it is asked to: every use is copy, synchronise, memcpy):
    // a batch of deferred nodes of a tree with hundreds of leaves, or of wide data with many tied features, is not a reason to stop training
    auto ensure_pin = [&](size_t bytes) -> int {
        if (t->tie_pin_cap >= bytes) return RL_OK;
        if (t->tie_pin) (void)hipHostFree(t->tie_pin);
        t->tie_pin = nullptr; t->tie_pin_cap = 0;
        const size_t want = bytes + bytes / 4;
        if (hipHostMalloc(&t->tie_pin, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(RL_ERR_HIP, "tie-break: no pinned host memory"); }
        t->tie_pin_cap = want;
        return RL_OK;
    };
    { int rcp = ensure_pin(sizeof(TreeState) + (size_t)(c.NC + 2) * sizeof(NodeRec) + (size_t)kTieMaxChain * c.F * 4 + ((size_t)1 << 20)); if (rcp) return rcp; }
    char *pin = (char *)t->tie_pin;
    RL_HIP(hipMemcpyAsync(pin, c.st, sizeof(TreeState), hipMemcpyDeviceToHost, s));
    RL_HIP(hipMemcpyAsync(pin + sizeof(TreeState), c.nodes, (size_t)c.NC * sizeof(NodeRec), hipMemcpyDeviceToHost, s));
    RL_HIP(hipStreamSynchronize(s));
    static const bool tie_prof = getenv("RLHIP_TIE_PROF") != nullptr;
    auto t_last = t_begin;
    auto mark = [&](int ph) { if (!tie_prof) return; const auto now = std::chrono::steady_clock::now(); t->tie_phase_us[ph] += (long long)std::chrono::duration_cast<std::chrono::microseconds>(now - t_last).count(); t_last = now; };
    mark(0);
    TreeState st;
    memcpy(&st, pin, sizeof(st));
    if (!deferred && (st.stall_n <= 0 || st.stall_n > kSpec)) return fail(RL_ERR_STATE, "resolve_ties without a stalled tree (internal error)");
    std::vector<NodeRec> nodes((size_t)st.n_nodes);
    memcpy(nodes.data(), pin + sizeof(TreeState), nodes.size() * sizeof(NodeRec));
    std::vector<int> todo;
    if (deferred) { for (int x = 0; x < st.n_nodes; x++) if (nodes[x].left >= 0 && (nodes[x].tie & 0xc0) == 0x40) todo.push_back(x); }
    else for (int x = 0; x < st.stall_n; x++) todo.push_back(st.stall_node[x]);
    if (todo.empty()) return RL_OK;
    std::vector<TieNode> an; std::vector<TiePred> preds; std::map<int, int> a_of;
    auto is_right = [&](int x) { return nodes[x].parent >= 0 && nodes[nodes[x].parent].pr == x; };
    auto direct = [&](int x) -> int {        // chain node of a directly accumulated node, with the split predicates of its path from the root
        auto it = a_of.find(x);
        if (it != a_of.end()) return it->second;
        TieNode A; A.node = x; A.pred0 = (int)preds.size(); A.npred = 0; A.is_root = (x == 0) ? 1 : 0; A.list0 = 0; A.count = nodes[x].gcount;
        A.gcount = nodes[x].gcount; A.pad = 0;             // count: this rank's members (the device sets it; == gcount on one GPU)
        for (int ch = x; nodes[ch].parent >= 0; ch = nodes[ch].parent) {
            const NodeRec &P = nodes[nodes[ch].parent];
            preds.push_back(TiePred{P.best_f, P.best_t, P.pl == ch ? 1 : 0});
            A.npred++;
        }
        an.push_back(A);
        a_of[x] = (int)an.size() - 1;
        return (int)an.size() - 1;
    };
    const int nx = (int)todo.size();
    // deferred ties over several features: the node was cut by its first candidate; that every tied candidate cuts it the same way is verified
    // document by document (k_tie_verify) -- these are the nodes, with the split predicates of their paths
    std::vector<TieNode> vn; std::vector<int32_t> vx;
    if (deferred)
        for (int x = 0; x < nx; x++) {
            if ((nodes[todo[x]].tie & 3) != 2) continue;
            TieNode V; memset(&V, 0, sizeof(V));
            V.node = todo[x]; V.pred0 = (int)preds.size(); V.is_root = (todo[x] == 0) ? 1 : 0; V.gcount = nodes[todo[x]].gcount;
            for (int ch = todo[x]; nodes[ch].parent >= 0; ch = nodes[ch].parent) {
                const NodeRec &P = nodes[nodes[ch].parent];
                preds.push_back(TiePred{P.best_f, P.best_t, P.pl == ch ? 1 : 0});
                V.npred++;
            }
            vn.push_back(V); vx.push_back(x);
        }
    const int nv = (int)vn.size();
    int32_t *d_vflag = nullptr;
    std::vector<std::vector<int>> chains((size_t)nx);
    size_t chain_cap = 1;
    for (int x = 0; x < nx; x++) {
        // J(X): X itself when the Java accumulates it; else J(parent) - J(left sibling), the parent first (top-down)
        std::vector<int> subs;               // left siblings, bottom-up
        int cur = todo[x];
        while (is_right(cur)) { subs.push_back(nodes[nodes[cur].parent].pl); cur = nodes[cur].parent; }
        chains[x].push_back(direct(cur));
        for (auto it = subs.rbegin(); it != subs.rend(); ++it) chains[x].push_back(direct(*it));
        chain_cap = std::max(chain_cap, chains[x].size());
    }
    const int nA = (int)an.size();
    const bool sharded = t->dist && t->n_ranks > 1;       // the members of a chain node are spread over the ranks: their values are gathered (below)
    const int R = sharded ? t->n_ranks : 1;
    // (the host tables above are complete: the pinned buffer may move now.  need[] and the chain nodes come back through it in stage 1)
    { int rcp = ensure_pin((size_t)nA * c.F * sizeof(int32_t) + (size_t)nA * sizeof(TieNode) + (size_t)std::max(R, 1) * nA * sizeof(int32_t) + ((size_t)1 << 20)); if (rcp) return rcp; pin = (char *)t->tie_pin; }
    size_t list_total = 0, u_total = 0;
    std::vector<long long> u0((size_t)nA);
    int maxcnt = 1;
    for (int i = 0; i < nA; i++) {
        TieNode &A = an[i];
        if (!A.is_root) { A.list0 = (int32_t)list_total; list_total += (size_t)std::min(A.gcount, c.N); }
        u0[i] = (long long)u_total; u_total += (size_t)A.gcount;
        maxcnt = std::max(maxcnt, std::min(A.gcount, c.N));
    }
    if (list_total > ((size_t)1 << 31) - 1) return fail(RL_ERR_UNSUPPORTED, "tie-break: member lists beyond 2^31 entries");
    std::vector<int32_t> xlen((size_t)nx), xchain((size_t)nx * chain_cap, 0), xnode((size_t)nx);
    for (int x = 0; x < nx; x++) {
        xnode[x] = todo[x]; xlen[x] = (int32_t)chains[x].size();
        for (size_t i = 0; i < chains[x].size(); i++) xchain[(size_t)x * chain_cap + i] = chains[x][i];
    }
    const int tiles = (c.N + kTieTile - 1) / kTieTile, nbg = (c.TS + 63) / 64;
    // short chains: the literal walk (one kernel, ~6 ns a document) beats the dozen launches and two more host round trips of the contiguous-chain path;
    // known before anything ran on the device, so stage 1 does not have to report back either
    static const size_t walk_max = getenv("RLHIP_TIE_WALK_MAX") ? (size_t)atoll(getenv("RLHIP_TIE_WALK_MAX")) : (size_t)24576;
    const bool walk_early = !sharded && (getenv("RLHIP_TIE_WALK") != nullptr || u_total <= walk_max || (size_t)kTsWaves * c.TS * 4 > (size_t)60 * 1024);
    // ---- stage 1: fixed-size scratch, the tied candidates, the member lists
    const size_t fixed_bytes = (size_t)nx * c.F * c.TS + ((size_t)nA * c.F + (size_t)nx * c.F + (size_t)nA * tiles + list_total + 64) * 4 + ((size_t)nA * c.F * c.TS + nA) * 8 +
                               (xchain.size() + 2 * (size_t)nx + 16) * 4 + (size_t)nA * sizeof(TieNode) + (preds.size() + 1) * sizeof(TiePred) + (size_t)nA * 8 + 64 * 256 +
                               (nv > 0 ? (size_t)nx * c.F * 12 + (size_t)nx * 4 + (size_t)nv * (sizeof(TieNode) + 4) + 1024 : 0) + (size_t)nx * c.F * 12 + 1024;
    if (tie_arena_reserve(t, std::max(fixed_bytes + ((size_t)64 << 20), t->tie_hint))) return fail(RL_ERR_HIP, "tie-break: out of device memory");
    TieArena ar;
    TieArgs a;
    long long *d_u0 = nullptr;
    std::vector<int32_t> need((size_t)nA * c.F), lcnt((size_t)nA, 0);
    // (a lambda: when stage 2 turns out to need a larger arena, the arena moves and stage 1 is simply run again)
    auto stage1 = [&]() -> int {
        ar = TieArena(); ar.base = (char *)t->tie_buf; ar.cap = t->tie_cap;
        memset(&a, 0, sizeof(a));
        a.nx = nx; a.nA = nA; a.chain_cap = (int32_t)chain_cap;
        // the small host tables travel as ONE copy
        std::vector<char> &blob = t->tie_blob; blob.clear();
        auto put = [&](const void *src, size_t bytes) { const size_t o = (blob.size() + 15) & ~(size_t)15; blob.resize(o + bytes); if (bytes) memcpy(blob.data() + o, src, bytes); return o; };
        const size_t o_xnode = put(xnode.data(), nx * sizeof(int32_t)), o_xlen = put(xlen.data(), nx * sizeof(int32_t)), o_xchain = put(xchain.data(), xchain.size() * sizeof(int32_t));
        const size_t o_an = put(an.data(), nA * sizeof(TieNode)), o_preds = put(preds.data(), preds.size() * sizeof(TiePred)), o_u0 = put(u0.data(), nA * sizeof(long long));
        const size_t o_vn = put(vn.data(), nv * sizeof(TieNode)), o_vx = put(vx.data(), nv * sizeof(int32_t));
        char *d_blob = ar.take<char>(blob.size() + 16);
        RL_HIP(hipMemcpyAsync(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice, s));
        int32_t *d_xnode = (int32_t *)(d_blob + o_xnode), *d_xlen = (int32_t *)(d_blob + o_xlen), *d_xchain = (int32_t *)(d_blob + o_xchain);
        a.an = (TieNode *)(d_blob + o_an); TiePred *d_preds = (TiePred *)(d_blob + o_preds);
        d_u0 = (long long *)(d_blob + o_u0);
        a.tmask = ar.take<uint8_t>((size_t)nx * c.F * c.TS); a.need = ar.take<int32_t>((size_t)nA * c.F); a.xf = ar.take<int32_t>((size_t)nx * c.F);
        a.tile_cnt = ar.take<int32_t>((size_t)nA * tiles); a.list = ar.take<int32_t>(list_total + 1);
        a.jbin = ar.take<double>((size_t)nA * c.F * c.TS); a.jtot = ar.take<double>(nA);
        a.fS = ar.take<double>((size_t)nx * c.F); a.ft = ar.take<int32_t>((size_t)nx * c.F);
        RL_HIP(hipMemsetAsync(a.need, 0, (size_t)nA * c.F * sizeof(int32_t), s));
        a.xnode = d_xnode; a.xlen = d_xlen; a.xchain = d_xchain; a.preds = d_preds;
        if (nv > 0) {
            a.vcnt = ar.take<int32_t>((size_t)nx + 1); a.vlist = ar.take<int32_t>((size_t)nx * c.F * 3);
            d_vflag = a.vcnt + nx;
            RL_HIP(hipMemsetAsync(a.vcnt, 0, ((size_t)nx + 1) * sizeof(int32_t), s));
        }
        hipLaunchKernelGGL(k_tie_cand, dim3(c.F, nx), dim3(kFinThreads), 0, s, c, a);
        if (nv > 0) hipLaunchKernelGGL(k_tie_verify, dim3(tiles, nv), dim3(kThreads), 0, s, c, a, (const TieNode *)(d_blob + o_vn), (const int32_t *)(d_blob + o_vx), d_vflag);
        bool any_list = false;
        for (auto &A : an) any_list |= !A.is_root;
        if (any_list) {
            hipLaunchKernelGGL(k_tie_count, dim3(tiles, nA), dim3(kThreads), 0, s, c, a, tiles);
            hipLaunchKernelGGL(k_tie_scan, dim3(nA), dim3(kThreads), 0, s, a, tiles);
            hipLaunchKernelGGL(k_tie_scatter, dim3(tiles, nA), dim3(kThreads), 0, s, c, a, tiles);
        }
        if (walk_early) return RL_OK;       // (need / lcnt stay zero: no pairs, the walk below)
        RL_HIP(hipMemcpyAsync(pin, a.need, need.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        RL_HIP(hipMemcpyAsync(pin + need.size() * sizeof(int32_t), a.an, (size_t)nA * sizeof(TieNode), hipMemcpyDeviceToHost, s));
        RL_HIP(hipStreamSynchronize(s));
        memcpy(need.data(), pin, need.size() * sizeof(int32_t));
        for (int i = 0; i < nA; i++) {      // this rank's member counts (k_tie_scan)
            TieNode A; memcpy(&A, pin + need.size() * sizeof(int32_t) + (size_t)i * sizeof(TieNode), sizeof(A));
            lcnt[i] = an[i].is_root ? c.N : (any_list ? A.count : an[i].gcount);
        }
        return RL_OK;
    };
    mark(1);
    { int rc1 = stage1(); if (rc1) return rc1; }
    mark(2);
    // ---- stage 2: the needed (chain node, feature) pairs, their bins' sizes, the chains and their segments
    std::vector<TiePair> pairs;
    size_t v_total = u_total, m_total = 0;
    int tiles_max = 1;
    for (int i = 0; i < nA; i++)
        for (int f = 0; f < c.F; f++)
            if (need[(size_t)i * c.F + f]) {
                TiePair P; P.a = i; P.f = f; P.tiles = (an[i].gcount + kTsTile - 1) / kTsTile; P.pad = 0; P.v0 = (long long)v_total; P.m0 = (long long)m_total;
                v_total += (size_t)an[i].gcount; m_total += (size_t)an[i].gcount; tiles_max = std::max(tiles_max, P.tiles);
                pairs.push_back(P);
            }
    const int npairs = (int)pairs.size();
    bool walk = walk_early || npairs == 0 || (size_t)kTsWaves * c.TS * 4 > (size_t)60 * 1024;      // (huge threshold tables: the sort's cursors would not fit the LDS)
    if (sharded) walk = false;                // the walk reads this rank's documents only; rl_init keeps the tie-break off for sharded runs with huge tables
    std::vector<int32_t> cnts((size_t)npairs * c.TS);
    if (!walk) {       // cumulative bin counts of the pairs (exact): where every bin's run starts in the sorted values
        { int rcp = ensure_pin(cnts.size() * sizeof(int32_t) + 4096); if (rcp) return rcp; pin = (char *)t->tie_pin; }
        for (int p = 0; p < npairs; p++)
            RL_HIP(hipMemcpyAsync(pin + (size_t)p * c.TS * sizeof(int32_t), c.cum_cnt + ((size_t)an[pairs[p].a].node * c.F + pairs[p].f) * c.TS, (size_t)c.TS * sizeof(int32_t),
                                  hipMemcpyDeviceToHost, s));
        RL_HIP(hipStreamSynchronize(s));
        memcpy(cnts.data(), pin, cnts.size() * sizeof(int32_t));
    }
    std::vector<TieChain> chs; std::vector<int32_t> win_chain, chunk_chain;
    auto add_chain = [&](long long off, int len, int out) {
        TieChain C; C.off = off; C.len = len; C.out = out; C.win0 = (int32_t)win_chain.size();
        C.win = std::min(16384, std::max(2048, ((len / 256 + 2047) / 2048) * 2048));
        const int nw = (len + C.win - 1) / C.win;
        for (int j = 0; j < nw; j++) win_chain.push_back((int32_t)chs.size());
        for (int j = 0; j <= nw; j++) chunk_chain.push_back((int32_t)chs.size());       // chunk ids: win0 + chain index + local chunk
        chs.push_back(C);
    };
    std::vector<int32_t> &h_nthr = t->h_nthr;
    if (!walk) {
        if ((int)h_nthr.size() != c.F) { h_nthr.resize(c.F); RL_HIP(hipMemcpy(h_nthr.data(), c.nthr, c.F * sizeof(int32_t), hipMemcpyDeviceToHost)); }
        for (int i = 0; i < nA; i++) add_chain(u0[i], an[i].gcount, -(i + 1));
        for (int p = 0; p < npairs; p++) {
            const int32_t *cc = cnts.data() + (size_t)p * c.TS;
            for (int b = 0; b < h_nthr[pairs[p].f]; b++) {
                const int start = b > 0 ? cc[b - 1] : 0;
                add_chain(pairs[p].v0 + start, cc[b] - start, (int)(((size_t)pairs[p].a * c.F + pairs[p].f) * c.TS + b));
            }
        }
    }
    const int nch = (int)chs.size(), nwin = (int)win_chain.size(), nchunks = (int)chunk_chain.size();
    size_t l_u = 0, l_m = 0;                 // this rank's members of the chain nodes / of the pairs' chain nodes
    for (int i = 0; i < nA; i++) l_u += (size_t)lcnt[i];
    for (int p = 0; p < npairs; p++) l_m += (size_t)lcnt[pairs[p].a];
    const size_t spec_bytes = (m_total + 64) * 2 + (sharded ? (l_u + u_total + 64) * 8 + (l_m + m_total + 64) * 2 + (size_t)(R + 1) * nA * 4 : 0) + (size_t)(nA + 2 * npairs + 8) * 8 + v_total * 8 + (size_t)npairs * tiles_max * c.TS * 4 + (size_t)nwin * (16 + 16 + 4) + (size_t)nchunks * (4 + 16 + 8 + 8 * kSpW + 4) +
                              (size_t)nch * (16 + 8 + 8 + sizeof(TieChain)) + (size_t)npairs * sizeof(TiePair) + (size_t)(nwin + nchunks) * 4 + 64 * 256;
    {
        // the arena has to grow: it moves, so stage 1 runs again in the new one (and later calls ask for this much up front)
        const bool grow = !walk && fixed_bytes + spec_bytes + ((size_t)1 << 20) > t->tie_cap;
        int oom = 0;
        if (grow) {
            t->tie_hint = fixed_bytes + spec_bytes + ((size_t)1 << 20);
            if (tie_arena_reserve(t, t->tie_hint)) oom = 1;
        }
        if (sharded) {
            // spec_bytes and the free memory differ from rank to rank, the exchange below does not: the out-of-memory decision is taken by ALL ranks
            // (a rank that fell back to the walk on its own would leave the others waiting in the all-to-all -- and the walk sums its own documents
            // only).  One 4-byte all-reduce per resolution of a sharded run.
            if (!t->d_tie_flag) RL_HIP(t->pool.alloc(&t->d_tie_flag, (size_t)4));
            int32_t *d_oom = t->d_tie_flag;
            RL_HIP(hipMemcpyAsync(d_oom, &oom, sizeof(oom), hipMemcpyHostToDevice, s));
            int rcd = t->dist->allreduce(d_oom, 1, DT_I32, OP_MAX, s);
            if (rcd) return rcd;
            int32_t any = 0;
            RL_HIP(hipMemcpyAsync(&any, d_oom, sizeof(any), hipMemcpyDeviceToHost, s));
            RL_HIP(hipStreamSynchronize(s));
            if (any) return fail(RL_ERR_HIP, "tie-break: out of device memory on a rank of the job (the sharded tie-break needs the gathered chains on every rank)");
            if (grow) { int rc1 = stage1(); if (rc1) return rc1; }
        } else if (grow) {
            if (oom) {      // no room for the contiguous chains: the literal walk in a minimal arena
                walk = true;
                if (tie_arena_reserve(t, fixed_bytes + ((size_t)1 << 20))) return fail(RL_ERR_HIP, "tie-break: out of device memory");
            }
            int rc1 = stage1(); if (rc1) return rc1;
        }
    }
    if (walk) {
        RL_HIP(hipMemsetAsync(a.jbin, 0, (size_t)nA * c.F * c.TS * sizeof(double), s));
        hipLaunchKernelGGL(k_tie_jsum, dim3(c.F, nbg + 1, nA), dim3(64), 0, s, c, a, nbg);
    } else {
        SpArgs sp; memset(&sp, 0, sizeof(sp));
        sp.nchains = nch; sp.nwin = nwin; sp.nchunks = nchunks; sp.npairs = npairs; sp.tiles_max = tiles_max;
        std::vector<char> &blob = t->tie_blob; blob.clear();
        auto put = [&](const void *src, size_t bytes) { const size_t o = (blob.size() + 15) & ~(size_t)15; blob.resize(o + bytes); if (bytes) memcpy(blob.data() + o, src, bytes); return o; };
        const size_t o_pairs = put(pairs.data(), npairs * sizeof(TiePair)), o_chs = put(chs.data(), nch * sizeof(TieChain));
        const size_t o_winc = put(win_chain.data(), nwin * sizeof(int32_t)), o_chunkc = put(chunk_chain.data(), nchunks * sizeof(int32_t));
        std::vector<long long> m0s((size_t)npairs); std::vector<int32_t> pair_a((size_t)npairs);
        for (int p2 = 0; p2 < npairs; p2++) { m0s[p2] = pairs[p2].m0; pair_a[p2] = pairs[p2].a; }
        const size_t o_m0 = put(m0s.data(), npairs * sizeof(long long)), o_paira = put(pair_a.data(), npairs * sizeof(int32_t));
        char *d_blob = ar.take<char>(blob.size() + 16);
        RL_HIP(hipMemcpyAsync(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice, s));
        TiePair *d_pairs = (TiePair *)(d_blob + o_pairs); TieChain *d_chs = (TieChain *)(d_blob + o_chs);
        int32_t *d_winc = (int32_t *)(d_blob + o_winc), *d_chunkc = (int32_t *)(d_blob + o_chunkc);
        long long *d_m0 = (long long *)(d_blob + o_m0); int32_t *d_paira = (int32_t *)(d_blob + o_paira);
        (void)d_paira;
        sp.vals = ar.take<double>(v_total + 1); sp.tbin = ar.take<int32_t>((size_t)npairs * tiles_max * c.TS);
        sp.wsum = ar.take<double2>(nwin + 1); sp.wpre = ar.take<double2>(nwin + 1);
        sp.cstart = ar.take<int32_t>(nchunks + 1); sp.cpre = ar.take<double2>(nchunks + 1);
        sp.centre = ar.take<unsigned long long>(nchunks + 1); sp.table = ar.take<unsigned long long>((size_t)nchunks * kSpW + 1);
        sp.cstate = ar.take<int32_t>((size_t)nch * 4); sp.ckey = ar.take<unsigned long long>(nch); sp.cshift = ar.take<double>(nch);
        sp.open = ar.take<int32_t>(4);
        if (ar.used > t->tie_cap) return fail(RL_ERR_HIP, "tie-break: scratch arena too small (internal error)");
        sp.pairs = d_pairs; sp.chains = d_chs; sp.win_chain = d_winc; sp.chunk_chain = d_chunkc; sp.u0 = d_u0;
        sp.mb = ar.take<uint16_t>(m_total + 64);
        const dim3 ggrid_u(std::min(4096, (maxcnt + kThreads - 1) / kThreads), nA), ggrid_m(std::min(4096, (maxcnt + kThreads - 1) / kThreads), std::max(npairs, 1));
        if (!sharded) {
            // one GPU: this rank's members ARE the members -- lambda and bins go straight to their global places
            hipLaunchKernelGGL(k_tie_gather, ggrid_u, dim3(kThreads), 0, s, c, a, sp.vals, (const long long *)d_u0);
            if (npairs > 0) hipLaunchKernelGGL(k_tie_gather_bins, ggrid_m, dim3(kThreads), 0, s, c, a, (const TiePair *)d_pairs, sp.mb, (const long long *)d_m0);
        } else {
            // sharded: rank order is global document order, so the global arrays are the ranks' pieces behind each other.  Every rank gathers its
            // own pieces, all ranks exchange them (an all-gather of variable pieces through the all-to-all primitive), and every rank then runs the
            // SAME evaluation on the same global arrays -- the decision is rank-invariant by construction.
            std::vector<long long> lu0((size_t)nA), lm0((size_t)std::max(npairs, 1));
            { long long o = 0; for (int i = 0; i < nA; i++) { lu0[i] = o; o += lcnt[i]; } }
            { long long o = 0; for (int p2 = 0; p2 < npairs; p2++) { lm0[p2] = o; o += lcnt[pairs[p2].a]; } }
            long long *d_lu0 = ar.take<long long>(nA), *d_lm0 = ar.take<long long>(std::max(npairs, 1));
            int32_t *d_lcnt = ar.take<int32_t>(nA), *d_cntR = ar.take<int32_t>((size_t)R * nA);
            double *d_ul = ar.take<double>(l_u + 8), *d_urecv = ar.take<double>(u_total + 8);
            uint16_t *d_mbl = ar.take<uint16_t>(l_m + 8), *d_mbrecv = ar.take<uint16_t>(m_total + 8);
            if (ar.used > t->tie_cap) return fail(RL_ERR_HIP, "tie-break: scratch arena too small (internal error)");
            RL_HIP(hipMemcpyAsync(d_lu0, lu0.data(), nA * sizeof(long long), hipMemcpyHostToDevice, s));
            if (npairs > 0) RL_HIP(hipMemcpyAsync(d_lm0, lm0.data(), npairs * sizeof(long long), hipMemcpyHostToDevice, s));
            RL_HIP(hipMemcpyAsync(d_lcnt, lcnt.data(), nA * sizeof(int32_t), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_tie_gather, ggrid_u, dim3(kThreads), 0, s, c, a, d_ul, (const long long *)d_lu0);
            if (npairs > 0) hipLaunchKernelGGL(k_tie_gather_bins, ggrid_m, dim3(kThreads), 0, s, c, a, (const TiePair *)d_pairs, d_mbl, (const long long *)d_lm0);
            int rcd = t->dist->allgather(d_lcnt, d_cntR, (size_t)nA * sizeof(int32_t), s);
            if (rcd) return rcd;
            std::vector<int32_t> cntR((size_t)R * nA);
            RL_HIP(hipMemcpyAsync(pin, d_cntR, cntR.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            RL_HIP(hipStreamSynchronize(s));
            memcpy(cntR.data(), pin, cntR.size() * sizeof(int32_t));
            std::vector<int64_t> scount(R), sdispl(R, 0), rcount(R), rdispl(R);
            {   // lambda pieces
                int64_t o = 0;
                for (int r = 0; r < R; r++) { int64_t n = 0; for (int i = 0; i < nA; i++) n += cntR[(size_t)r * nA + i]; rcount[r] = n * 8; rdispl[r] = o; o += n * 8; scount[r] = (int64_t)l_u * 8; }
                if ((size_t)o != u_total * 8) return fail(RL_ERR_COMM, "tie-break: the ranks' member counts do not add up to the nodes' document counts");
                rcd = t->dist->alltoallv(d_ul, scount.data(), sdispl.data(), d_urecv, rcount.data(), rdispl.data(), s);
                if (rcd) return rcd;
                hipLaunchKernelGGL(k_tie_place<double>, dim3(nA, R), dim3(kThreads), 0, s, (const double *)d_urecv, sp.vals, (const int32_t *)d_cntR, (const int32_t *)nullptr,
                                   (const long long *)d_u0, nA, nA, R);
            }
            if (npairs > 0) {   // bins of the pairs
                int64_t o = 0;
                for (int r = 0; r < R; r++) { int64_t n = 0; for (int p2 = 0; p2 < npairs; p2++) n += cntR[(size_t)r * nA + pairs[p2].a]; rcount[r] = n * 2; rdispl[r] = o; o += n * 2; scount[r] = (int64_t)l_m * 2; }
                rcd = t->dist->alltoallv(d_mbl, scount.data(), sdispl.data(), d_mbrecv, rcount.data(), rdispl.data(), s);
                if (rcd) return rcd;
                hipLaunchKernelGGL(k_tie_place<uint16_t>, dim3(npairs, R), dim3(kThreads), 0, s, (const uint16_t *)d_mbrecv, sp.mb, (const int32_t *)d_cntR, (const int32_t *)d_paira,
                                   (const long long *)d_m0, npairs, nA, R);
            }
        }
        hipLaunchKernelGGL(k_ts_count, dim3(tiles_max, npairs), dim3(kThreads), (size_t)c.TS * 4, s, c, a, sp);
        hipLaunchKernelGGL(k_ts_scan, dim3(npairs, nbg), dim3(64), 0, s, c, a, sp);
        hipLaunchKernelGGL(k_ts_scatter, dim3(tiles_max, npairs), dim3(kTsWaves * 64), (size_t)kTsWaves * c.TS * 4, s, c, a, sp);
        const int cb = (nch + kThreads - 1) / kThreads;
        if (nwin > 0) hipLaunchKernelGGL(k_sp_sum, dim3(nwin), dim3(64), 0, s, sp);
        hipLaunchKernelGGL(k_sp_scan, dim3(cb), dim3(kThreads), 0, s, sp);
        if (nwin > 0) hipLaunchKernelGGL(k_sp_bounds, dim3(nwin), dim3(64), 0, s, sp);
        hipLaunchKernelGGL(k_sp_run<false>, dim3(nchunks), dim3(kSpW), 0, s, sp);
        hipLaunchKernelGGL(k_sp_drift, dim3(cb), dim3(kThreads), 0, s, sp);
        hipLaunchKernelGGL(k_sp_run<false>, dim3(nchunks), dim3(kSpW), 0, s, sp);
        int32_t open = 0;
        for (int rep = 0; rep <= kSpRepairs; rep++) {
            RL_HIP(hipMemsetAsync(sp.open, 0, sizeof(int32_t), s));
            hipLaunchKernelGGL(k_sp_stitch, dim3(cb), dim3(kThreads), 0, s, sp, a, rep == kSpRepairs ? 1 : 0);
            RL_HIP(hipMemcpyAsync(pin, sp.open, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            RL_HIP(hipStreamSynchronize(s));
            memcpy(&open, pin, sizeof(open));
            if (open == 0) break;
            t->tie_spec_repairs++;
            hipLaunchKernelGGL(k_sp_run<true>, dim3(nchunks), dim3(kSpW), 0, s, sp);
        }
        if (c.steplog) {       // debug statistics (RLHIP_STEPLOG): window misses / serial chunks of this resolution
            std::vector<int32_t> cst((size_t)nch * 4);
            RL_HIP(hipMemcpyAsync(cst.data(), sp.cstate, cst.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            RL_HIP(hipStreamSynchronize(s));
            for (int i = 0; i < nch; i++) { t->tie_spec_miss += cst[4 * (size_t)i + 1]; t->tie_spec_serial += cst[4 * (size_t)i + 2]; }
        }
        t->tie_spec_segs += nchunks;
    }
    mark(3);
    // prefixes of all needed rows at once, the tied candidates of all (feature, node) pairs at once, then one block: arg-max, node records, select_step
    hipLaunchKernelGGL(k_tie_prefix, dim3(c.F, nA), dim3(64), (size_t)c.TS * 8, s, c, a);
    hipLaunchKernelGGL(k_tie_eval, dim3(c.F, nx), dim3(kFinThreads), 0, s, c, a);
    hipLaunchKernelGGL(k_tie_finish, dim3(1), dim3(kFinThreads), fin_lds, s, c, a, nodes_in_lds, deferred ? 1 : 0);
    RL_HIP(hipGetLastError());
    if (nv > 0 && sharded) {      // every rank checked its own documents: a cut that differs anywhere makes every rank grow the tree again
        int rcd = t->dist->allreduce(d_vflag, 1, DT_I32, OP_MAX, s);
        if (rcd) return rcd;
    }
    if (nv > 0) RL_HIP(hipMemcpyAsync(pin, d_vflag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    RL_HIP(hipStreamSynchronize(s));
    mark(4);
    if (nv > 0 && other_cut) { int32_t fl = 0; memcpy(&fl, pin, sizeof(fl)); *other_cut = (fl != 0) || getenv("RLHIP_TIE_FORCE_REGROW") != nullptr; }
    t->tie_stalls++; t->tie_nodes += nx; t->tie_chain_nodes += nA; if (deferred) t->tie_batches++;
    for (auto &A : an) t->tie_chain_docs += A.count;
    t->tie_us += (long long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_begin).count();
    return RL_OK;
}

static int enqueue_round(rl_trainer *t)
{
    Ctx &c = t->ctx;
    hipStream_t s = t->stream;
    const int m = t->round;
    // round scalars
    RL_HIP(hipMemsetAsync(&c.st->maxabs_bits, 0, sizeof(unsigned long long) + sizeof(long long), s));
    int n_max = 0;       // blocks that reported their max |lambda| (folded by k_max_reduce)
    if (c.mart) {    // MART: residuals instead of lambdas (weights stay 0)
        ScopedTiming tm(t, RL_KERNEL_LAMBDA, (double)c.N * 20.0);
        n_max = std::min(2048, (c.N + kThreads - 1) / kThreads);
        hipLaunchKernelGGL(k_mart_residual, dim3(n_max), dim3(kThreads), 0, s, c.labels, (const double *)c.scores, c.lw, c.N, t->d_wmax);
    } else {   // K1 lambdas: pair terms in parallel, then ordered accumulation (ranked order comes from the previous
        // round's k_rank_* / from rl_init for round 0)
        ScopedTiming tm(t, RL_KERNEL_LAMBDA, (double)c.N * 28.0);
        const double *ideal = (c.metric == RL_METRIC_NDCG) ? (m == 0 ? c.ideal0 : c.ideal1) : nullptr;
        LamArgs g{t->tr.d_ss, t->tr.d_sl, t->tr.d_srel, t->tr.d_sidx, c.qoff, t->tr.d_docq, ideal, c.disc,
                  t->d_T, c.lw, &c.st->maxabs_bits, c.N, c.k, c.k, c.metric, t->p.metric_k,
                  t->tr.d_aux_i, t->tr.d_aux_a, t->tr.d_aux_b, t->d_wmax, t->tr.d_ext_rd};
        if (t->d_T == nullptr) {
            const int mode = (c.metric == RL_METRIC_ERR) ? 1 : (c.metric == RL_METRIC_MAP) ? 2 : 0;
            // RLHIP_LAMBDA_COMPACT=1: NDCG / DCG pair terms from per-wavefront lists of the active pairs (k_lambda_fused<., 0, true>) instead of column by row.
            // Built and measured slower at NDCG@10 (profiles/r05i_ab_lambda_c2.txt: a wavefront's ~390 active pairs are 3.05 steps of 128, i.e. 4 against the 5 of
            // ten rows in pairs, and the lists cost LDS, registers and six ds_bpermute per pair); it is the shorter way from about NDCG@16 on.  Off by default.
            const bool cp = t->lam_compact && mode == 0;
            auto lds_of = [&](int bt) { return (size_t)c.k * (bt + 8) * 16 + (size_t)c.k * 24 + lambda_fused_extra_bytes(mode, c.k, bt) + (cp ? lambda_fused_cp_bytes(c.k, bt) : 0); };
            n_max = 0;
            const DataSet &d = t->tr;
            // (ls: the stream of this class -- the main one, or one of the three side streams forked below)
            int lam_used = 0;
            const bool fork = t->lam_streams;       // (sharded runs too, round 6: the classes only touch this rank's lists; the collectives follow on the main stream behind the join)
            if (fork) { RL_HIP(hipEventRecord(t->ev_lam_fork, s)); }
            auto lam_stream = [&]() -> hipStream_t {
                const int lam_side = t->lam_side;      // side streams used (the rest of the classes: the main stream).  One: the widest class beside the
                // three others in a row on the main stream -- c2 423.1 -> 426.5 rounds/s against three side streams, c1 / c3 / c1ns within their noise
                // (profiles/r05q_ab_lambda_side_*): the classes fill the chip either way, and a kernel that ends on a side stream is a cross-stream wait
                if (!fork || lam_used >= lam_side) return s;
                hipStream_t ls = t->lam_s[lam_used++];
                (void)hipStreamWaitEvent(ls, t->ev_lam_fork, 0);
                return ls;
            };
#define RL_LAUNCH_FUSED(BT, cls)                                                                                                             \
            if (d.n_qcls[cls] > 0) {                                                                                                         \
                hipStream_t ls = lam_stream();                                                                                               \
                if (cp) hipLaunchKernelGGL((k_lambda_fused<BT, 0, true>), dim3(d.n_qcls[cls]), dim3(BT), lds_of(BT), ls, g, (const int *)d.d_qcls[cls], d.n_qcls[cls]); \
                else if (mode == 0) hipLaunchKernelGGL((k_lambda_fused<BT, 0>), dim3(d.n_qcls[cls]), dim3(BT), lds_of(BT), ls, g, (const int *)d.d_qcls[cls], d.n_qcls[cls]); \
                else if (mode == 1) hipLaunchKernelGGL((k_lambda_fused<BT, 1>), dim3(d.n_qcls[cls]), dim3(BT), lds_of(BT), ls, g, (const int *)d.d_qcls[cls], d.n_qcls[cls]); \
                else hipLaunchKernelGGL((k_lambda_fused<BT, 2>), dim3(d.n_qcls[cls]), dim3(BT), lds_of(BT), ls, g, (const int *)d.d_qcls[cls], d.n_qcls[cls]); \
                n_max += d.n_qcls[cls]; g.blockmax = t->d_wmax + n_max;                                                                      \
            }
            if (d.n_qcls[4] > 0 && mode == 0) {
                const int nb = (d.n_qcls[4] + kLambdaTinyGroups - 1) / kLambdaTinyGroups;
                hipLaunchKernelGGL(k_lambda_tiny, dim3(nb), dim3(kLambdaTinyDocs * kLambdaTinyGroups),
                                   (size_t)kLambdaTinyGroups * lambda_tiny_group_bytes(c.k), s, g, (const int *)d.d_qcls[4], d.n_qcls[4]);
                n_max += nb; g.blockmax = t->d_wmax + n_max;
            } else {
                // ERR / MAP: the lists of at most 16 documents take the block-per-query kernel too -- on the main stream: the side stream is for the
                // widest class below (the first class launched used to take it)
                const int keep = lam_used; lam_used = 1 << 20;
                RL_LAUNCH_FUSED(64, 4)
                lam_used = keep;
            }
            // longest lists first on the side streams (they take longest per block), the shortest class last on the main stream
            RL_LAUNCH_FUSED(256, 3)
            RL_LAUNCH_FUSED(192, 2)
            RL_LAUNCH_FUSED(128, 1)
            RL_LAUNCH_FUSED(64, 0)
#undef RL_LAUNCH_FUSED
            for (int i = 0; i < lam_used; i++) { RL_HIP(hipEventRecord(t->ev_lam_join[i], t->lam_s[i])); RL_HIP(hipStreamWaitEvent(s, t->ev_lam_join[i], 0)); }
        } else {
            const unsigned nb = (unsigned)((c.N + kThreads - 1) / kThreads);
            hipLaunchKernelGGL(k_pair_terms, dim3(nb), dim3(kThreads), 0, s, g);
            hipLaunchKernelGGL(k_lambda_acc, dim3(nb), dim3(kThreads), 0, s, g);
            n_max = (int)nb;
        }
    }
    hipLaunchKernelGGL(k_max_reduce, dim3((unsigned)std::max(1, std::min(16, (n_max + 4095) / 4096))), dim3(1024), 0, s, (const double *)t->d_wmax, n_max, &c.st->maxabs_bits);
    if (t->dist) { int rcd = t->dist->allreduce(&c.st->maxabs_bits, 1, DT_U64, OP_MAX, s); if (rcd) return rcd; }
    // The plain one-GPU root pass makes the fixed-point lambdas itself (k_hist<.., FQ>): one pass over the documents and one launch less a round.
    // Sharded, strict-order and sparse-column runs (their kernels between here and the root pass read q) and a regrown tree (q exists) keep k_quantize.
    static const bool fq_env = !(getenv("RLHIP_FUSED_QUANT") && atoi(getenv("RLHIP_FUSED_QUANT")) == 0);
    bool root_quant_fused = fq_env && !c.java && !c.sp_on && c.sub == 16 && c.TS <= kHistLdsStride && !c.any_runs;       // (sharded runs too, round 6: max |lambda| is all-reduced before this point, nothing between here and the root pass reads q)
    if (!root_quant_fused) hipLaunchKernelGGL(k_quantize, dim3(std::min(2048, (c.N + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, c);
    const size_t hist_lds = (size_t)c.sub * ((c.sub == 16 && c.TS <= kHistLdsStride) ? kHistLdsStride : c.TS) * 12;    // int64 sums + int32 counts
    const int hist_gx = c.numFG * (kHistFG / c.sub);
    const size_t red_lds = (size_t)c.TS * 20;
    const int rootCs = std::min(kChunk, std::max(kMinChunk, (((c.N + 63) / 64 + 255) & ~255)));   // == chunk_docs<true>(N)
    const int rootChunks = (c.N + rootCs - 1) / rootCs;
    // (a tree is grown a second time, from its root histogram, when the deferred tie-break finds that a tie over several features it took for one
    // cut is not one -- rl_tie.inc, k_tie_verify; nothing a round keeps has been written by then)
    const int tie_mode = c.tie_on;
    struct TieModeRestore { Ctx &c; int v; ~TieModeRestore() { c.tie_on = v; } } tie_mode_restore{c, tie_mode};
  regrow:
    {   // K2 root histogram: dense groups from their 32-byte rows, groups of sparse columns from their entry lists (rl_csc.inc)
        const double root_bytes = c.sp_on ? (double)c.N * ((double)(c.numFG - c.sp_ngroups) * kHistFG * 2.0 + 8.0) + (double)t->sp_entries * 4.0
                                          : (double)c.N * ((double)c.F * 2.0 + 8.0);
        ScopedTiming tm(t, RL_KERNEL_HIST_ROOT, root_bytes);
        launch_hist<true>(c, hist_gx, rootChunks, hist_lds, s, root_quant_fused);
        root_quant_fused = false;        // (a regrown tree reads the q / r this pass has stored)
        if (c.sp_on) {
            hipLaunchKernelGGL(k_hist_sp<kHistLdsStride>, dim3(c.sp_ngroups, rootChunks), dim3(kSpThreads), (size_t)kHistFG * kHistLdsStride * 8, s, c, rootCs);
        }
    }
    // the last block of k_hist_finish runs the growth bookkeeping (select_step); node records live in LDS when they fit
    const int nodes_in_lds = (select_lds_bytes(c.L, c.NC, true, c.F, c.fs_on != 0) <= 60 * 1024) ? 1 : 0;
    const size_t par_lds = (c.TS <= kParCacheTS) ? (size_t)c.TS * 20 + 8 : 0;       // the parent's entries next to the bins (hist_finish_body)
    const size_t fin_lds = std::max((size_t)c.TS * (c.java ? 28 : 20) + 8 + par_lds, select_lds_bytes(c.L, c.NC, nodes_in_lds != 0, c.F, c.fs_on != 0));      // bins + the parent's entries (hist_finish_body)
    const int jbg = (c.TS + 63) / 64;        // RL_FLAG_JAVA_ORDER: 64-bin groups of k_jhist (+ 1 block for the node totals)
    if (fin_lds > 128 * 1024) return fail(RL_ERR_UNSUPPORTED, "too many features / leaves for the growth bookkeeping in LDS (feature sampling needs 64 bytes per feature)");
    if (t->dist) {
        hipLaunchKernelGGL(k_hist_reduce, dim3(c.F), dim3(kFinThreads), red_lds, s, c, 1);
        int rcd = t->dist->allreduce(c.dist_buf, (size_t)c.F * c.TS * c.limb_words + 4, DT_I64, OP_SUM, s);
        if (rcd) return rcd;
        hipLaunchKernelGGL((k_hist_finish<true, true>), dim3(c.F), dim3(kFinThreads), fin_lds, s, c, nodes_in_lds);
    } else if (c.java) {
        hipLaunchKernelGGL(k_jgather, dim3((c.N + kPartTile - 1) / kPartTile), dim3(kThreads), 0, s, c, 1);
        if (c.jmap) hipLaunchKernelGGL(k_jhist2, dim3(c.n_live + 1, 1), dim3(kJ2Threads), 0, s, c, 1, c.jmap, c.jinv, c.jone);
        else hipLaunchKernelGGL(k_jhist, dim3(c.n_live, jbg + 2, 1), dim3(64), 0, s, c, 1, jbg);
        hipLaunchKernelGGL((k_hist_finish<true, false, true>), dim3(c.n_live), dim3(kFinThreads), fin_lds, s, c, nodes_in_lds);
    } else if (t->step2 && c.TS <= kFin2MaxT) {       // rl_step2.inc
        hipLaunchKernelGGL(k_fin2_root, dim3(c.n_live), dim3(kFin2RootThreads), 0, s, c, rootChunks);
        hipLaunchKernelGGL(k_select_root, dim3(1), dim3(kFinThreads), fin_lds, s, c, nodes_in_lds);
    } else hipLaunchKernelGGL((k_hist_finish<true, false>), dim3(c.n_live), dim3(kFinThreads), fin_lds, s, c, nodes_in_lds);
    // Growth steps: each prepares up to kSpec queue nodes and commits as many splits as the fit loop allows; L-1 steps
    // always suffice (every step commits at least the head of the queue); finished trees make the rest no-ops.
    const int steps = std::max(c.L - 1, 1);
    const size_t slot_words = (size_t)c.F * c.TS * c.limb_words + 4;
    t->tree_seq++;
    bool throttle = c.progress != nullptr;
    // lazy tie-break (rl_tie.inc): the device may STALL the tree (no slots, progress word bit 31) until resolve_ties has run.  Growth kernels
    // enqueued meanwhile are no-ops; afterwards the host carries on from the device's own step count.  `extra`: the stalled select_step call and
    // its resumption count as steps of the device without committing a split.
    const auto stalled = [&](unsigned long long w) { return c.tie_on && (w >> 32) == t->tree_seq && ((w >> 31) & 1ull); };
    bool defer_seen = false;       // the finished tree holds nodes whose stored threshold awaits the batched tie-break (progress word bit 30)
    int extra = 0, it = 0;
    bool saw_end = false;
    auto after_stall = [&](bool &ended) -> int {        // stream idle, tree stalled: resolve, then continue at the device's step
        int rcs = resolve_ties(t, fin_lds, nodes_in_lds);
        if (rcs) return rcs;
        extra += 2;
        if (c.progress) {       // resolve_ties waited for k_tie_finish, whose select_step left (step, done, deferred ties) in the pinned progress word
            const unsigned long long w = __atomic_load_n(t->h_progress, __ATOMIC_ACQUIRE);
            if ((w >> 32) != t->tree_seq) return fail(RL_ERR_STATE, "tie-break: the progress word is not this tree's (internal error)");
            ended = (w & 1ull) != 0;
            if (ended) defer_seen = ((w >> 30) & 1ull) != 0;
            it = (int)((unsigned)(w & 0x3fffffffull) >> 1);
            return RL_OK;
        }
        TreeState sth;
        RL_HIP(hipMemcpy(&sth, c.st, sizeof(sth), hipMemcpyDeviceToHost));
        ended = sth.done != 0;
        if (ended) defer_seen = sth.defer_any != 0;
        it = sth.step;
        return RL_OK;
    };
  grow:
    for (; it < steps + extra; it++) {
        if (throttle && c.tie_on && !t->dist) {
            const unsigned long long w0 = __atomic_load_n(t->h_progress, __ATOMIC_ACQUIRE);
            if (stalled(w0)) {       // (resolve_ties reads the tree state through the stream: no drain of its own needed here)
                bool ended = false;
                int rcs = after_stall(ended);
                if (rcs) return rcs;
                if (ended) { saw_end = true; break; }
                it--;
                continue;
            }
        }
        const int ahead = t->dist ? t->dist_ahead : t->step_ahead;
        if (throttle && it >= ahead) {
            // wait (bounded) until growth step it - step_ahead has been selected, then look at the tree's done flag.  Purely a
            // scheduling hint: on a timeout the remaining steps are enqueued blindly, which is always correct.
            const unsigned long long want = ((unsigned long long)t->tree_seq << 32) | ((unsigned long long)(unsigned)(it - ahead) << 1);
            const auto finished = [&](unsigned long long w) { return (w >> 32) == t->tree_seq && (w & 1); };
            unsigned long long w = __atomic_load_n(t->h_progress, __ATOMIC_ACQUIRE);
            if (t->dist) {
                // Deterministic over the ranks (they grow the same tree): wait -- without a timeout -- until growth step
                // it - step_ahead has been selected or the tree is finished, and stop only if the tree was finished by a step
                // <= it - step_ahead.  The word keeps (step at which `done` was set, done) once the tree is finished, so a rank
                // that looks early and one that looks late take the same decision at the same `it`.
                unsigned spins = 0;
                const auto t0w = std::chrono::steady_clock::now();
                while ((w & ~(3ull << 30)) < want && !finished(w) && !stalled(w)) {       // (bits 30 / 31 of the low word are flags)
                    w = __atomic_load_n(t->h_progress, __ATOMIC_ACQUIRE);
                    if ((++spins & 0xfffff) == 0) {
                        const hipError_t q = hipStreamQuery(s);
                        if (q != hipSuccess && q != hipErrorNotReady) return fail(RL_ERR_HIP, std::string("device error while growing a tree: ") + hipGetErrorString(q));
                        // the wait ends when THIS rank's device finishes a growth step, which needs every other rank's share of the step's
                        // collective: a rank that died or fell behind for good must surface as an error here, not as a hang
                        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0w).count() > t->dist_timeout_s)
                            return fail(RL_ERR_COMM, "timed out after " + std::to_string((int)t->dist_timeout_s) + " s waiting for growth step " + std::to_string(it - ahead) +
                                                     " of tree " + std::to_string(t->tree_seq) + " (a rank of the job is missing from a collective?)");
                    }
                }
                const int step_w = (int)((unsigned)(w & 0x3fffffffull) >> 1);
                if (finished(w) && step_w <= it - ahead) { saw_end = true; defer_seen = ((w >> 30) & 1ull) != 0; break; }
                // a stalled tree (rl_tie.inc): the word keeps the step at which it stalled, so -- as for the end of the tree -- every rank acts on
                // it at the same `it`, after the same number of (empty) steps and their collectives
                if (stalled(w) && step_w <= it - ahead) {
                    bool ended = false;
                    int rcs = after_stall(ended);
                    if (rcs) return rcs;
                    if (ended) { saw_end = true; break; }
                    it--;
                    continue;
                }
            } else {
                // (bits 30 / 31 of the low word are flags: the step is compared field by field)
                const auto behind = [&](unsigned long long v) {
                    if ((v >> 32) != t->tree_seq) return true;                   // still the previous tree's word
                    return (unsigned)((v & 0x3fffffffull) >> 1) < (unsigned)(it - ahead) && !(v & 1) && !((v >> 31) & 1);
                };
                if (behind(w)) {
                    const auto t0 = std::chrono::steady_clock::now();
                    unsigned spins = 0;
                    while (behind(w = __atomic_load_n(t->h_progress, __ATOMIC_ACQUIRE))) {
                        if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) { throttle = false; break; }
                    }
                }
            }
            if (!t->dist && stalled(w)) { it--; continue; }       // handled at the top of the loop
            if (!t->dist && finished(w)) { saw_end = true; defer_seen = ((w >> 30) & 1ull) != 0; break; }
        }
        if (t->dist && !c.cum_cnt_loc) {      // local child sizes unknown in advance: count pass, then scatter (round 5; RLHIP_DIST_COUNT_PASS=1)
            hipLaunchKernelGGL(k_part_count, dim3(c.nTiles), dim3(kThreads), 0, s, c);
            hipLaunchKernelGGL(k_part_scatter<false>, dim3(c.nTiles), dim3(kThreads), 0, s, c);
        } else hipLaunchKernelGGL(k_part_scatter<true>, dim3(c.nTiles), dim3(kThreads), 0, s, c);
        {
            ScopedTiming tm(t, RL_KERNEL_HIST_NODE, 0.0);
            launch_hist<false>(c, hist_gx, c.maxChunks, hist_lds, s);
        }
        if (t->dist) {
            hipLaunchKernelGGL(k_hist_reduce, dim3(c.F, kSpec), dim3(kFinThreads), red_lds, s, c, 0);
            // growth step `it` works on at most min(kSpec, 2^it) nodes (1 after the root, then at most twice the commits of the step
            // before): the first steps -- the ones with the largest histograms to wait for -- reduce one or two slots, not kSpec
            const int max_slots = std::min(kSpec, 1 << std::min(it, 8));
            int rcd = t->dist->allreduce(c.dist_buf, slot_words * max_slots, DT_I64, OP_SUM, s);
            if (rcd) return rcd;
            // round 6: the finish and the bookkeeping of the plain path's step (rl_step2.inc) on the all-reduced limbs; the round-4 fused kernel for what
            // k_select2 does not cover (feature sampling, more than 160 features, more than 62 leaves)
            const size_t sel2_lds_d = select2_lds_bytes(c.L, c.NC);
            if (t->step2 && c.TS <= kFin2MaxT && !c.fs_on && c.F <= kSel2MaxF && c.L > 0 && c.L + 2 <= 64 && sel2_lds_d <= 60 * 1024) {
                hipLaunchKernelGGL(k_fin2<true>, dim3(c.F, kSpec), dim3(kFin2Threads), 0, s, c);
                hipLaunchKernelGGL(k_select2<false>, dim3(1), dim3(kSel2Threads), sel2_lds_d, s, c);
            } else hipLaunchKernelGGL((k_hist_finish<false, true>), dim3(c.F, kSpec), dim3(kFinThreads), fin_lds, s, c, nodes_in_lds);
            // Sharded runs pay a collective per step even when the tree is already finished, so the host looks at the
            // (rank-invariant) `done` flag now and then and stops enqueuing: a stream sync costs far less than the
            // all-reduces of ~20 empty steps.  One GPU keeps the fully asynchronous schedule (an empty step is 3 tiny launches).
            if (!c.progress && it + 1 < steps + extra && ((it >= 7 && (it - 7) % 3 == 0) || c.tie_on)) {
                TreeState sth;
                RL_HIP(hipStreamSynchronize(s));
                RL_HIP(hipMemcpy(&sth, c.st, sizeof(sth), hipMemcpyDeviceToHost));
                if (sth.stall_n > 0) {       // (without a progress word every step of a tie-breaking sharded run is looked at: rank-invariant by construction)
                    bool ended = false;
                    int rcs = after_stall(ended);
                    if (rcs) return rcs;
                    if (ended) { saw_end = true; break; }
                    it--;
                    continue;
                }
                if (sth.done) { saw_end = true; defer_seen = sth.defer_any != 0; break; }
            }
        } else if (c.java) {
            hipLaunchKernelGGL(k_jgather, dim3(c.nTiles), dim3(kThreads), 0, s, c, 0);
            if (c.jmap) hipLaunchKernelGGL(k_jhist2, dim3(c.n_live + 1, kSpec), dim3(kJ2Threads), 0, s, c, 0, c.jmap, c.jinv, c.jone);
            else hipLaunchKernelGGL(k_jhist, dim3(c.n_live, jbg + 2, kSpec), dim3(64), 0, s, c, 0, jbg);
            hipLaunchKernelGGL((k_hist_finish<false, false, true>), dim3(c.n_live, kSpec), dim3(kFinThreads), fin_lds, s, c, nodes_in_lds);
        } else if (t->step2 && c.TS <= kFin2MaxT) {
            // rl_step2.inc: one bin per thread, DPP scans, plain stores -- then the bookkeeping as a launch of its own
            hipLaunchKernelGGL(k_fin2<false>, dim3(c.n_live, kSpec), dim3(kFin2Threads), 0, s, c);
            const size_t sel2_lds = select2_lds_bytes(c.L, c.NC);
            if (!c.fs_on && c.F <= kSel2MaxF && c.L > 0 && c.L + 2 <= 64 && sel2_lds <= 60 * 1024)
                hipLaunchKernelGGL(k_select2<false>, dim3(1), dim3(kSel2Threads), sel2_lds, s, c);
            else if (t->sel2_wide && !c.fs_on && c.F <= 32 * kWideS && c.L > 0 && c.L + 2 <= 64 && sel2_lds <= 60 * 1024)
                hipLaunchKernelGGL(k_select2<true>, dim3(1), dim3(kSel2Threads), sel2_lds, s, c);       // (hundreds of features: the Yahoo-set1 shape)
            else hipLaunchKernelGGL(k_select, dim3(1), dim3(kFinThreads), fin_lds, s, c, nodes_in_lds);
        } else if (t->fin_split || !nodes_in_lds) {       // (wide data; or node records that do not fit the LDS: the fused kernel has no path for them)
            hipLaunchKernelGGL(k_hist_finish_wide, dim3(c.n_live, kSpec), dim3(kFinWideThreads), (size_t)c.TS * 20 + 8 + par_lds, s, c);
            hipLaunchKernelGGL(k_select, dim3(1), dim3(kFinThreads), fin_lds, s, c, nodes_in_lds);
        } else hipLaunchKernelGGL((k_hist_finish<false, false>), dim3(c.n_live, kSpec), dim3(kFinThreads), fin_lds, s, c, nodes_in_lds);
    }
    if (c.tie_on && !saw_end) {
        // every step was enqueued without the host ever seeing the end of the tree (no progress word, a spin timeout, or a tree of fewer steps
        // than the host keeps in flight): a stall may have gone unnoticed -- look, once per round
        RL_HIP(hipStreamSynchronize(s));
        TreeState sth;
        RL_HIP(hipMemcpy(&sth, c.st, sizeof(sth), hipMemcpyDeviceToHost));
        if (sth.stall_n > 0) {
            bool ended = false;
            int rcs = after_stall(ended);
            if (rcs) return rcs;
            if (!ended) goto grow;
        } else defer_seen = sth.defer_any != 0;
    }
    // the score update streams over the documents when the leaf sums' gather can leave every document's leaf behind (one GPU, parallel chains, <= 1024 leaves)
    const bool stream_env = !(getenv("RLHIP_SCORE_STREAM") && atoi(getenv("RLHIP_SCORE_STREAM")) == 0);        // (read per round: a test switches it)
    const bool stream_scores = stream_env && !(t->p.flags & RL_FLAG_SERIAL_CHAIN) && c.leaf_of != nullptr && c.L > 0 && c.L <= 1024;       // (sharded runs too, round 6: the local gather in leaf order leaves every document's leaf behind)
    hipLaunchKernelGGL(k_leaf_table, dim3(1), dim3(kThreads), 0, s, c, t->leaf_chain, t->d_seg_buf);
    if (t->p.flags & RL_FLAG_SERIAL_CHAIN) {
        hipLaunchKernelGGL(k_leaf_chain, dim3(c.L), dim3(64), 0, s, c);
    } else if (t->dist) {
        // multi-GPU: gather lambda / weight in leaf order from every rank and evaluate the chains over the whole leaf
        // multi-GPU, the leaf-owner exchange (rl_dist.inc): lambda / weight of a leaf's documents go to the leaf's owner rank only
        ChainSource src{nullptr, nullptr, c.lw, c.idx[0], c.idx[1], t->d_seg_buf, stream_scores ? c.leaf_of : nullptr};
        if (t->piece_chains) {      // round 6: every rank evaluates its own pieces of every leaf, only tables travel (rl_dist.inc)
            int rcp = enqueue_leaf_chains_pieces(t, src);
            if (rcp) return rcp;
            hipLaunchKernelGGL(k_leaf_output, dim3((c.L + kThreads - 1) / kThreads), dim3(kThreads), 0, s, c, t->leaf_chain);
        } else {
        const ChainBufs &lb = t->leaf_chain;
        const int R = t->n_ranks, me = t->dist->rank, nseg = std::max(c.L, 2), MS = t->gchain.maxseg;      // -leaf 1 still has two leaves (the root always splits)
        hipLaunchKernelGGL(k_chain_prefix, dim3((unsigned)((lb.cap_tiles + 3) / 4)), dim3(kThreads), 0, s, lb, src);      // local values in leaf order -> lb.xs
        int rcd = t->dist->allgather(c.leaf_start, t->d_gls, (size_t)t->lsstride * sizeof(int32_t), s);
        if (rcd) return rcd;
        std::vector<int64_t> scount(R), sdispl(R), rcount(R), rdispl(R);
        const bool dev_plan = t->d_xmail != nullptr && !getenv("RLHIP_DIST_HOST_PLAN") && nseg <= kPlanMaxSeg && R <= 64;       // (RLHIP_DIST_HOST_PLAN=1: the host plan behind a stream synchronisation, as until round 5)
        if (dev_plan) {
            // the plan on the device; the host only needs the byte counts of the transfers and reads them from a pinned mailbox below, after it has
            // enqueued the pack kernel (k_plan_exchange)
            hipLaunchKernelGGL(k_plan_exchange, dim3(1), dim3(64), 0, s, (const int32_t *)t->d_gls, R, t->lsstride, nseg, MS, me, t->d_own, t->d_xtab, t->d_xmail, ++t->xmail_tag);
        } else {
            // the send / receive counts of the exchange have to be known to the host: one small copy per round (sharded runs are host-paced anyway)
            std::vector<int32_t> &gls = t->h_gls;
            gls.resize((size_t)R * t->lsstride);
            RL_HIP(hipMemcpyAsync(gls.data(), t->d_gls, gls.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            RL_HIP(hipStreamSynchronize(s));
            auto len_of = [&](int r, int l) { return (long long)gls[(size_t)r * t->lsstride + l + 1] - gls[(size_t)r * t->lsstride + l]; };
            std::vector<int32_t> &own = t->h_own; own.assign((size_t)MS, 0);
            {   // owners: largest leaf first onto the least loaded rank (every rank computes the same map from the same table)
                std::vector<long long> glen((size_t)nseg, 0), load((size_t)R, 0);
                std::vector<int32_t> order((size_t)nseg);
                for (int l = 0; l < nseg; l++) { order[l] = l; for (int r = 0; r < R; r++) glen[l] += len_of(r, l); }
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return glen[a] > glen[b]; });
                for (int l : order) {
                    if (glen[l] == 0) { own[l] = l % R; continue; }
                    int o = 0;
                    for (int r = 1; r < R; r++) if (load[r] < load[o]) o = r;
                    own[l] = o; load[o] += glen[l];
                }
            }
            std::vector<long long> &tab = t->h_xtab; tab.assign((size_t)MS * (R + 1), 0);       // pack_off [MS] | asm_off [R][MS]
            long long cur = 0;
            for (int d = 0; d < R; d++) {
                sdispl[d] = cur * 8;
                if (d != me) for (int l = 0; l < nseg; l++) if (own[l] == d) { tab[l] = cur; cur += 2 * len_of(me, l); }       // (own leaves: not packed, k_chain_assemble)
                scount[d] = cur * 8 - sdispl[d];
            }
            cur = 0;
            for (int r = 0; r < R; r++) {
                rdispl[r] = cur * 8;
                if (r != me) for (int l = 0; l < nseg; l++) if (own[l] == me) { tab[(size_t)MS * (1 + r) + l] = cur; cur += 2 * len_of(r, l); }
                rcount[r] = cur * 8 - rdispl[r];
            }
            RL_HIP(hipMemcpyAsync(t->d_own, own.data(), (size_t)MS * sizeof(int32_t), hipMemcpyHostToDevice, s));
            RL_HIP(hipMemcpyAsync(t->d_xtab, tab.data(), tab.size() * sizeof(long long), hipMemcpyHostToDevice, s));
        }
        const LeafExchange lx{t->d_own, t->d_xtab, t->d_xtab + MS};
        hipLaunchKernelGGL(k_chain_pack, dim3(nseg, 2, kLeafXferZ), dim3(kThreads), 0, s, (const double *)lb.xs, lb.cap_n, (const int32_t *)c.leaf_start, nseg, lx, t->d_send, me);
        if (dev_plan) {       // the mailbox: the tag is stored last (release); a device error or a dead peer must end the wait
            const auto t0w = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (__atomic_load_n(&t->h_xmail[4 * R], __ATOMIC_ACQUIRE) != t->xmail_tag) {
                if ((++spins & 0xffff) == 0) {
                    const hipError_t q = hipStreamQuery(s);
                    if (q != hipSuccess && q != hipErrorNotReady) return fail(RL_ERR_HIP, std::string("device error before the leaf-owner exchange: ") + hipGetErrorString(q));
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0w).count() > t->dist_timeout_s)
                        return fail(RL_ERR_COMM, "timed out waiting for the plan of the leaf-owner exchange (a rank of the job is missing from a collective?)");
                }
            }
            for (int r = 0; r < R; r++) { scount[r] = t->h_xmail[r]; sdispl[r] = t->h_xmail[R + r]; rcount[r] = t->h_xmail[2 * R + r]; rdispl[r] = t->h_xmail[3 * R + r]; }
        }
        rcd = t->dist->alltoallv(t->d_send, scount.data(), sdispl.data(), t->d_gx, rcount.data(), rdispl.data(), s);
        if (rcd) return rcd;
        hipLaunchKernelGGL(k_plan_global, dim3(1), dim3(64), 0, s, (const int32_t *)t->d_gls, R, t->lsstride, nseg, t->gchain, (const int32_t *)t->d_own, me);
        hipLaunchKernelGGL(k_chain_assemble, dim3(nseg, 2, kLeafXferZ), dim3(kThreads), 0, s, (const double *)t->d_gx, (const int32_t *)t->d_gls, R, t->lsstride, nseg, lx,
                           t->gchain, me, (const double *)lb.xs, lb.cap_n, (const int32_t *)c.leaf_start);
        ChainSource gsrc{t->gchain.xs, t->gchain.xs + t->gchain.cap_n, nullptr, nullptr, nullptr, nullptr};
        enqueue_chain(t, t->gchain, gsrc);
        if (R > 1) {       // every rank evaluated its own leaves: exchange the 2 L float sums
            rcd = t->dist->allgather(t->gchain.result, t->d_gres, (size_t)2 * MS * sizeof(float), s);
            if (rcd) return rcd;
            hipLaunchKernelGGL(k_chain_pick, dim3((2 * nseg + kThreads - 1) / kThreads), dim3(kThreads), 0, s, (const float *)t->d_gres, R, 2,
                               MS, nseg, (const int32_t *)t->d_own, t->gchain.result);
        }
        hipLaunchKernelGGL(k_leaf_output, dim3((c.L + kThreads - 1) / kThreads), dim3(kThreads), 0, s, c, t->gchain);
        }
    } else {   // K7: the two Java float running sums of every leaf, exactly, in parallel (rl_chain.inc)
        ChainSource src{nullptr, nullptr, c.lw, c.idx[0], c.idx[1], t->d_seg_buf, stream_scores ? c.leaf_of : nullptr};
        enqueue_chain(t, t->leaf_chain, src);
        hipLaunchKernelGGL(k_leaf_output, dim3((c.L + kThreads - 1) / kThreads), dim3(kThreads), 0, s, c, t->leaf_chain);
    }
    if (c.tie_on && defer_seen) {
        // deferred ties (plateaus of right children, several features over one cut): the tree was grown with the partition the tied candidates
        // share; the (feature, threshold) the Java's rounding noise would store is decided now, in one batch (the leaf sums above are already
        // enqueued and run meanwhile; the score update waits, because a tie that turns out to hide two different cuts restarts the tree)
        bool other_cut = false;
        int rcs = resolve_ties(t, fin_lds, nodes_in_lds, true, &other_cut);
        if (rcs) return rcs;
        if (other_cut && (c.tie_on & 2)) { c.tie_on = 1; t->tie_regrown++; goto regrow; }
    }
    if (stream_scores) hipLaunchKernelGGL(k_score_stream, dim3(std::max(1, std::min(2048, (c.N + kScoreBatch * kThreads - 1) / (kScoreBatch * kThreads)))), dim3(kThreads), 0, s, c);
    else hipLaunchKernelGGL(k_score_update, dim3(std::max(1, std::min(4096, (c.N + kScoreBatch * kThreads - 1) / (kScoreBatch * kThreads)))), dim3(kThreads), 0, s, c);
    hipLaunchKernelGGL(k_export_tree, dim3(1), dim3(kThreads), 0, s, c, t->ens, m);
    RL_HIP(hipGetLastError());
    // per-round training metric (LambdaMART.java:216)
    // (sharded runs, round 6: the per-query values are gathered on the main stream -- one communicator, one stream -- and the float mean over the
    // gathered lists runs on the side stream beside the next round's lambdas, as the one-GPU mean does)
    const bool use_side = !t->has_valid;
    if (use_side && t->side_pending) RL_HIP(hipStreamWaitEvent(s, t->ev_metric, 0));     // the previous round's metric still reads d_ndcg (sharded: the gathered copy)
    int rc = launch_rank(t, t->tr, c.scores, t->tr.d_ndcg, true);      // also the ranking of round m+1's lambdas
    if (rc != RL_OK) return rc;
    if (use_side) {
        const double *mq = t->tr.d_ndcg; int mQ = t->tr.Q;
        if (t->dist) { rc = gather_queries(t, t->tr.d_ndcg, &mq); if (rc != RL_OK) return rc; mQ = t->Qglobal; }
        RL_HIP(hipEventRecord(t->ev_ranked, s));
        RL_HIP(hipStreamWaitEvent(t->side, t->ev_ranked, 0));
        enqueue_metric_mean(t, mq, mQ, c.round_metric + 2 * (size_t)m, t->side);
        RL_HIP(hipEventRecord(t->ev_metric, t->side));
        t->side_pending = true;
    } else if (t->dist) {
        const double *gq = nullptr;
        rc = gather_queries(t, t->tr.d_ndcg, &gq);
        if (rc != RL_OK) return rc;
        enqueue_metric_mean(t, gq, t->Qglobal, c.round_metric + 2 * (size_t)m);
    } else enqueue_metric_mean(t, t->tr.d_ndcg, t->tr.Q, c.round_metric + 2 * (size_t)m);
    if (t->has_valid) {   // :228-237
        hipLaunchKernelGGL(k_valid_update, dim3(std::min<int64_t>(4096, (t->va.N + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                           t->ens, c.MAXN, m, (const float *)t->va.d_X, (int)t->va.N, t->F, c.lr, t->va.d_scores);
        rc = launch_rank(t, t->va, t->va.d_scores, t->va.d_ndcg, false);
        if (rc != RL_OK) return rc;
        if (t->dist) {      // every rank holds a shard of the validation lists: the float mean runs over all of them in list order
            const double *gq = nullptr;
            rc = gather_queries(t, t->va.d_ndcg, &gq, true);
            if (rc != RL_OK) return rc;
            enqueue_metric_mean(t, gq, t->vQglobal, c.round_metric + 2 * (size_t)m + 1);
        } else enqueue_metric_mean(t, t->va.d_ndcg, t->va.Q, c.round_metric + 2 * (size_t)m + 1);
    }
    RL_HIP(hipGetLastError());
    t->round = m + 1;
    t->n_kept = t->round;
    return RL_OK;
}

static int sync_rounds(rl_trainer *t)
{
    RL_HIP(hipStreamSynchronize(t->stream));
    RL_HIP(hipStreamSynchronize(t->side));
    collect_timing(t);
    TreeState st;
    RL_HIP(hipMemcpy(&st, t->ctx.st, sizeof(st), hipMemcpyDeviceToHost));
    if (st.error) return fail(RL_ERR_HIP, "device tree growth ran out of node slots (internal error)");
    if (t->round > t->synced_rounds) {
        t->h_metrics.resize((size_t)t->round * 2);
        RL_HIP(hipMemcpy(t->h_metrics.data() + 2 * (size_t)t->synced_rounds, t->ctx.round_metric + 2 * (size_t)t->synced_rounds,
                         (size_t)(t->round - t->synced_rounds) * 2 * sizeof(float), hipMemcpyDeviceToHost));
        t->synced_rounds = t->round;
    }
    return RL_OK;
}

static int fetch_tree(const rl_trainer *tc, int i, HostTree &out)
{
    rl_trainer *t = const_cast<rl_trainer *>(tc);
    if ((int)t->trees.size() <= i) t->trees.resize((size_t)i + 1);
    if (t->trees[i].n_nodes > 0) { out = t->trees[i]; return RL_OK; }
    const int MAXN = t->ctx.MAXN;
    int32_t nn = 0;
    RL_HIP(hipMemcpy(&nn, t->ens.n_nodes + i, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (nn <= 0 || nn > MAXN) return fail(RL_ERR_STATE, "tree " + std::to_string(i) + " has not been built");
    std::vector<int32_t> fi(nn), le(nn), ri(nn), cn(nn);
    std::vector<float> th(nn), ou(nn);
    std::vector<double> dv(nn);
    const size_t o = (size_t)i * MAXN;
    RL_HIP(hipMemcpy(fi.data(), t->ens.feat_idx + o, nn * sizeof(int32_t), hipMemcpyDeviceToHost));
    RL_HIP(hipMemcpy(le.data(), t->ens.left + o, nn * sizeof(int32_t), hipMemcpyDeviceToHost));
    RL_HIP(hipMemcpy(ri.data(), t->ens.right + o, nn * sizeof(int32_t), hipMemcpyDeviceToHost));
    RL_HIP(hipMemcpy(cn.data(), t->ens.count + o, nn * sizeof(int32_t), hipMemcpyDeviceToHost));
    RL_HIP(hipMemcpy(th.data(), t->ens.thr + o, nn * sizeof(float), hipMemcpyDeviceToHost));
    RL_HIP(hipMemcpy(ou.data(), t->ens.out + o, nn * sizeof(float), hipMemcpyDeviceToHost));
    RL_HIP(hipMemcpy(dv.data(), t->ens.deviance + o, nn * sizeof(double), hipMemcpyDeviceToHost));
    HostTree h;
    h.weight = t->p.learning_rate;
    // creation order -> pre-order (root, left subtree, right subtree)
    std::vector<int> stack{0};
    std::vector<int> order, newid(nn, -1);
    while (!stack.empty()) {
        const int x = stack.back(); stack.pop_back();
        newid[x] = (int)order.size(); order.push_back(x);
        if (fi[x] != -1) { stack.push_back(ri[x]); stack.push_back(le[x]); }
    }
    h.n_nodes = (int)order.size();
    for (int x : order) {
        const bool leaf = fi[x] == -1;
        h.feature.push_back(leaf ? -1 : t->feature_ids[fi[x]]);
        h.threshold.push_back(th[x]);
        h.left.push_back(leaf ? -1 : newid[le[x]]);
        h.right.push_back(leaf ? -1 : newid[ri[x]]);
        h.output.push_back(ou[x]);
        h.deviance.push_back(dv[x]);
        h.count.push_back(cn[x]);
    }
    t->trees[i] = h;
    out = h;
    return RL_OK;
}

__global__ void k_debug_root_sum(const Ctx c, double *out, long long *out_fixed)
{
    const int f = blockIdx.x;
    for (int t = threadIdx.x; t < c.nthr[f]; t += blockDim.x) {
        const size_t o = (size_t)f * c.TS + t;
        if (out) out[o] = fixed_to_double(make_i128(c.cum_hi[o], c.cum_lo[o]), c.st->E);
        if (out_fixed) { out_fixed[2 * o] = c.cum_hi[o]; out_fixed[2 * o + 1] = (long long)c.cum_lo[o]; }
    }
}

}  // namespace rl

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int rl_abi_version(void) { return RLHIP_ABI_VERSION; }
const char *rl_last_error(void) { return g_err.c_str(); }

int rl_device_count(int32_t *n)
{
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { if (n) *n = 0; return fail(RL_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    if (n) *n = c;
    return RL_OK;
}

void rl_params_default(rl_params *p)
{   // learning/tree/LambdaMART.java:37-42
    if (!p) return;
    p->n_trees = 1000; p->n_leaves = 10; p->n_threshold = 256; p->min_leaf_support = 1; p->early_stop_rounds = 100;
    p->learning_rate = 0.1F; p->metric = RL_METRIC_NDCG; p->metric_k = 10; p->device = 0; p->flags = 0;
    p->ranker = RL_RANKER_LAMBDAMART;
    p->feature_sampling_rate = 1.0f; p->seed = 0;
}

int rl_set_external_judgments(rl_trainer *t, int32_t validation, const double *ideal_dcg, const int32_t *rel_doc_count)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (t->inited) return fail(RL_ERR_STATE, "rl_set_external_judgments must be called before rl_init");
    if (validation ? !t->has_valid : !t->has_train) return fail(RL_ERR_STATE, "set the data first");
    DataSet &d = validation ? t->va : t->tr;
    d.ext_ideal.clear(); d.ext_rd.clear();
    if (ideal_dcg) d.ext_ideal.assign(ideal_dcg, ideal_dcg + d.Q);
    if (rel_doc_count) {
        for (int q = 0; q < d.Q; q++) if (rel_doc_count[q] < 0) return fail(RL_ERR_INVALID, "negative relevant-document count");
        d.ext_rd.assign(rel_doc_count, rel_doc_count + d.Q);
    }
    return RL_OK;
}

int rl_set_err_max(double max_gain)
{
    if (!(max_gain > 0.0) || !std::isfinite(max_gain)) return fail(RL_ERR_INVALID, "ERRScorer.MAX must be positive and finite");
    g_err_max = max_gain;
    return RL_OK;
}

int rl_create(const rl_params *p, rl_trainer **out)
{
    if (!p || !out) return fail(RL_ERR_INVALID, "null argument");
    *out = nullptr;
    if (p->metric < RL_METRIC_NDCG || p->metric > RL_METRIC_ERR)
        return fail(RL_ERR_UNSUPPORTED, "train metric must be NDCG, DCG, MAP or ERR (P / RR / BEST are not built for training)");
    if (p->metric == RL_METRIC_MAP ? p->metric_k < 0 : p->metric_k < 1) return fail(RL_ERR_UNSUPPORTED, "metric k out of range");
    if (p->ranker != RL_RANKER_LAMBDAMART && p->ranker != RL_RANKER_MART)
        return fail(RL_ERR_UNSUPPORTED, "ranker must be RL_RANKER_LAMBDAMART (6) or RL_RANKER_MART (0)");
    if (!(p->feature_sampling_rate >= 0.0f && p->feature_sampling_rate <= 1.0f)) return fail(RL_ERR_INVALID, "feature_sampling_rate must be in [0, 1]");
    if (p->n_trees < 1) return fail(RL_ERR_INVALID, "n_trees must be >= 1");
    if (p->n_leaves < 1 && p->n_leaves != -1) return fail(RL_ERR_INVALID, "n_leaves must be >= 1, or -1 for trees limited by min_leaf_support only");
    if (p->min_leaf_support < 1) return fail(RL_ERR_INVALID, "min_leaf_support must be >= 1");
    if (p->n_threshold != -1 && (p->n_threshold < 1 || p->n_threshold > (1 << 24)))
        return fail(RL_ERR_UNSUPPORTED, "n_threshold must be -1 or in [1, 2^24]");
    if (p->flags & ~(RL_FLAG_TIMING | RL_FLAG_TIMING_NODES | RL_FLAG_SERIAL_CHAIN | RL_FLAG_JAVA_ORDER | RL_FLAG_FIRST_TIE)) return fail(RL_ERR_INVALID, "unknown bit in rl_params.flags");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(RL_ERR_NO_DEVICE, "no HIP device visible: librlhip has no CPU fallback");
    if (p->device < 0 || p->device >= ndev) return fail(RL_ERR_INVALID, "device ordinal out of range");
    RL_HIP(hipSetDevice(p->device));
    hipDeviceProp_t prop;
    RL_HIP(hipGetDeviceProperties(&prop, p->device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(RL_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", librlhip is built for gfx950 only");
    std::unique_ptr<rl_trainer> t(new rl_trainer());
    t->p = *p;
    t->err_max = g_err_max;
    memset(&t->ctx, 0, sizeof(t->ctx));
    memset(&t->ens, 0, sizeof(t->ens));
    RL_HIP(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    RL_HIP(hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking));
    RL_HIP(hipEventCreateWithFlags(&t->ev_ranked, hipEventDisableTiming)); RL_HIP(hipEventCreateWithFlags(&t->ev_metric, hipEventDisableTiming));
    t->lam_streams = !(getenv("RLHIP_LAMBDA_STREAMS") && atoi(getenv("RLHIP_LAMBDA_STREAMS")) == 0);
    if (const char *e = getenv("RLHIP_LAMBDA_SIDE")) t->lam_side = std::max(0, std::min(3, atoi(e)));
    t->lam_compact = getenv("RLHIP_LAMBDA_COMPACT") && atoi(getenv("RLHIP_LAMBDA_COMPACT")) != 0;
    if (t->lam_streams) {
        RL_HIP(hipEventCreateWithFlags(&t->ev_lam_fork, hipEventDisableTiming));
        for (int i = 0; i < 3; i++) { RL_HIP(hipStreamCreateWithFlags(&t->lam_s[i], hipStreamNonBlocking)); RL_HIP(hipEventCreateWithFlags(&t->ev_lam_join[i], hipEventDisableTiming)); }
    }
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<true, 16, kHistLdsStride>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<false, 16, kHistLdsStride>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<true, 16, kHistLdsStride, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<false, 16, kHistLdsStride, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<true, 16, kHistLdsStride, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<false, 16, kHistLdsStride, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<true, 16, kHistLdsStride, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<false, 16, kHistLdsStride, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<true, 16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<false, 16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<true, 8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<false, 8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<true, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<false, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<true, 2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<false, 2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<true, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist<false, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistLdsBytes));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist_reduce, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxBins * 20));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist_sp<kHistLdsStride>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistFG * kHistLdsStride * 8));
    RL_HIP(hipFuncSetAttribute((const void *)k_select, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist_finish_wide, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist_finish<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist_finish<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist_finish<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist_finish<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist_finish<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    RL_HIP(hipFuncSetAttribute((const void *)k_hist_finish<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    RL_HIP(hipFuncSetAttribute((const void *)k_rank_block, hipFuncAttributeMaxDynamicSharedMemorySize, ((kLambdaBlockCap + 63) & ~63) * kRankLdsPerDoc));      // (max_big is rounded up to 64 documents)
    RL_HIP(hipFuncSetAttribute((const void *)k_rank_mixed, hipFuncAttributeMaxDynamicSharedMemorySize, std::max((kLambdaBlockCap + 63) & ~63, (kRankBlockThreads / 64) * kLambdaWaveCap) * kRankLdsPerDoc));
    RL_HIP(hipFuncSetAttribute((const void *)k_lambda_tiny, hipFuncAttributeMaxDynamicSharedMemorySize, kLambdaTinyGroups * lambda_tiny_group_bytes(kLambdaFusedMaxK)));
    RL_HIP(hipFuncSetAttribute((const void *)k_lambda_fused<256, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLambdaFusedMaxK * (256 + 8) * 16 + 2048));
    RL_HIP(hipFuncSetAttribute((const void *)k_lambda_fused<256, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLambdaFusedMaxK * (256 + 8) * 16 + 2048 + lambda_fused_cp_bytes(kLambdaFusedMaxK, 256)));
    RL_HIP(hipFuncSetAttribute((const void *)k_lambda_fused<256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLambdaFusedMaxK * (256 + 8) * 16 + 8192));
    RL_HIP(hipFuncSetAttribute((const void *)k_lambda_fused<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLambdaFusedMaxK * (256 + 8) * 16 + 8192));
    RL_HIP(hipFuncSetAttribute((const void *)k_chain_stitch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RL_HIP(hipFuncSetAttribute((const void *)k_tie_finish, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    *out = t.release();
    return RL_OK;
}

void rl_destroy(rl_trainer *t)
{
    if (!t) return;
    (void)hipSetDevice(t->p.device);
    if (getenv("RLHIP_TIE_PROF") && t->tie_stalls > 0)
        fprintf(stderr, "[rlhip] tie-break: %lld resolutions (%lld batches, %lld trees regrown), host us: first read %lld, chains %lld, candidates+lists %lld, sums %lld, finish %lld, total %lld\n",
                t->tie_stalls, t->tie_batches, t->tie_regrown, t->tie_phase_us[0], t->tie_phase_us[1], t->tie_phase_us[2], t->tie_phase_us[3], t->tie_phase_us[4], t->tie_us);
    if (getenv("RLHIP_CHAIN_PROF"))
        fprintf(stderr, "[rlhip] float chains: %lld watched evaluations with %lld repair passes, %lld blind ones with %lld; progress-word time-outs %lld; host waited %lld us for stitches\n",
                t->chain_calls[0], t->chain_repairs[0], t->chain_calls[1], t->chain_repairs[1], t->chain_timeouts, t->chain_wait_us);
    if (t->stream) { (void)hipStreamSynchronize(t->stream); }
    if (t->side) { (void)hipStreamSynchronize(t->side); (void)hipStreamDestroy(t->side); }
    if (t->ev_lam_fork) (void)hipEventDestroy(t->ev_lam_fork);
    for (int i = 0; i < 3; i++) { if (t->ev_lam_join[i]) (void)hipEventDestroy(t->ev_lam_join[i]); if (t->lam_s[i]) (void)hipStreamDestroy(t->lam_s[i]); }
    if (t->ev_ranked) (void)hipEventDestroy(t->ev_ranked);
    if (t->ev_metric) (void)hipEventDestroy(t->ev_metric);
    for (int w = 0; w < RL_KERNEL_COUNT_; w++)
        for (auto &pr : t->ev_pending[w]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto e : t->ev_free) (void)hipEventDestroy(e);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    if (t->h_progress) (void)hipHostFree(t->h_progress);
    if (t->h_xmail) (void)hipHostFree(t->h_xmail);
    if (t->tie_buf) (void)hipFree(t->tie_buf);
    if (t->tie_pin) (void)hipHostFree(t->tie_pin);
    for (void *q : t->pinned) (void)hipHostFree(q);
    delete t;
}

int rl_set_train(rl_trainer *t, const float *X, int64_t n_docs, int32_t n_features, const float *labels, const int32_t *qoff,
                 int32_t n_queries, const int32_t *feature_ids, const int32_t *qkey)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (t->has_train) return fail(RL_ERR_STATE, "training set already set");
    int rc = validate_dataset(X, n_docs, n_features, labels, qoff, n_queries);
    if (rc) return rc;
    RL_HIP(hipSetDevice(t->p.device));
    t->F = n_features;
    t->feature_ids.resize(n_features);
    for (int f = 0; f < n_features; f++) {
        t->feature_ids[f] = feature_ids ? feature_ids[f] : f + 1;
        if (t->feature_ids[f] <= 0) return fail(RL_ERR_INVALID, "Cannot use feature numbering less than or equal to zero. Start your features at 1.");
    }
    rc = load_dataset(t, t->tr, X, n_docs, labels, qoff, n_queries, qkey);
    if (rc) return rc;
    t->has_train = true;
    return RL_OK;
}

int rl_set_validation(rl_trainer *t, const float *X, int64_t n_docs, const float *labels, const int32_t *qoff, int32_t n_queries,
                      const int32_t *qkey)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!t->has_train) return fail(RL_ERR_STATE, "set the training set first");
    if (t->inited) return fail(RL_ERR_STATE, "validation set must be set before rl_init");
    if (t->has_valid) return fail(RL_ERR_STATE, "validation set already set");
    int rc = validate_dataset(X, n_docs, t->F, labels, qoff, n_queries);
    if (rc) return rc;
    RL_HIP(hipSetDevice(t->p.device));
    rc = load_dataset(t, t->va, X, n_docs, labels, qoff, n_queries, qkey);
    if (rc) return rc;
    t->has_valid = true;
    return RL_OK;
}

int rl_set_rows(rl_trainer *t, int32_t validation, int64_t first_doc, int64_t n_docs, const float *X)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (t->inited) return fail(RL_ERR_STATE, "rows must be delivered before rl_init");
    if (validation ? !t->has_valid : !t->has_train) return fail(RL_ERR_STATE, "rl_set_rows before rl_set_train / rl_set_validation");
    DataSet &d = validation ? t->va : t->tr;
    if (d.rows_next < 0) return fail(RL_ERR_STATE, "the rows of this data set were already given to rl_set_train / rl_set_validation");
    if (!X || n_docs <= 0) return fail(RL_ERR_INVALID, "bad argument");
    if (first_doc != d.rows_next || first_doc + n_docs > d.N) return fail(RL_ERR_INVALID, "row blocks must be consecutive and stay inside the data set");
    RL_HIP(hipSetDevice(t->p.device));
    RL_HIP(hipMemcpy(d.d_X + (size_t)first_doc * t->F, X, (size_t)n_docs * t->F * sizeof(float), hipMemcpyHostToDevice));
    d.rows_next += n_docs;
    return RL_OK;
}

int rl_init(rl_trainer *t)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!t->has_train) return fail(RL_ERR_STATE, "no training set");
    if (t->inited) return fail(RL_ERR_STATE, "rl_init called twice");
    if ((t->tr.rows_next >= 0 && t->tr.rows_next != t->tr.N) || (t->has_valid && t->va.rows_next >= 0 && t->va.rows_next != t->va.N))
        return fail(RL_ERR_STATE, "rl_set_rows has not delivered every row yet");
    RL_HIP(hipSetDevice(t->p.device));
    RL_HIP(hipDeviceSynchronize());      // uploads of rl_set_* went through the null stream; t->stream is non-blocking
    Ctx &c = t->ctx;
    hipStream_t s = t->stream;
    const int N = (int)t->tr.N;
    int F = t->F;                        // columns of the row matrix until the threshold tables are built, histogram features after (rl_init: virtual features)
    const int Npad = (N + 127) / 128 * 128;
    // -leaf -1 (RegressionTree.java:72 `nodes == -1`): growth ends when no leaf can be split any more.  Every leaf holds at least
    // min_leaf_support documents, so a tree has at most floor(N / mls) leaves -- and with exactly that many none is left with 2 * mls
    // documents: a budget of floor(N / mls) leaves never binds before the Java's own loop ends.
    int L_eff = t->p.n_leaves;
    if (L_eff == -1) {
        long long Nall = N;          // sharded: the budget comes from the GLOBAL document count (every rank grows the same tree)
        if (t->dist) {
            long long *d_n = nullptr;
            RL_HIP(t->pool.alloc(&d_n, (size_t)1));
            RL_HIP(hipMemcpy(d_n, &Nall, sizeof(Nall), hipMemcpyHostToDevice));
            int rcd = t->dist->allreduce(d_n, 1, DT_I64, OP_SUM, s); if (rcd) return rcd;
            RL_HIP(hipStreamSynchronize(s));
            RL_HIP(hipMemcpy(&Nall, d_n, sizeof(Nall), hipMemcpyDeviceToHost));
            t->pool.release(d_n);
        }
        L_eff = (int)std::max<long long>(1, std::min<long long>(Nall / std::max(1, t->p.min_leaf_support), 1 << 28));
    }
    t->L_eff = L_eff;
    c.N = N; c.Npad = Npad; c.Q = t->tr.Q; c.F = F; c.L = L_eff;
    // the root is split unconditionally before the leaf budget is looked at (RegressionTree.java:62-67): even -leaf 1 gives 3 nodes
    c.MAXN = std::max(2 * L_eff - 1, 3);
    c.NC = 4 * L_eff + 2;     // node records: committed (2L-1) + prepared but never reached (see select_step)
    c.mls = t->p.min_leaf_support; c.lr = t->p.learning_rate;
    // chunks a child node is cut into: every chunk flushes a partial histogram of F x T x 12 bytes that the finish reads back, so wide data wants fewer
    // (measured, rounds/s: c3, 700 columns: 3 / 4 / 6 / 8 / 12 / 16 / 24 -> 542 / 548 / 548 / 549-556 / 534 / 537 / 513; c2, 136 columns: flat from 8 to 32)
    c.node_div = std::max(4, std::min(24, (int)(24.0 * 136.0 / (double)std::max(F, 1) + 0.5)));      // (round 5, k_fin2: 16 partials per thread in one batch -- c2 sustained 345 -> 351 from 12 to 24 chunks; wide data keeps few)
    c.node_min = 256;      // smallest chunk of a child node.  Every round-5 number was measured at 256 (the assignment of kMinChunk had been swallowed by a comment; 256 is what ran)
    c.fs_size = F; c.fs_on = 0; c.seed = t->p.seed;
    if (t->p.feature_sampling_rate > 0.0f && t->p.feature_sampling_rate < 1.0f) { c.fs_size = (int32_t)(t->p.feature_sampling_rate * (float)F); c.fs_on = c.fs_size < F ? 1 : 0; }   // :274
    c.hist_nt = kThreads; c.sub_child = 16;
    if (const char *e = getenv("RLHIP_SUB_CHILD")) { const int v = atoi(e); if (v == 4 || v == 8) c.sub_child = v; }
    if (const char *e = getenv("RLHIP_HIST_NT")) c.hist_nt = atoi(e);
    if (const char *e = getenv("RLHIP_NODE_DIV")) c.node_div = std::max(1, atoi(e));          // tuning knobs (tools/), not API
    if (const char *e = getenv("RLHIP_NODE_MIN")) c.node_min = std::max(256, atoi(e) & ~255);
    // largest chunk: smaller ones spread a mid-sized node over more blocks (measured, same box: c1, 1.2 M x 136: 4096 -> +0.9 % / +1.6 % sustained; c2, 3.77 M: -0.7 %;
    // c3, 700 columns: -1.8 %: every chunk more is another partial histogram of F x T x 12 bytes)
    c.node_chunk = (N <= (2 << 20) && F <= 256) ? 4096 : kNodeChunk;
    if (const char *e = getenv("RLHIP_NODE_CHUNK")) c.node_chunk = std::min(kNodeChunk, std::max(1024, atoi(e) & ~255));
    // balanced chunks for the steps that fill the chip (balance_slots): rows of the child-pass grid = chunks per round of blocks, largest chunk, steps of at most balance_min chunks keep chunk_docs' rule
    // Rows: about two blocks per CU -- 512 / (feature-group blocks per chunk), to the nearest multiple of 8 (k_hist's XCD map), at least 16.
    // Measured, same box, rounds/s: c2 / c1 (9 groups, 56 rows) 362.6 -> 372.7, 316.5 -> 326.0 over 300 rounds, c2ns 346.5 -> 363.9, c1 632 -> 646 (80 rows: +1 %, 64 / 72: worse
    // than none, 48: +0.5 %); c3 (44 groups) 486 -> 522 with 16 rows (8 rows: 477, 24 rows: 494).
    {
        const int gxb = (F + kHistFG - 1) / kHistFG;
        c.balance = 1; c.balance_cap = kChunk;
        c.balance_target = std::max(16, ((512 / std::max(gxb, 1) + 4) / 8) * 8);
        c.balance_min = c.balance_target / 2;
    }
    if (const char *e = getenv("RLHIP_BALANCE")) c.balance = atoi(e) != 0;
    c.skip_last = 1;
    if (const char *e = getenv("RLHIP_SKIP_LAST")) c.skip_last = atoi(e) != 0;
    t->step2 = !(getenv("RLHIP_STEP2") && atoi(getenv("RLHIP_STEP2")) == 0);
    t->sel2_wide = !(getenv("RLHIP_SELECT2_WIDE") && atoi(getenv("RLHIP_SELECT2_WIDE")) == 0);
    if (const char *e = getenv("RLHIP_BALANCE_CAP")) c.balance_cap = std::min(kChunk, std::max(1024, atoi(e) & ~255));
    if (const char *e = getenv("RLHIP_BALANCE_TARGET")) c.balance_target = std::max(8, atoi(e) & ~7);
    if (const char *e = getenv("RLHIP_BALANCE_MIN")) c.balance_min = std::max(1, atoi(e));
    c.metric = t->p.metric; c.mart = (t->p.ranker == RL_RANKER_MART) ? 1 : 0;
    // lazy Java-order tie-break (rl_tie.inc): the default path's exact ties resolved as the Java's summation order resolves them.  Not with
    // feature sampling (the Java's draw is unseeded: nothing to match), not sharded (the Java's order is ONE sequence over all documents), not in
    // the strict mode (every candidate already comes from the Java-order histogram)
    c.tie_on = (c.fs_size == F && !(t->p.flags & (RL_FLAG_JAVA_ORDER | RL_FLAG_FIRST_TIE)) && !getenv("RLHIP_TIE_OFF")) ? 1 : 0;      // (sharded runs: decided below, once TS is known)
    // rows of a ranked list whose pairs the lambda loop visits (LambdaMART.java:375-377: j <= cutoff or k <= cutoff); for
    // NDCG / DCG / ERR row `cutoff` itself only holds zero swap changes
    c.k = (t->p.metric == RL_METRIC_MAP) ? t->p.metric_k + 1 : t->p.metric_k;
    c.rank = t->rank; c.n_ranks = t->n_ranks; c.sharded = t->dist ? 1 : 0;
    if (c.sharded) c.skip_last = 0;       // (the last split's lambda^2 sums travel with its histogram's all-reduce)

    // ---- K9: thresholds + bins on the device ----------------------------------------------------
    float *Xt = nullptr;
    RL_HIP(t->pool.alloc(&Xt, (size_t)F * Npad));
    hipLaunchKernelGGL(k_transpose, dim3((N + 31) / 32, (F + 31) / 32), dim3(kThreads), 0, s, (const float *)t->tr.d_X, Xt, N, F, Npad);
    const int nT = t->p.n_threshold;
    FeatStats fs;
    fs.limit = (nT == -1 || nT > kMaxBins - 1) ? kMaxBins - 1 : nT;        // the device's distinct-value sets hold up to 4 095 values; larger tables are built on the host (below)
    fs.HS = next_pow2(2 * (fs.limit + 2));
    RL_HIP(t->pool.alloc(&fs.minkey, (size_t)F)); RL_HIP(t->pool.alloc(&fs.maxkey, (size_t)F));
    RL_HIP(t->pool.alloc(&fs.set, (size_t)F * fs.HS)); RL_HIP(t->pool.alloc(&fs.nset, (size_t)F));
    RL_HIP(t->pool.alloc(&fs.overflow, (size_t)F)); RL_HIP(t->pool.alloc(&fs.bad, (size_t)1));
    RL_HIP(hipMemsetAsync(fs.minkey, 0xFF, F * sizeof(uint32_t), s));
    RL_HIP(hipMemsetAsync(fs.maxkey, 0, F * sizeof(uint32_t), s));
    RL_HIP(hipMemsetAsync(fs.set, 0, (size_t)F * fs.HS * sizeof(uint32_t), s));
    RL_HIP(hipMemsetAsync(fs.nset, 0, F * sizeof(int32_t), s));
    RL_HIP(hipMemsetAsync(fs.overflow, 0, F * sizeof(int32_t), s));
    RL_HIP(hipMemsetAsync(fs.bad, 0, sizeof(int32_t), s));
    const int slices = std::max(1, std::min(64, N / 4096));
    hipLaunchKernelGGL(k_feat_stats, dim3(F, slices), dim3(kThreads), fs.HS * sizeof(uint32_t), s, (const float *)Xt, N, Npad, fs);
    RL_HIP(hipGetLastError());
    if (t->dist) {      // global min / max / distinct sets: every rank must build the same threshold table
        int rcd = t->dist->allreduce(fs.minkey, F, DT_U32, OP_MIN, s); if (rcd) return rcd;
        rcd = t->dist->allreduce(fs.maxkey, F, DT_U32, OP_MAX, s); if (rcd) return rcd;
        rcd = t->dist->allreduce(fs.overflow, F, DT_I32, OP_MAX, s); if (rcd) return rcd;
        rcd = t->dist->allreduce(fs.bad, 1, DT_I32, OP_MAX, s); if (rcd) return rcd;
        uint32_t *gsets = nullptr;
        RL_HIP(t->pool.alloc(&gsets, (size_t)t->n_ranks * F * fs.HS));
        rcd = t->dist->allgather(fs.set, gsets, (size_t)F * fs.HS * sizeof(uint32_t), s); if (rcd) return rcd;
        RL_HIP(hipMemsetAsync(fs.set, 0, (size_t)F * fs.HS * sizeof(uint32_t), s));
        RL_HIP(hipMemsetAsync(fs.nset, 0, F * sizeof(int32_t), s));
        hipLaunchKernelGGL(k_merge_sets, dim3(F, t->n_ranks), dim3(kThreads), 0, s, (const uint32_t *)gsets, t->n_ranks, fs);
        RL_HIP(hipGetLastError());
        RL_HIP(hipStreamSynchronize(s));
        t->pool.release(gsets);
    }
    std::vector<int32_t> h_over(F);
    int32_t h_bad = 0;
    RL_HIP(hipStreamSynchronize(s));
    RL_HIP(hipMemcpy(h_over.data(), fs.overflow, F * sizeof(int32_t), hipMemcpyDeviceToHost));
    RL_HIP(hipMemcpy(&h_bad, fs.bad, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (h_bad) return fail(RL_ERR_INVALID, "NaN feature value (resolve NaN to 0 as DataPoint.getFeatureValue does)");
    if (nT == 1) {
        // -tc 1 on a column that overflows into the step table [fmin, MAX_VALUE]: a +Infinity value is above every threshold, and the Java's binning loop
        // (FeatureHistogram.java:88-107) then never assigns it -- stMap stays 0 and the counts exclude it, i.e. node counts that do not add up.  Not
        // reproduced (the binning here would put it into the last bin and count it): refused, with the reason
        std::vector<uint32_t> h_max(F);
        RL_HIP(hipMemcpy(h_max.data(), fs.maxkey, F * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int f = 0; f < F; f++)
            if (h_over[f] && h_max[f] == 0xFF800000u)      // float_key(+Infinity)
                return fail(RL_ERR_UNSUPPORTED, "-tc 1 with a +Infinity value in feature column " + std::to_string(f) + ": RankLib leaves such documents out of the histogram counts (unsupported)");
    }
    const bool want_big = (nT == -1 || nT > kMaxBins - 1);         // tables of more than 4 095 entries are possible
    const int TS0 = fs.limit + 1;
    float *thr0 = nullptr; int32_t *d_nthr = nullptr;
    RL_HIP(t->pool.alloc(&thr0, (size_t)F * TS0)); RL_HIP(t->pool.alloc(&d_nthr, (size_t)F));
    hipLaunchKernelGGL(k_thresholds, dim3(F), dim3(kThreads), fs.HS * sizeof(uint32_t), s, fs, want_big ? fs.limit : nT, TS0, thr0, d_nthr);
    RL_HIP(hipGetLastError());
    std::vector<int32_t> h_nthr(F);
    RL_HIP(hipStreamSynchronize(s));
    RL_HIP(hipMemcpy(h_nthr.data(), d_nthr, F * sizeof(int32_t), hipMemcpyDeviceToHost));
    // ---- threshold tables of more than 4 095 entries (-tc -1 on a column with that many distinct values, or -tc N > 4095: learning/tree/
    // LambdaMART.java:135-149 has no limit).  The histogram kernels keep a feature's bins in LDS, so such a REAL feature becomes several VIRTUAL
    // features, one per run of 4 094 consecutive thresholds: virtual feature r has the table [thr[r W], .., thr[r W + W - 1], MAX_VALUE] over the same
    // column.  "Smallest t with value <= table[t]" then clamps a document's real bin into the run -- documents below it join the run's first bin,
    // documents above it the MAX_VALUE bin -- so the run's cumulative histogram is the real feature's cumulative histogram on its thresholds, every
    // real candidate is a candidate of exactly one virtual feature, in the Java's scan order, and the MAX_VALUE bin of a run that is not the last
    // never splits (nothing on its right).  Everything after this block sees F = the number of virtual features; trees are exported with the real
    // column (Ctx::vcol).  The large tables are built on the host (sort + unique of the column).
    std::vector<std::vector<float>> big((size_t)F);
    bool any_big = false;
    if (want_big) {
        std::vector<float> colv((size_t)N);
        for (int f = 0; f < F; f++) {
            if (!h_over[f]) continue;
            RL_HIP(hipMemcpy(colv.data(), Xt + (size_t)f * Npad, (size_t)N * sizeof(float), hipMemcpyDeviceToHost));
            float fmax = -std::numeric_limits<float>::infinity(), fmin = 3.4028234663852886e38f;       // :114-115
            for (auto &v : colv) { if (v == 0.f) v = 0.f; if (fmax < v) fmax = v; if (fmin > v) fmin = v; }      // (-0.0 folded as on the device)
            std::vector<float> vals(colv);
            std::sort(vals.begin(), vals.end());
            vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
            if (t->dist) {
                // sharded: the table is built from the distinct values of the column over ALL ranks (LambdaMART.java:108-150 walks one sorted list of every
                // sample).  Every rank contributes its own sorted distinct values -- counts first, then the values padded to the largest count -- and merges
                // what it receives; the overflow flags were all-reduced above, so the ranks take this branch for the same columns in the same order.
                const int R = t->n_ranks;
                // (temporaries are released on every path -- DevTmp -- and nothing returns between the two collectives of a column for a LOCAL
                // reason: a copy that fails is reported after the second all-gather, so the peers are not left waiting in it)
                struct DevTmp { void *p = nullptr; ~DevTmp() { if (p) (void)hipFree(p); } hipError_t get(size_t bytes) { return hipMalloc(&p, bytes); } };
                DevTmp t_cnt, t_cnts, t_v, t_all;
                hipError_t herr = t_cnt.get(sizeof(int32_t));
                if (herr == hipSuccess) herr = t_cnts.get((size_t)R * sizeof(int32_t));
                if (herr != hipSuccess) return fail(RL_ERR_HIP, std::string("hipMalloc (threshold-table merge): ") + hipGetErrorString(herr));
                const int32_t mycnt = (int32_t)vals.size();
                herr = hipMemcpy(t_cnt.p, &mycnt, sizeof(int32_t), hipMemcpyHostToDevice);
                int rcd = t->dist->allgather(t_cnt.p, t_cnts.p, sizeof(int32_t), s);
                if (rcd) return rcd;
                std::vector<int32_t> cnts((size_t)R);
                if (herr == hipSuccess) herr = hipStreamSynchronize(s);
                if (herr == hipSuccess) herr = hipMemcpy(cnts.data(), t_cnts.p, (size_t)R * sizeof(int32_t), hipMemcpyDeviceToHost);
                if (herr != hipSuccess) return fail(RL_ERR_HIP, std::string("threshold-table merge (counts): ") + hipGetErrorString(herr));
                const size_t mx = std::max<size_t>((size_t)*std::max_element(cnts.begin(), cnts.end()), 1);
                herr = t_v.get(mx * sizeof(float));
                if (herr == hipSuccess) herr = t_all.get(mx * R * sizeof(float));
                if (herr != hipSuccess) return fail(RL_ERR_HIP, std::string("hipMalloc (threshold-table merge): ") + hipGetErrorString(herr));
                herr = hipMemset(t_v.p, 0, mx * sizeof(float));
                if (herr == hipSuccess && !vals.empty()) herr = hipMemcpy(t_v.p, vals.data(), vals.size() * sizeof(float), hipMemcpyHostToDevice);
                rcd = t->dist->allgather(t_v.p, t_all.p, mx * sizeof(float), s);
                if (rcd) return rcd;
                std::vector<float> all(mx * R);
                if (herr == hipSuccess) herr = hipStreamSynchronize(s);
                if (herr == hipSuccess) herr = hipMemcpy(all.data(), t_all.p, all.size() * sizeof(float), hipMemcpyDeviceToHost);
                if (herr != hipSuccess) return fail(RL_ERR_HIP, std::string("threshold-table merge (values): ") + hipGetErrorString(herr));
                vals.clear();
                for (int r = 0; r < R; r++) vals.insert(vals.end(), all.begin() + (size_t)r * mx, all.begin() + (size_t)r * mx + cnts[r]);
                std::sort(vals.begin(), vals.end());
                vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
                if (!vals.empty()) { fmin = vals.front(); fmax = vals.back(); }
            }
            std::vector<float> &tab = big[f];
            if (nT == -1 || (long long)vals.size() <= (long long)nT) { tab = vals; tab.push_back(3.4028234663852886e38f); }      // :135-140
            else {                                                                                                              // :141-149
                const float step = fabsf(fmax - fmin) / (float)nT;
                tab.resize((size_t)nT + 1);
                tab[0] = fmin;
                for (int j = 1; j < nT; j++) tab[j] = tab[j - 1] + step;
                tab[nT] = 3.4028234663852886e38f;
            }
            any_big = true;
        }
    }
    float *d_thr = nullptr;
    int TS = 2;
    t->vcol.clear();
    c.vcol = nullptr;
    if (!any_big) {
        for (int f = 0; f < F; f++) TS = std::max(TS, h_nthr[f]);
        c.TS = TS;
        RL_HIP(t->pool.alloc(&d_thr, (size_t)F * TS));
        RL_HIP(hipMemsetAsync(d_thr, 0, (size_t)F * TS * sizeof(float), s));
        RL_HIP(hipMemcpy2DAsync(d_thr, TS * sizeof(float), thr0, TS0 * sizeof(float), TS * sizeof(float), F, hipMemcpyDeviceToDevice, s));
    } else {
        if (t->p.flags & RL_FLAG_JAVA_ORDER) return fail(RL_ERR_UNSUPPORTED, "a threshold table of more than 4095 entries with RL_FLAG_JAVA_ORDER (the Java's prefix over ALL bins of a feature is one f64 chain)");
        std::vector<float> h_thr0((size_t)F * TS0);
        RL_HIP(hipMemcpy(h_thr0.data(), thr0, h_thr0.size() * sizeof(float), hipMemcpyDeviceToHost));
        constexpr int W = kMaxBins - 2;
        std::vector<std::vector<float>> rows;
        for (int f = 0; f < F; f++) {
            const float *tab = big[f].empty() ? h_thr0.data() + (size_t)f * TS0 : big[f].data();
            const long long T = big[f].empty() ? h_nthr[f] : (long long)big[f].size();
            if (T <= kMaxBins - 1) { rows.emplace_back(tab, tab + T); t->vcol.push_back(f); continue; }
            for (long long r0 = 0; r0 < T; r0 += W) {
                const long long e1 = std::min<long long>(r0 + W, T);
                std::vector<float> row(tab + r0, tab + e1);
                if (e1 < T) row.push_back(3.4028234663852886e38f);
                rows.push_back(std::move(row)); t->vcol.push_back(f);
            }
        }
        if (rows.size() > (size_t)(1 << 20)) return fail(RL_ERR_UNSUPPORTED, "threshold tables of more than 2^32 entries in total");
        F = (int)rows.size();
        for (auto &r : rows) TS = std::max(TS, (int)r.size());
        c.TS = TS;
        std::vector<float> h_thr((size_t)F * TS, 0.f);
        h_nthr.assign((size_t)F, 0);
        for (int v = 0; v < F; v++) { memcpy(h_thr.data() + (size_t)v * TS, rows[v].data(), rows[v].size() * sizeof(float)); h_nthr[v] = (int32_t)rows[v].size(); }
        t->pool.release(d_nthr); d_nthr = nullptr;
        RL_HIP(t->pool.alloc(&d_thr, (size_t)F * TS)); RL_HIP(t->pool.alloc(&d_nthr, (size_t)F));
        RL_HIP(hipMemcpy(d_thr, h_thr.data(), h_thr.size() * sizeof(float), hipMemcpyHostToDevice));
        RL_HIP(hipMemcpy(d_nthr, h_nthr.data(), (size_t)F * sizeof(int32_t), hipMemcpyHostToDevice));
        int32_t *d_vcol = nullptr;
        RL_HIP(t->pool.alloc(&d_vcol, (size_t)F));
        RL_HIP(hipMemcpy(d_vcol, t->vcol.data(), (size_t)F * sizeof(int32_t), hipMemcpyHostToDevice));
        c.vcol = d_vcol;
        c.F = F; if (!c.fs_on) c.fs_size = F;          // (with feature sampling fs_size stays a number of REAL features: the draw is over columns)
        c.node_div = std::max(4, std::min(24, (int)(24.0 * 136.0 / (double)std::max(F, 1) + 0.5)));      // (round 5, k_fin2: 16 partials per thread in one batch -- c2 sustained 345 -> 351 from 12 to 24 chunks; wide data keeps few)
        if (const char *e = getenv("RLHIP_NODE_DIV")) c.node_div = std::max(1, atoi(e));
        // exact ties: the first candidate in the Java's scan order.  The lazy re-decision needs the Java's own f64 prefix over ALL bins of a real feature,
        // which a run of a split table does not hold (its first bin is a merged sum)
        c.tie_on = 0;
    }
    c.thr = d_thr; c.nthr = d_nthr;
    c.live = nullptr; c.live_nthr = nullptr; c.n_live = F;
    if (!t->dist) {       // features that can split at all (> 1 distinct value <=> more than the value + Float.MAX_VALUE thresholds)
        std::vector<int32_t> live, live_n;
        for (int f = 0; f < F; f++) if (h_nthr[f] > 2) { live.push_back(f); live_n.push_back(h_nthr[f]); }
        if (live.empty()) { live.push_back(0); live_n.push_back(h_nthr[0]); }
        int32_t *d_live = nullptr, *d_live_n = nullptr;
        RL_HIP(t->pool.alloc(&d_live, live.size())); RL_HIP(t->pool.alloc(&d_live_n, live.size()));
        RL_HIP(hipMemcpy(d_live, live.data(), live.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        RL_HIP(hipMemcpy(d_live_n, live_n.data(), live_n.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        c.live = d_live; c.live_nthr = d_live_n; c.n_live = (int32_t)live.size();
    }
    {   // more finish blocks a step than the fused kernel keeps resident (5 a CU): the per-feature work and the bookkeeping become two launches
        const char *e = getenv("RLHIP_FIN_SPLIT");
        int n_cu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, t->p.device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
        t->fin_split = !t->dist && !(t->p.flags & RL_FLAG_JAVA_ORDER) && (e ? atoi(e) != 0 : (long long)c.n_live * kSpec > 5ll * n_cu);
    }
    if ((size_t)TS * 12 > (size_t)kHistLdsBytes) return fail(RL_ERR_UNSUPPORTED, "too many threshold candidates for the LDS histogram");
    // sharded runs evaluate the tie-break on gathered arrays through the contiguous-chain path only: with threshold tables too large for its sort
    // (the literal walk reads one rank's documents) they keep the first candidate.  TS is the same on every rank, so is the decision.
    if (t->dist && t->n_ranks > 1 && (size_t)kTsWaves * TS * 4 > (size_t)60 * 1024) c.tie_on = 0;
    // bit 1: ties over several features that all cut a node the same way are deferred to the end of the tree like plateau ties (the check that it
    // IS one cut reads the node's documents, rl_tie.inc k_tie_verify: every rank its own, the verdict is all-reduced)
    if (c.tie_on && !getenv("RLHIP_TIE_NO_XDEFER")) c.tie_on |= 2;
    if (t->p.n_leaves == -1) {      // -leaf -1: the node histograms are sized for floor(N / mls) leaves -- say so before an allocation fails
        const double need = (double)c.NC * F * TS * ((t->p.flags & RL_FLAG_JAVA_ORDER) ? 28.0 : 20.0);
        size_t mem_free = 0, mem_total = 0;
        RL_HIP(hipMemGetInfo(&mem_free, &mem_total));
        if (need > 0.8 * (double)mem_free) return fail(RL_ERR_UNSUPPORTED, "-leaf -1: up to " + std::to_string(L_eff) + " leaves would need " + std::to_string((long long)(need / 1e9)) +
                                                                          " GB of node histograms; raise -mls or set -leaf");
    }
    // features of a 16-feature group handled by one histogram block: all 16 when the LDS budget allows
    c.FG = kHistFG;
    c.numFG = (F + kHistFG - 1) / kHistFG;
    c.sub = 16;
    while (c.sub > 1 && (size_t)c.sub * TS * 12 > (size_t)kHistLdsBytes) c.sub >>= 1;

    uint16_t *d_bins = nullptr, *d_gbins = nullptr;
    RL_HIP(t->pool.alloc(&d_bins, (size_t)F * Npad));
    RL_HIP(hipMemsetAsync(d_bins, 0, (size_t)F * Npad * sizeof(uint16_t), s));
    RL_HIP(t->pool.alloc(&d_gbins, (size_t)c.numFG * Npad * kHistFG));
    RL_HIP(hipMemsetAsync(d_gbins, 0, (size_t)c.numFG * Npad * kHistFG * sizeof(uint16_t), s));
    c.bins = d_bins; c.gbins = d_gbins;
    // chunks of one growth step (all slots; see prepare_children), and of the root pass
    c.maxChunks = N / c.node_chunk + 67 * kSpec + 2;
    c.nTiles = (N + kPartTile - 1) / kPartTile + kSpec;      // tiles of one growth step (disjoint nodes, one ragged tile each)
    RL_HIP(t->pool.alloc(&c.cum_hi, (size_t)c.NC * F * TS));
    RL_HIP(t->pool.alloc(&c.cum_lo, (size_t)c.NC * F * TS));
    RL_HIP(t->pool.alloc(&c.cum_cnt, (size_t)c.NC * F * TS));
    RL_HIP(hipMemsetAsync(c.cum_cnt, 0, (size_t)F * TS * sizeof(int32_t), s));
    c.java = (t->p.flags & RL_FLAG_JAVA_ORDER) ? 1 : 0;
    if (c.java) {
        if (t->dist) return fail(RL_ERR_UNSUPPORTED, "RL_FLAG_JAVA_ORDER with multi-GPU training: the Java's summation order is a single sequence over all documents");
        RL_HIP(t->pool.alloc(&c.jl, (size_t)Npad)); RL_HIP(t->pool.alloc(&c.jb, (size_t)F * Npad));
        RL_HIP(t->pool.alloc(&c.jbin, (size_t)kSpec * F * TS)); RL_HIP(t->pool.alloc(&c.jtot, (size_t)kSpec * 2));
        RL_HIP(t->pool.alloc(&c.jcum, (size_t)c.NC * F * TS));
        RL_HIP(hipMemsetAsync(c.jcum, 0, (size_t)c.NC * F * TS * sizeof(double), s));
        RL_HIP(hipMemsetAsync(c.jbin, 0, (size_t)kSpec * F * TS * sizeof(double), s));
    }
    hipLaunchKernelGGL(k_binning, dim3(F, slices), dim3(kThreads), (size_t)TS * 8, s, (const float *)Xt, N, Npad, TS, (const float *)d_thr,
                       (const int32_t *)d_nthr, d_bins, d_gbins, c.cum_cnt, c.vcol);
    c.cum_cnt_loc = nullptr;
    if (t->dist && !getenv("RLHIP_DIST_COUNT_PASS")) {       // this rank's own root counts, kept beside the all-reduced ones (RLHIP_DIST_COUNT_PASS=1: round 5's count pass + two-pass partition)
        RL_HIP(t->pool.alloc(&c.cum_cnt_loc, (size_t)c.NC * F * TS));
        RL_HIP(hipMemsetAsync(c.cum_cnt_loc, 0, (size_t)c.NC * F * TS * sizeof(int32_t), s));
        RL_HIP(hipMemcpyAsync(c.cum_cnt_loc, c.cum_cnt, (size_t)F * TS * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    }
    if (t->dist) { int rcd = t->dist->allreduce(c.cum_cnt, (size_t)F * TS, DT_I32, OP_SUM, s); if (rcd) return rcd; }
    {
        uint16_t *d_dbins = nullptr;
        c.dm_gstride = getenv("RLHIP_DM_NOALIGN") ? c.numFG : (c.numFG + 3) & ~3;
        RL_HIP(t->pool.alloc(&d_dbins, (size_t)c.dm_gstride * Npad * kHistFG));
        RL_HIP(hipMemsetAsync(d_dbins, 0, (size_t)c.dm_gstride * Npad * kHistFG * sizeof(uint16_t), s));
        hipLaunchKernelGGL(k_docmajor, dim3(4096), dim3(kThreads), 0, s, (const uint16_t *)d_gbins, d_dbins, Npad, c.numFG, c.dm_gstride);
        c.dbins = d_dbins;
        // packed rows: one byte per bin + a mask for bin 256 (possible when no table has more than 257 entries; RLHIP_P8=0 keeps the 16-bit rows)
        c.p8 = 0;
        const int p8_mode = getenv("RLHIP_P8") ? atoi(getenv("RLHIP_P8")) : 1;        // 0 = 16-bit rows everywhere, 1 = packed rows for the root pass (default), 2 = also for child passes
        if (TS <= 257 && c.sub == 16 && p8_mode > 0) {
            unsigned char *d_pb = nullptr, *d_pd = nullptr; uint16_t *d_ph = nullptr;
            c.pd_stride = (c.numFG * 18 + 63) & ~63;
            RL_HIP(t->pool.alloc(&d_pb, (size_t)c.numFG * Npad * 16)); RL_HIP(t->pool.alloc(&d_ph, (size_t)c.numFG * Npad));
            if (p8_mode > 1) {
                RL_HIP(t->pool.alloc(&d_pd, (size_t)Npad * c.pd_stride));
                RL_HIP(hipMemsetAsync(d_pd, 0, (size_t)Npad * c.pd_stride, s));
            }
            hipLaunchKernelGGL(k_pack_rows, dim3(4096), dim3(kThreads), 0, s, (const uint16_t *)d_gbins, d_pb, d_ph, d_pd, Npad, c.numFG, c.pd_stride);
            c.pbins = d_pb; c.phib = d_ph; c.pdbins = d_pd; c.p8 = p8_mode > 1 ? 2 : 1;
        }
        c.dm_root = 0; c.dm_div = 1;      // measured at c2 (profiles/r02d_dm_sweep.txt): every child pass gains, the root pass loses
        if (const char *e = getenv("RLHIP_DM_ROOT")) c.dm_root = atoi(e) ? 1 : 0;      // tuning knobs (tools/), not API
        if (const char *e = getenv("RLHIP_DM_DIV")) c.dm_div = std::max(0, atoi(e));
    }
    int32_t *d_mode = nullptr;
    RL_HIP(t->pool.alloc(&d_mode, (size_t)F));
    c.mode = d_mode;
    hipLaunchKernelGGL(k_cumulate_counts, dim3(F), dim3(64), 0, s, TS, (const int32_t *)d_nthr, c.cum_cnt, d_mode);
    if (c.cum_cnt_loc) {       // (the mode bins are the GLOBAL ones: the local pass only cumulates)
        int32_t *d_mode_scratch = nullptr;
        RL_HIP(t->pool.alloc(&d_mode_scratch, (size_t)F));
        hipLaunchKernelGGL(k_cumulate_counts, dim3(F), dim3(64), 0, s, TS, (const int32_t *)d_nthr, c.cum_cnt_loc, d_mode_scratch);
    }
    {   // columns whose bins come in runs (nine in ten documents outside the mode bin are followed by an equal bin: a quad agrees 3 times in 4)
        unsigned long long *d_rs = nullptr;
        RL_HIP(t->pool.alloc(&d_rs, (size_t)2 * F));
        hipLaunchKernelGGL(k_run_stats, dim3(F), dim3(kThreads), 0, s, (const uint16_t *)d_bins, (const int32_t *)d_mode, N, Npad, d_rs);
        std::vector<unsigned long long> h_rs((size_t)2 * F);
        RL_HIP(hipMemcpyAsync(h_rs.data(), d_rs, h_rs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        RL_HIP(hipStreamSynchronize(s));
        t->pool.release(d_rs);
        std::vector<uint32_t> h_runs((size_t)c.numFG, 0u);
        const bool runs_on = !getenv("RLHIP_RUNS_OFF") && c.sub == 16 && TS <= kHistLdsStride;      // the instantiation that exists
        c.any_runs = 0;
        for (int f = 0; f < F && runs_on; f++)
            if (h_rs[2 * f] >= 64 && 10 * h_rs[2 * f + 1] >= 9 * h_rs[2 * f]) { h_runs[f / kHistFG] |= 1u << (f % kHistFG); c.any_runs = 1; }
        uint32_t *d_runs = nullptr;
        RL_HIP(t->pool.alloc(&d_runs, h_runs.size()));
        RL_HIP(hipMemcpy(d_runs, h_runs.data(), h_runs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c.runs = d_runs;
    }
    c.crows = nullptr; c.cr_stride = 0; c.cr_grp = nullptr;
    {   // compact rows for the child passes of sparse data (BASELINE.json configs[3]: 85 % of the cells sit in their column's mode bin), decided per
        // 16-column group: a group takes them when its rows average at most 5 entries outside the mode bins and at most one row in 20 needs the
        // dense fallback; built when at least half of the groups do.  RLHIP_CROWS=0 / 1 forces them off / on for every group
        const char *e = getenv("RLHIP_CROWS");
        const int force = e ? atoi(e) : -1;
        bool maybe = force == 1;
        if (c.any_runs) maybe = false;      // (launch_hist takes the RUNS instantiation on such data, which reads dense rows: the compact rows would be built and never read -- ADVICE r04)
        if (force < 0 && TS <= kHistLdsStride && c.sub == 16 && !c.any_runs) {
            // cheap look first (the exact root counts are on the device already): dense data -- more than 5 cells a row outside the mode bins on
            // average -- never builds the rows (a gigabyte of transient memory and 8 ms at the MSLR-WEB30K shape)
            std::vector<int32_t> hc((size_t)F * TS), hm((size_t)F);
            RL_HIP(hipStreamSynchronize(s));
            RL_HIP(hipMemcpy(hc.data(), c.cum_cnt, hc.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            RL_HIP(hipMemcpy(hm.data(), d_mode, hm.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            double cells = 0;
            const double n_all = (double)hc[(size_t)h_nthr[0] - 1];           // documents over all ranks (the counts are all-reduced when sharded)
            for (int f = 0; f < F; f++) {
                const int m = hm[f];
                cells += n_all - ((double)hc[(size_t)f * TS + m] - (m > 0 ? (double)hc[(size_t)f * TS + m - 1] : 0.0));
            }
            maybe = cells <= 5.0 * n_all * (double)c.numFG;
        }
        if (maybe && TS <= kHistLdsStride && c.sub == 16) {
            unsigned long long *d_st = nullptr; uint4 *d_cr = nullptr;
            const int crs = (c.numFG + 7) & ~7;           // rows start on 128-byte lines
            RL_HIP(t->pool.alloc(&d_st, (size_t)2 * c.numFG)); RL_HIP(hipMemsetAsync(d_st, 0, (size_t)2 * c.numFG * sizeof(unsigned long long), s));
            RL_HIP(t->pool.alloc(&d_cr, (size_t)Npad * crs));
            RL_HIP(hipMemsetAsync(d_cr, 0xff, (size_t)Npad * crs * sizeof(uint4), s));
            hipLaunchKernelGGL(k_compact_rows, dim3(4096), dim3(kThreads), 0, s, (const uint16_t *)d_gbins, (const int32_t *)d_mode, d_cr, N, Npad, c.numFG, crs, F, d_st);
            std::vector<unsigned long long> h_st((size_t)2 * c.numFG);
            RL_HIP(hipMemcpyAsync(h_st.data(), d_st, h_st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            RL_HIP(hipStreamSynchronize(s));
            t->pool.release(d_st);
            std::vector<uint8_t> h_cg((size_t)c.numFG, 0);
            int n_on = 0; double ents = 0, over = 0;
            for (int g = 0; g < c.numFG; g++) {
                const double eg = (double)h_st[2 * g], og = (double)h_st[2 * g + 1];
                const bool on = force == 1 || (eg <= 5.0 * (double)N && og * 20.0 <= (double)N);
                h_cg[g] = on ? 1 : 0;
                if (on) { n_on++; ents += eg; over += og; }
            }
            if (force == 1 || 2 * n_on >= c.numFG) {
                uint8_t *d_cg = nullptr;
                RL_HIP(t->pool.alloc(&d_cg, (size_t)c.numFG));
                RL_HIP(hipMemcpy(d_cg, h_cg.data(), h_cg.size(), hipMemcpyHostToDevice));
                c.crows = d_cr; c.cr_stride = crs; c.cr_grp = d_cg;
                t->cr_groups = n_on; t->cr_entries = ents; t->cr_overflow = over;
            } else t->pool.release(d_cr);
        }
    }
    RL_HIP(hipGetLastError());
    RL_HIP(hipStreamSynchronize(s));
    t->pool.release(Xt); t->pool.release(thr0); t->pool.release(fs.set);
    {   // ---- sparse-column path of the root pass (rl_csc.inc): the groups whose live columns keep at most 1 / dens of their cells
        // outside the mode bins get entry lists, blocked by the root pass's chunks
        c.sp_on = 0; c.sp_ngroups = 0;
        int dens = 3;
        if (const char *e = getenv("RLHIP_CSC_DENS")) dens = atoi(e);               // 0 = no sparse path (tools/, tests)
        const int rootCs = std::min(kChunk, std::max(kMinChunk, (((N + 63) / 64 + 255) & ~255)));   // == chunk_docs<true>(N)
        const int rootChunks = (N + rootCs - 1) / rootCs;
        if (dens > 0 && !t->dist && TS <= kHistLdsStride && c.sub == 16) {
            std::vector<int32_t> h_cnt((size_t)F * TS), h_mode(F);
            RL_HIP(hipMemcpy(h_cnt.data(), c.cum_cnt, h_cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            RL_HIP(hipMemcpy(h_mode.data(), d_mode, F * sizeof(int32_t), hipMemcpyDeviceToHost));
            std::vector<int32_t> glist; std::vector<uint8_t> isg(c.numFG, 0);
            int sp_cols = 0;
            for (int g = 0; g < c.numFG; g++) {
                int64_t cells = 0, live = 0;
                for (int f = g * kHistFG; f < std::min(F, (g + 1) * kHistFG); f++) {
                    if (h_nthr[f] <= 2) continue;                                   // dead column: its only bin is its mode bin
                    const int m = h_mode[f];
                    cells += (int64_t)N - ((int64_t)h_cnt[(size_t)f * TS + m] - (m > 0 ? h_cnt[(size_t)f * TS + m - 1] : 0));
                    live++;
                }
                if (cells * dens <= (int64_t)N * live) { isg[g] = 1; glist.push_back(g); sp_cols += (int)live; }   // groups of dead columns too: nothing to read
            }
            if (!glist.empty()) {
                const int nsg = (int)glist.size();
                int32_t *d_glist = nullptr, *d_cnt = nullptr, *d_off = nullptr; uint8_t *d_isg = nullptr; uint32_t *d_ent = nullptr;
                RL_HIP(t->pool.alloc(&d_glist, (size_t)nsg)); RL_HIP(t->pool.alloc(&d_cnt, (size_t)nsg * rootChunks)); RL_HIP(t->pool.alloc(&d_off, (size_t)nsg * rootChunks + 1));
                RL_HIP(t->pool.alloc(&d_isg, (size_t)c.numFG));
                RL_HIP(hipMemcpy(d_glist, glist.data(), nsg * sizeof(int32_t), hipMemcpyHostToDevice));
                RL_HIP(hipMemcpy(d_isg, isg.data(), c.numFG, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(k_sp_build<false>, dim3(nsg, rootChunks), dim3(kThreads), 0, s, c, (const int32_t *)d_glist, nsg, rootCs, d_cnt, (const int32_t *)nullptr, (uint32_t *)nullptr);
                RL_HIP(hipGetLastError());
                RL_HIP(hipStreamSynchronize(s));
                std::vector<int32_t> cnt((size_t)nsg * rootChunks), off((size_t)nsg * rootChunks + 1);
                RL_HIP(hipMemcpy(cnt.data(), d_cnt, cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
                int64_t E = 0;
                for (size_t i = 0; i < cnt.size(); i++) { off[i] = (int32_t)E; E += cnt[i]; }
                if (E < 2000000000ll) {
                    off[cnt.size()] = (int32_t)E;
                    RL_HIP(t->pool.alloc(&d_ent, (size_t)E));
                    RL_HIP(hipMemcpy(d_off, off.data(), off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                    hipLaunchKernelGGL(k_sp_build<true>, dim3(nsg, rootChunks), dim3(kThreads), 0, s, c, (const int32_t *)d_glist, nsg, rootCs, (int32_t *)nullptr, (const int32_t *)d_off, d_ent);
                    RL_HIP(hipGetLastError());
                    RL_HIP(hipStreamSynchronize(s));
                    c.sp_on = 1; c.sp_ngroups = nsg; c.sp_grp = d_isg; c.sp_glist = d_glist; c.sp_ent = d_ent; c.sp_off = d_off;
                    t->sp_entries = E; t->sp_cols = sp_cols;
                }
            }
        }
    }

    if (c.java && TS <= kJ2MaxBins && !getenv("RLHIP_JHIST_V1")) {
        // k_jhist2's deal of a feature's bins to the 8 wavefronts of its block: consecutive bins until a wavefront owns about an
        // eighth of the documents (root counts) or 64 bins; a bin that fills a share on its own keeps the wavefront to itself
        std::vector<int32_t> h_cnt((size_t)F * TS);
        RL_HIP(hipMemcpy(h_cnt.data(), c.cum_cnt, h_cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        std::vector<uint16_t> jmap((size_t)F * TS, 0), jinv((size_t)F * kJ2MaxBins, 0xffffu);
        std::vector<uint32_t> jone(F, 0);
        for (int f = 0; f < F; f++) {
            const int T = h_nthr[f];
            // a wavefront's quota = (documents not dealt yet) / (wavefronts left), taken when it starts; a bin joins the current wavefront
            // while that brings it closer to its quota.  A wavefront that owns ONE bin adds without an owner test (8 cycles a sample
            // against ~20): a bin that fills more than 40 % of a quota-sized share on its own is therefore kept alone.
            int w = 0, used = 0, wbins[kJ2Waves] = {0};
            int64_t docs = 0, left = N, quota = std::max<int64_t>(1, (int64_t)N / kJ2Waves);
            for (int tb = 0; tb < T; tb++) {
                const int64_t nb = (int64_t)h_cnt[(size_t)f * TS + tb] - (tb > 0 ? h_cnt[(size_t)f * TS + tb - 1] : 0);
                const bool room_later = (T - tb) <= (kJ2Waves - w - 1) * 64;          // the wavefronts after this one can still hold all remaining bins
                const bool alone = nb * 5 > quota * 2;
                const bool full = used == 64 || docs + nb / 2 > quota || alone || (used == 1 && docs * 5 > quota * 2);
                if (w < kJ2Waves - 1 && used > 0 && room_later && full) {
                    w++; used = 0; left -= docs; docs = 0;
                    quota = std::max<int64_t>(1, left / (kJ2Waves - w));
                }
                jmap[(size_t)f * TS + tb] = (uint16_t)((w << 8) | used);
                jinv[(size_t)f * kJ2MaxBins + w * 64 + used] = (uint16_t)tb;
                used++; docs += nb; wbins[w]++;
            }
            for (int v = 0; v < kJ2Waves; v++) if (wbins[v] == 1) jone[f] |= 1u << v;
        }
        uint16_t *d_map = nullptr, *d_inv = nullptr; uint32_t *d_one = nullptr;
        RL_HIP(t->pool.alloc(&d_map, jmap.size())); RL_HIP(t->pool.alloc(&d_inv, jinv.size())); RL_HIP(t->pool.alloc(&d_one, jone.size()));
        RL_HIP(hipMemcpy(d_map, jmap.data(), jmap.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        RL_HIP(hipMemcpy(d_inv, jinv.data(), jinv.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        RL_HIP(hipMemcpy(d_one, jone.data(), jone.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c.jmap = d_map; c.jinv = d_inv; c.jone = d_one;
    }

    // ---- query side: ideal DCGs with the qid-keyed cache quirk (NDCGScorer.java:114-122,134-143) --
    int maxq = std::max(t->tr.maxq, t->has_valid ? t->va.maxq : 0);
    std::vector<double> disc((size_t)maxq + 2);
    for (size_t i = 0; i < disc.size(); i++) disc[i] = discount_of((int)i);
    double *d_disc = nullptr;
    RL_HIP(t->pool.alloc(&d_disc, disc.size()));
    RL_HIP(hipMemcpy(d_disc, disc.data(), disc.size() * sizeof(double), hipMemcpyHostToDevice));
    c.disc = d_disc;
    {
        std::map<int64_t, double> cache;
        // -qrel: NDCGScorer.loadExternalRelevanceJudgment fills idealGains BEFORE any list is scored (:50-96): those qids never compute their own
        auto preload = [&](DataSet &d, int64_t anon_base) {
            for (int q = 0; q < d.Q && !d.ext_ideal.empty(); q++)
                if (d.ext_ideal[q] == d.ext_ideal[q]) cache[d.has_key ? (int64_t)d.qkey[q] : anon_base + q] = d.ext_ideal[q];
        };
        preload(t->tr, (int64_t)1 << 40);
        if (t->has_valid) preload(t->va, (int64_t)1 << 41);
        const std::map<int64_t, double> external = cache;
        auto run = [&](DataSet &d, int64_t anon_base, std::vector<double> &own, std::vector<double> &cached) {
            own.resize(d.Q); cached.resize(d.Q);
            for (int q = 0; q < d.Q; q++) {
                const int n = d.qoff[q + 1] - d.qoff[q];
                const int size = std::min(n, t->p.metric_k);
                const int64_t key = d.has_key ? (int64_t)d.qkey[q] : anon_base + q;
                { auto pre = external.find(key); if (pre != external.end()) { own[q] = cached[q] = pre->second; continue; } }
                own[q] = ideal_dcg(d.labels.data() + d.qoff[q], n, size, disc);
                auto it = cache.find(key);
                if (it == cache.end()) it = cache.emplace(key, own[q]).first;   // score() fills the cache in list order
                cached[q] = it->second;
            }
        };
        auto upload_rd = [&](DataSet &d) -> int {
            if (d.ext_rd.empty()) return RL_OK;
            RL_HIP(t->pool.alloc(&d.d_ext_rd, (size_t)d.Q));
            RL_HIP(hipMemcpy(d.d_ext_rd, d.ext_rd.data(), (size_t)d.Q * sizeof(int32_t), hipMemcpyHostToDevice));
            return RL_OK;
        };
        { int rcu = upload_rd(t->tr); if (rcu) return rcu; if (t->has_valid) { rcu = upload_rd(t->va); if (rcu) return rcu; } }
        std::vector<double> own, cached;
        run(t->tr, (int64_t)1 << 40, own, cached);
        int rc = upload_query_side(t, t->tr, own, cached);
        if (rc) return rc;
        if (t->has_valid) {
            run(t->va, (int64_t)1 << 41, own, cached);
            rc = upload_query_side(t, t->va, own, cached);
            if (rc) return rc;
        }
    }
    c.labels = t->tr.d_labels; c.qoff = t->tr.d_qoff; c.ideal0 = t->tr.d_ideal0; c.ideal1 = t->tr.d_ideal1;
    c.scores = t->tr.d_scores; c.ndcg_q = t->tr.d_ndcg;
    int32_t *d_fid = nullptr;
    RL_HIP(t->pool.alloc(&d_fid, (size_t)F));
    {   // feature sampling draws REAL features: the column behind every histogram feature, the later runs of a split table marked (half_wave_best_feature)
        std::vector<int32_t> fc((size_t)F);
        for (int v = 0; v < F; v++) {
            const int col = t->vcol.empty() ? v : t->vcol[v];
            fc[v] = (v > 0 && !t->vcol.empty() && t->vcol[v - 1] == col) ? (int32_t)((uint32_t)col | 0x80000000u) : col;
        }
        int32_t *d_fc = nullptr;
        RL_HIP(t->pool.alloc(&d_fc, (size_t)F));
        RL_HIP(hipMemcpy(d_fc, fc.data(), (size_t)F * sizeof(int32_t), hipMemcpyHostToDevice));
        c.fcol = d_fc;
    }
    {   // (ids of the histogram features: a virtual feature carries its real column's id)
        std::vector<int32_t> ids((size_t)F);
        for (int v = 0; v < F; v++) ids[v] = t->feature_ids[t->vcol.empty() ? v : t->vcol[v]];
        RL_HIP(hipMemcpy(d_fid, ids.data(), F * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    c.feature_ids = d_fid;

    // ---- per-round state -------------------------------------------------------------------------
    RL_HIP(t->pool.alloc(&c.lw, (size_t)N));
    RL_HIP(hipMemset(c.lw, 0, (size_t)N * sizeof(double2)));   // MART writes zero weights (MART.java:47-51)
    RL_HIP(t->pool.alloc(&c.q, (size_t)N)); RL_HIP(t->pool.alloc(&c.r, (size_t)N));
    RL_HIP(t->pool.alloc(&c.idx[0], (size_t)N)); RL_HIP(t->pool.alloc(&c.idx[1], (size_t)N));
    RL_HIP(t->pool.alloc(&c.ql[0], (size_t)N)); RL_HIP(t->pool.alloc(&c.ql[1], (size_t)N));
    RL_HIP(t->pool.alloc(&c.nodes, (size_t)c.NC + 2)); RL_HIP(t->pool.alloc(&c.st, (size_t)1));
    RL_HIP(hipMemset(c.st, 0, sizeof(TreeState)));
    t->tree_seq = 0;
    if (const char *e = getenv("RLHIP_STEP_AHEAD")) t->step_ahead = std::max(0, atoi(e));     // tuning knob
    if (const char *e = getenv("RLHIP_DIST_STEP_AHEAD")) t->dist_ahead = std::max(0, atoi(e));
    if (const char *e = getenv("RLHIP_DIST_TIMEOUT_S")) t->dist_timeout_s = std::max(1.0, atof(e));
    // the progress word is an optimisation: without host-visible coherent memory the host simply enqueues every step
    c.progress = nullptr;
    if (!t->h_progress && hipHostMalloc((void **)&t->h_progress, sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        t->h_progress = nullptr; (void)hipGetLastError();
    }
    if (t->h_progress) {
        *t->h_progress = 0;
        if (t->step_ahead > 0 && hipHostGetDevicePointer((void **)&c.progress, t->h_progress, 0) != hipSuccess) {
            c.progress = nullptr; (void)hipGetLastError();
        }
    }
    if (t->dist) {
        // Sharded runs: every rank must enqueue the same collectives, so the way a tree's end is detected has to be the same on
        // all of them: the progress word (deterministic rule in enqueue_round) only if every rank has one, else the stream
        // synchronisation at fixed steps.
        int32_t have = (c.progress != nullptr && !getenv("RLHIP_DIST_SYNC_STOP")) ? 1 : 0, *d_have = nullptr;
        RL_HIP(t->pool.alloc(&d_have, (size_t)1));
        RL_HIP(hipMemcpy(d_have, &have, sizeof(have), hipMemcpyHostToDevice));
        int rcd = t->dist->allreduce(d_have, 1, DT_I32, OP_MIN, t->stream);
        if (rcd) return rcd;
        RL_HIP(hipStreamSynchronize(t->stream));
        RL_HIP(hipMemcpy(&have, d_have, sizeof(have), hipMemcpyDeviceToHost));
        if (!have) c.progress = nullptr;
    }
    RL_HIP(hipMemset(c.nodes, 0, ((size_t)c.NC + 2) * sizeof(NodeRec)));
    RL_HIP(t->pool.alloc(&c.queue, (size_t)c.MAXN + 2));
    RL_HIP(t->pool.alloc(&c.part_sum, (size_t)c.maxChunks * F * TS)); RL_HIP(t->pool.alloc(&c.part_cnt, (size_t)c.maxChunks * F * TS));
    RL_HIP(t->pool.alloc(&c.part_tot, (size_t)std::max(c.maxChunks, (N + kMinChunk - 1) / kMinChunk) + 1));
    RL_HIP(t->pool.alloc(&c.fb, (size_t)kSpec * 2 * F));
    {   // records of features that never get a finish block: "no admissible split"
        std::vector<FeatBest> init((size_t)kSpec * 2 * F);
        const double m1 = -1.0;
        for (auto &r : init) { memcpy(&r.S, &m1, 8); r.hi = 0; r.lo = 0; r.tc = 0; }
        RL_HIP(hipMemcpy(c.fb, init.data(), init.size() * sizeof(FeatBest), hipMemcpyHostToDevice));
    }
    RL_HIP(t->pool.alloc(&c.fb_root, (size_t)2 + kSpec)); c.fb_sq = c.fb_root + 2;
    RL_HIP(t->pool.alloc(&c.tile_cnt, (size_t)c.nTiles)); RL_HIP(t->pool.alloc(&c.tile_sq, (size_t)c.nTiles));
    RL_HIP(t->pool.alloc(&c.tile_desc, (size_t)c.nTiles)); RL_HIP(hipMemset(c.tile_desc, 0, (size_t)c.nTiles * 8));
    RL_HIP(t->pool.alloc(&c.tile_gdesc, (size_t)c.nTiles / 64 + kSpec + 2)); RL_HIP(hipMemset(c.tile_gdesc, 0, ((size_t)c.nTiles / 64 + kSpec + 2) * 8));
    RL_HIP(t->pool.alloc(&c.grow_stats, (size_t)4)); RL_HIP(hipMemset(c.grow_stats, 0, 16));
    RL_HIP(t->pool.alloc(&c.grow_docs, (size_t)10)); RL_HIP(hipMemset(c.grow_docs, 0, 80));      // ([4..9]: bubble stamps, RL_ARR_BUBBLES)
    c.steplog = nullptr;
    if (getenv("RLHIP_STEPLOG")) { RL_HIP(t->pool.alloc(&c.steplog, (size_t)8 + 8 * kStepLogCap)); RL_HIP(hipMemset(c.steplog, 0, ((size_t)8 + 8 * kStepLogCap) * sizeof(int32_t))); }
    RL_HIP(t->pool.alloc(&c.clk, (size_t)64 * 32));
    RL_HIP(hipMemset(c.clk, 0, 64 * 32 * sizeof(long long)));
    c.trace_tree = -1;
    c.trace = nullptr;
    if (const char *e = getenv("RLHIP_TRACE_TREE")) {
        c.trace_tree = atoi(e);
        RL_HIP(t->pool.alloc(&c.trace, (size_t)64 * 3 * kTraceBlocks * kTraceStamps)); RL_HIP(hipMemset(c.trace, 0, (size_t)64 * 3 * kTraceBlocks * kTraceStamps * sizeof(long long)));
    }
    RL_HIP(t->pool.alloc(&c.leaf_node, (size_t)c.MAXN + 1)); RL_HIP(t->pool.alloc(&c.leaf_start, (size_t)c.MAXN + 2));
    RL_HIP(t->pool.alloc(&c.leaf_of, (size_t)c.Npad + 64));
    RL_HIP(t->pool.alloc(&c.round_metric, (size_t)2 * t->p.n_trees));
    RL_HIP(hipMemset(c.round_metric, 0, (size_t)2 * t->p.n_trees * sizeof(float)));
    RL_HIP(hipMemset(c.lw, 0, N * sizeof(double2)));
    // ensemble
    const size_t en = (size_t)t->p.n_trees * c.MAXN;
    RL_HIP(t->pool.alloc(&t->ens.feat_idx, en)); RL_HIP(t->pool.alloc(&t->ens.thr, en));
    RL_HIP(t->pool.alloc(&t->ens.left, en)); RL_HIP(t->pool.alloc(&t->ens.right, en));
    RL_HIP(t->pool.alloc(&t->ens.out, en)); RL_HIP(t->pool.alloc(&t->ens.count, en)); RL_HIP(t->pool.alloc(&t->ens.deviance, en));
    RL_HIP(t->pool.alloc(&t->ens.n_nodes, (size_t)t->p.n_trees));
    RL_HIP(hipMemset(t->ens.n_nodes, 0, t->p.n_trees * sizeof(int32_t)));
    RL_HIP(t->pool.alloc(&t->d_mean, (size_t)2));
    {
        int64_t Nmax = N;
        t->Qglobal = t->tr.Q; t->Qmax = t->tr.Q; t->Nglobal = N; c.Nglobal = N;
        if (t->dist) {      // sizes of every rank
            // every rank either has a validation shard or none has (checked below through the gathered sizes)
            int32_t *d_sz = nullptr, *d_all = nullptr;
            RL_HIP(t->pool.alloc(&d_sz, (size_t)4)); RL_HIP(t->pool.alloc(&d_all, (size_t)4 * t->n_ranks));
            const int32_t mine[4] = {N, t->tr.Q, t->has_valid ? t->va.Q : 0, t->has_valid ? 1 : 0};
            RL_HIP(hipMemcpy(d_sz, mine, sizeof(mine), hipMemcpyHostToDevice));
            int rcd = t->dist->allgather(d_sz, d_all, sizeof(mine), s); if (rcd) return rcd;
            std::vector<int32_t> all((size_t)4 * t->n_ranks), all_vQ;
            RL_HIP(hipStreamSynchronize(s));
            RL_HIP(hipMemcpy(all.data(), d_all, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            t->all_N.clear(); t->all_Q.clear(); t->Nglobal = 0; t->Qglobal = 0; t->Qmax = 0; t->vQglobal = 0; t->vQmax = 0;
            for (int r = 0; r < t->n_ranks; r++) {
                t->all_N.push_back(all[4 * r]); t->all_Q.push_back(all[4 * r + 1]); all_vQ.push_back(all[4 * r + 2]);
                t->Nglobal += all[4 * r]; t->Qglobal += all[4 * r + 1]; t->vQglobal += all[4 * r + 2];
                Nmax = std::max<int64_t>(Nmax, all[4 * r]); t->Qmax = std::max(t->Qmax, all[4 * r + 1]); t->vQmax = std::max(t->vQmax, all[4 * r + 2]);
                if ((all[4 * r + 3] != 0) != t->has_valid) return fail(RL_ERR_INVALID, "multi-GPU training: either every rank sets a validation shard or none does");
            }
            if (t->has_valid) {
                RL_HIP(t->pool.alloc(&t->d_vqsend, (size_t)t->vQmax)); RL_HIP(t->pool.alloc(&t->d_vqgath, (size_t)t->n_ranks * t->vQmax));
                RL_HIP(t->pool.alloc(&t->d_vqcat, (size_t)t->vQglobal)); RL_HIP(t->pool.alloc(&t->d_vallQ, (size_t)t->n_ranks));
                RL_HIP(hipMemcpy(t->d_vallQ, all_vQ.data(), t->n_ranks * sizeof(int32_t), hipMemcpyHostToDevice));
                RL_HIP(hipMemset(t->d_vqsend, 0, (size_t)t->vQmax * sizeof(double)));
            }
            if (t->Nglobal >= (int64_t)2147483647 - 4096) return fail(RL_ERR_UNSUPPORTED, "more than 2^31 documents in total");
            c.Nglobal = (int32_t)t->Nglobal;
        }
        int rc = alloc_chain(t, t->leaf_chain, c.MAXN + 1, 2, Nmax, true);
        if (rc) return rc;
        rc = alloc_chain(t, t->metric_chain, 1, 1, std::max(t->Qglobal, t->has_valid ? std::max(t->va.Q, t->vQglobal) : 0));
        if (rc) return rc;
        RL_HIP(t->pool.alloc(&t->d_seg_buf, (size_t)c.MAXN + 2));
        if (t->dist) {
            t->lsstride = c.MAXN + 2;
            // piece mode (rl_dist.inc): the leaves' float chains from every rank's own pieces; the leaf-owner exchange for what it does not cover
            // (more than 256 leaves: the gathered tables grow with leaves x ranks; RL_FLAG_SERIAL_CHAIN) or on request (RLHIP_DIST_OWNER_CHAINS=1)
            // pinned mailbox: the transfer sizes of k_plan_exchange / the pending pieces of k_chain_cross reach the host without a stream synchronisation
            if (!t->h_xmail && hipHostMalloc((void **)&t->h_xmail, (size_t)(4 * 64 + 1) * sizeof(long long), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
                memset(t->h_xmail, 0, (size_t)(4 * 64 + 1) * sizeof(long long));
                if (hipHostGetDevicePointer((void **)&t->d_xmail, t->h_xmail, 0) != hipSuccess) { t->d_xmail = nullptr; (void)hipGetLastError(); }
            } else (void)hipGetLastError();
            t->piece_force = getenv("RLHIP_PIECE_FORCE_MISS") ? atoi(getenv("RLHIP_PIECE_FORCE_MISS")) : 0;
            t->piece_chains = t->d_xmail != nullptr && !getenv("RLHIP_DIST_OWNER_CHAINS") && c.MAXN + 1 <= 256 && t->n_ranks <= 64 && !(t->p.flags & RL_FLAG_SERIAL_CHAIN);
            const bool owner_bufs = !t->piece_chains;
            if (owner_bufs) { rc = alloc_chain(t, t->gchain, c.MAXN + 1, 2, t->Nglobal, true); if (rc) return rc; }
            else { rc = alloc_chain(t, t->gchain, c.MAXN + 1, 2, 1024, true); if (rc) return rc; }       // (a stub: rl_get_array's statistics read leaf_chain in piece mode)
            RL_HIP(t->pool.alloc(&t->d_gx, owner_bufs ? (size_t)2 * t->Nglobal + 2 : (size_t)2));              // at worst one rank owns every leaf
            RL_HIP(t->pool.alloc(&t->d_send, owner_bufs ? (size_t)2 * N + 2 : (size_t)2));
            if (t->piece_chains) {
                const size_t n2 = (size_t)2 * (c.MAXN + 1), Rn = (size_t)t->n_ranks;
                RL_HIP(t->pool.alloc(&t->d_pc_loc, n2)); RL_HIP(t->pool.alloc(&t->d_pc_all, Rn * n2)); RL_HIP(t->pool.alloc(&t->d_pc_base, n2));
                RL_HIP(t->pool.alloc(&t->d_ptab, n2 * (kChainW + 1))); RL_HIP(t->pool.alloc(&t->d_gtab, Rn * n2 * (kChainW + 1)));
                RL_HIP(t->pool.alloc(&t->d_res_loc, n2)); RL_HIP(t->pool.alloc(&t->d_res_all, Rn * n2));
                RL_HIP(t->pool.alloc(&t->xstate.key, n2)); RL_HIP(t->pool.alloc(&t->xstate.rank, n2)); RL_HIP(t->pool.alloc(&t->xstate.pending, n2));
                RL_HIP(hipMemset(t->xstate.key, 0, n2 * 4)); RL_HIP(hipMemset(t->xstate.rank, 0, n2 * 4)); RL_HIP(hipMemset(t->xstate.pending, 0, n2 * 4));
                RL_HIP(hipMemset(t->d_res_all, 0, Rn * n2 * 4)); RL_HIP(hipMemset(t->d_ptab, 0, n2 * (kChainW + 1) * 4));
            }
            RL_HIP(t->pool.alloc(&t->d_own, (size_t)c.MAXN + 1)); RL_HIP(t->pool.alloc(&t->d_xtab, (size_t)(c.MAXN + 1) * (t->n_ranks + 1)));
            RL_HIP(t->pool.alloc(&t->d_gls, (size_t)t->n_ranks * t->lsstride));
            RL_HIP(t->pool.alloc(&t->d_gres, (size_t)t->n_ranks * 2 * (c.MAXN + 1) + 8));
            RL_HIP(t->pool.alloc(&t->d_qsend, (size_t)t->Qmax)); RL_HIP(t->pool.alloc(&t->d_qgath, (size_t)t->n_ranks * t->Qmax));
            RL_HIP(t->pool.alloc(&t->d_qcat, (size_t)t->Qglobal)); RL_HIP(t->pool.alloc(&t->d_allQ, (size_t)t->n_ranks));
            RL_HIP(hipMemcpy(t->d_allQ, t->all_Q.data(), t->n_ranks * sizeof(int32_t), hipMemcpyHostToDevice));
            RL_HIP(hipMemset(t->d_qsend, 0, (size_t)t->Qmax * sizeof(double)));
            c.limb_words = (t->Nglobal < (1ll << 25) && !getenv("RLHIP_LIMBS3")) ? 2 : 3;
            RL_HIP(t->pool.alloc(&c.dist_buf, ((size_t)F * TS * 3 + 4) * kSpec));
            RL_HIP(hipMemset(c.dist_buf, 0, ((size_t)F * TS * 3 + 4) * kSpec * sizeof(long long)));
        }
    }
    {   // ranked-order arrays + pair-term matrix of the lambda kernels
        DataSet &d = t->tr;
        RL_HIP(t->pool.alloc(&d.d_ss, (size_t)N)); RL_HIP(t->pool.alloc(&d.d_sl, (size_t)N));
        RL_HIP(t->pool.alloc(&d.d_srel, (size_t)N)); RL_HIP(t->pool.alloc(&d.d_sidx, (size_t)N)); RL_HIP(t->pool.alloc(&d.d_docq, (size_t)N));
        std::vector<int32_t> docq((size_t)N);
        for (int q = 0; q < d.Q; q++) for (int i = d.qoff[q]; i < d.qoff[q + 1]; i++) docq[i] = q;
        RL_HIP(hipMemcpy(d.d_docq, docq.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice));
        // the LDS-resident fused kernel serves every metric with few rows (k <= kLambdaFusedMaxK); longer cutoffs go through the pair-term matrix
        const bool fused = !c.mart && c.k <= kLambdaFusedMaxK && !getenv("RLHIP_LAMBDA_UNFUSED");
        if (!c.mart && !fused) {
            if ((size_t)N * c.k * sizeof(double2) > ((size_t)16 << 30)) return fail(RL_ERR_UNSUPPORTED, "this metric cutoff needs more than 16 GiB of pair terms");
            RL_HIP(t->pool.alloc(&t->d_T, (size_t)N * c.k));
        }
        RL_HIP(t->pool.alloc(&t->d_wmax, (size_t)std::max(std::max(d.Q, (N + kThreads - 1) / kThreads), 2048) + 1));
        if (c.metric == RL_METRIC_MAP) RL_HIP(t->pool.alloc(&d.d_aux_i, (size_t)N));
        if (c.metric == RL_METRIC_ERR) {     // R[] / np[]: the rank kernels only ever write the top min(k, n) positions of a list, the rest stays 0
            RL_HIP(t->pool.alloc(&d.d_aux_a, (size_t)N)); RL_HIP(t->pool.alloc(&d.d_aux_b, (size_t)N));
            RL_HIP(hipMemset(d.d_aux_a, 0, (size_t)N * sizeof(double))); RL_HIP(hipMemset(d.d_aux_b, 0, (size_t)N * sizeof(double)));
        }
        RL_HIP(hipDeviceSynchronize());
        int rc = launch_rank(t, d, c.scores, d.d_ndcg, true);      // ranking of the all-zero start scores (file order)
        if (rc) return rc;
    }
    RL_HIP(hipDeviceSynchronize());
    t->inited = true;
    return RL_OK;
}

int rl_boost_round(rl_trainer *t, rl_tree *out, float *train_metric, float *valid_metric, int32_t *stop)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!t->inited) return fail(RL_ERR_STATE, "rl_init has not been called");
    if (t->finished) return fail(RL_ERR_STATE, "rl_finish has been called");
    if (t->round >= t->p.n_trees) return fail(RL_ERR_STATE, "all n_trees rounds are done");
    RL_HIP(hipSetDevice(t->p.device));
    const int m = t->round;
    int rc = enqueue_round(t);
    if (rc) return rc;
    rc = sync_rounds(t);
    if (rc) return rc;
    if (train_metric) *train_metric = t->h_metrics[2 * (size_t)m];
    if (t->has_valid) {
        const float vm = t->h_metrics[2 * (size_t)m + 1];
        if (valid_metric) *valid_metric = vm;
        const double score = vm;                                     // LambdaMART.java:237-243
        if (score > t->best_score) { t->best_score = score; t->best_round = t->round - 1; }
    }
    if (stop) *stop = (m - t->best_round > t->p.early_stop_rounds) ? 1 : 0;      // :248
    if (out) return rl_get_tree(t, m, out);
    return RL_OK;
}

int rl_boost_rounds_async(rl_trainer *t, int32_t n)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!t->inited) return fail(RL_ERR_STATE, "rl_init has not been called");
    if (t->has_valid) return fail(RL_ERR_STATE, "asynchronous rounds cannot honour early stopping; use rl_boost_round");
    if (t->round + n > t->p.n_trees) return fail(RL_ERR_STATE, "more rounds than n_trees");
    RL_HIP(hipSetDevice(t->p.device));
    for (int i = 0; i < n; i++) { int rc = enqueue_round(t); if (rc) return rc; }
    return RL_OK;
}

int rl_sync(rl_trainer *t)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!t->inited) return fail(RL_ERR_STATE, "rl_init has not been called");
    RL_HIP(hipSetDevice(t->p.device));
    return sync_rounds(t);
}

static int final_score(rl_trainer *t, DataSet &d, double *out)
{
    hipStream_t s = t->stream;
    double *d_sc = nullptr;
    RL_HIP(t->pool.alloc(&d_sc, (size_t)d.N));
    hipLaunchKernelGGL(k_ensemble_eval, dim3((unsigned)std::min<int64_t>(8192, (d.N + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, t->ens,
                       t->ctx.MAXN, t->n_kept, (const float *)d.d_X, d.N, t->F, t->p.learning_rate, (float *)nullptr, d_sc);
    int rc = launch_rank(t, d, d_sc, d.d_ndcg, false);
    if (rc) return rc;
    if (t->dist) {
        const double *gq = nullptr;
        const bool valid = (&d == &t->va);
        rc = gather_queries(t, d.d_ndcg, &gq, valid);
        if (rc) return rc;
        hipLaunchKernelGGL(k_double_mean, dim3(1), dim3(64), 0, s, gq, valid ? t->vQglobal : t->Qglobal, t->d_mean);
    } else hipLaunchKernelGGL(k_double_mean, dim3(1), dim3(64), 0, s, (const double *)d.d_ndcg, d.Q, t->d_mean);
    RL_HIP(hipGetLastError());
    RL_HIP(hipStreamSynchronize(s));
    RL_HIP(hipMemcpy(out, t->d_mean, sizeof(double), hipMemcpyDeviceToHost));
    t->pool.release(d_sc);
    return RL_OK;
}

int rl_finish(rl_trainer *t, double *train_score, double *valid_score)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!t->inited) return fail(RL_ERR_STATE, "rl_init has not been called");
    RL_HIP(hipSetDevice(t->p.device));
    int rc = sync_rounds(t);
    if (rc) return rc;
    if ((int64_t)t->n_kept > (int64_t)t->best_round + 1) t->n_kept = t->best_round + 1;      // LambdaMART.java:254-256
    double ts = 0, vs = 0;
    rc = final_score(t, t->tr, &ts);                                                          // :259
    if (rc) return rc;
    if (train_score) *train_score = ts;
    if (t->has_valid) {
        rc = final_score(t, t->va, &vs);                                                      // :263
        if (rc) return rc;
        t->best_score = vs;
        if (valid_score) *valid_score = vs;
    }
    t->finished = true;
    return RL_OK;
}

int rl_num_trees(const rl_trainer *t, int32_t *n)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (n) *n = t->n_kept;
    return RL_OK;
}

int rl_tree_capacity(const rl_trainer *t, int32_t *cap)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!t->inited) return fail(RL_ERR_STATE, "rl_init has not been called");
    if (cap) *cap = t->ctx.MAXN;
    return RL_OK;
}

int rl_get_tree(const rl_trainer *t, int32_t i, rl_tree *out)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!out) return fail(RL_ERR_INVALID, "null tree");
    if (i < 0 || i >= t->synced_rounds) return fail(RL_ERR_INVALID, "tree index out of range (did you rl_sync?)");
    RL_HIP(hipSetDevice(t->p.device));
    HostTree h;
    int rc = fetch_tree(t, i, h);
    if (rc) return rc;
    out->n_nodes = h.n_nodes;
    if (out->cap < h.n_nodes) return fail(RL_ERR_INVALID, "rl_tree.cap too small");
    for (int j = 0; j < h.n_nodes; j++) {
        out->feature[j] = h.feature[j]; out->threshold[j] = h.threshold[j]; out->left[j] = h.left[j]; out->right[j] = h.right[j];
        out->output[j] = h.output[j];
        if (out->deviance) out->deviance[j] = h.deviance[j];
        if (out->count) out->count[j] = h.count[j];
    }
    return RL_OK;
}

int rl_get_round_metrics(const rl_trainer *t, int32_t round, float *train_metric, float *valid_metric)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (round < 0 || round >= t->synced_rounds) return fail(RL_ERR_INVALID, "round out of range (did you rl_sync?)");
    if (train_metric) *train_metric = t->h_metrics[2 * (size_t)round];
    if (valid_metric && t->has_valid) *valid_metric = t->h_metrics[2 * (size_t)round + 1];
    return RL_OK;
}

int rl_best_validation(const rl_trainer *t, int32_t *best_round, double *best_score)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (best_round) *best_round = t->best_round;
    if (best_score) *best_score = t->best_score;
    return RL_OK;
}

int rl_predict(rl_trainer *t, const float *X, int64_t n_docs, float *out)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!t->inited) return fail(RL_ERR_STATE, "rl_init has not been called");
    if (!X || !out || n_docs < 0) return fail(RL_ERR_INVALID, "bad argument");
    if (n_docs == 0) return RL_OK;
    RL_HIP(hipSetDevice(t->p.device));
    int rc = sync_rounds(t);
    if (rc) return rc;
    float *dX = nullptr, *dO = nullptr;
    RL_HIP(hipMalloc((void **)&dX, (size_t)n_docs * t->F * sizeof(float)));
    RL_HIP(hipMalloc((void **)&dO, (size_t)n_docs * sizeof(float)));
    RL_HIP(hipMemcpyAsync(dX, X, (size_t)n_docs * t->F * sizeof(float), hipMemcpyHostToDevice, t->stream));
    hipLaunchKernelGGL(k_ensemble_eval, dim3((unsigned)std::min<int64_t>(8192, (n_docs + kThreads - 1) / kThreads)), dim3(kThreads), 0, t->stream,
                       t->ens, t->ctx.MAXN, t->n_kept, (const float *)dX, n_docs, t->F, t->p.learning_rate, dO, (double *)nullptr);
    RL_HIP(hipGetLastError());
    RL_HIP(hipStreamSynchronize(t->stream));
    RL_HIP(hipMemcpy(out, dO, (size_t)n_docs * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(dX); (void)hipFree(dO);
    return RL_OK;
}

int rl_model_to_text(const rl_trainer *t, char *buf, int64_t cap, int64_t *needed)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!t->inited) return fail(RL_ERR_STATE, "rl_init has not been called");
    RL_HIP(hipSetDevice(t->p.device));
    if (t->synced_rounds < t->n_kept) return fail(RL_ERR_STATE, "rounds still in flight: call rl_sync first");
    std::vector<HostTree> trees((size_t)t->n_kept);
    for (int i = 0; i < t->n_kept; i++) { int rc = fetch_tree(t, i, trees[i]); if (rc) return rc; }
    ModelHeader h{t->p.n_trees, t->p.n_leaves, t->p.n_threshold, t->p.learning_rate, t->p.early_stop_rounds,
                  t->p.ranker == RL_RANKER_MART ? "MART" : "LambdaMART"};
    const std::string s = model_to_text(h, trees);                // LambdaMART.model()  LambdaMART.java:290-301
    if (needed) *needed = (int64_t)s.size() + 1;
    if (buf && cap >= (int64_t)s.size() + 1) memcpy(buf, s.c_str(), s.size() + 1);
    return RL_OK;
}

// ---- multi-GPU (rl_dist.inc) -------------------------------------------------------------------------
int rl_dist_unique_id(void *id_out)
{
    if (!id_out) return fail(RL_ERR_INVALID, "null id buffer");
    RcclApi &r = rccl();
    if (!r.lib || !r.GetUniqueId) return fail(RL_ERR_COMM, "librccl.so could not be loaded");
    const int rc = r.GetUniqueId(id_out);
    return rc == 0 ? RL_OK : fail(RL_ERR_COMM, std::string("ncclGetUniqueId: ") + r.GetErrorString(rc));
}

static int dist_precheck(rl_trainer *t, int32_t rank, int32_t n_ranks)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (t->inited) return fail(RL_ERR_STATE, "rl_dist_init must be called before rl_init");
    if (n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks) return fail(RL_ERR_INVALID, "bad rank / n_ranks (1..64 ranks)");
    return RL_OK;
}

int rl_dist_init(rl_trainer *t, const void *id, int32_t rank, int32_t n_ranks)
{
    int rc = dist_precheck(t, rank, n_ranks);
    if (rc) return rc;
    if (!id) return fail(RL_ERR_INVALID, "null unique id");
    RL_HIP(hipSetDevice(t->p.device));
    RcclApi &r = rccl();
    if (!r.lib || !r.CommInitRank || !r.AllReduce || !r.AllGather) return fail(RL_ERR_COMM, "librccl.so could not be loaded");
    std::unique_ptr<RcclBackend> b(new RcclBackend());
    UidVal u;
    memcpy(u.b, id, RL_UNIQUE_ID_BYTES);
    const int nrc = r.CommInitRank(&b->comm, n_ranks, u, rank);
    if (nrc != 0) return fail(RL_ERR_COMM, std::string("ncclCommInitRank: ") + r.GetErrorString(nrc));
    b->rank = rank; b->n = n_ranks;
    t->rank = rank; t->n_ranks = n_ranks;
    t->dist = std::move(b);
    return RL_OK;
}

int rl_dist_stats(const rl_trainer *t, int64_t *out)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!out) return fail(RL_ERR_INVALID, "null argument");
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (t->dist) {
        out[0] = t->dist->n_allreduce; out[1] = t->dist->b_allreduce; out[2] = t->dist->n_allgather; out[3] = t->dist->b_allgather;
        out[4] = t->dist->n_alltoall; out[5] = t->dist->b_alltoall; out[6] = t->dist->n_tie; out[7] = t->dist->b_tie;
    }
    return RL_OK;
}

int rl_dist_init_callback(rl_trainer *t, int32_t rank, int32_t n_ranks, rl_host_allreduce_fn allreduce, rl_host_allgather_fn allgather,
                          rl_host_alltoallv_fn alltoallv, void *user)
{
    int rc = dist_precheck(t, rank, n_ranks);
    if (rc) return rc;
    if (!allreduce || !allgather) return fail(RL_ERR_INVALID, "null callback");
    std::unique_ptr<CallbackBackend> b(new CallbackBackend());
    b->ar = allreduce; b->ag = allgather; b->aa = alltoallv; b->user = user; b->rank = rank; b->n = n_ranks;
    t->rank = rank; t->n_ranks = n_ranks;
    t->dist = std::move(b);
    return RL_OK;
}

// ---- introspection -------------------------------------------------------------------------------
int rl_hist_features(const rl_trainer *t, int32_t *n, int32_t *columns, int32_t cap)
{
    if (check_trainer(t) || !n) return fail(RL_ERR_INVALID, "null argument");
    if (!t->inited) return fail(RL_ERR_STATE, "rl_init has not been called");
    *n = t->ctx.F;
    if (columns) for (int32_t v = 0; v < t->ctx.F && v < cap; v++) columns[v] = t->vcol.empty() ? v : t->vcol[v];
    return RL_OK;
}

int rl_bin_stride(const rl_trainer *t, int32_t *stride)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!t->inited) return fail(RL_ERR_STATE, "rl_init has not been called");
    if (stride) *stride = t->ctx.TS;
    return RL_OK;
}

int rl_quant_exponent(const rl_trainer *t, int32_t *e)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!t->inited) return fail(RL_ERR_STATE, "rl_init has not been called");
    TreeState st;
    RL_HIP(hipMemcpy(&st, t->ctx.st, sizeof(st), hipMemcpyDeviceToHost));
    if (e) *e = st.E;
    return RL_OK;
}

int rl_get_array(rl_trainer *t, int32_t which, void *out, int64_t cap_bytes)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (!t->inited) return fail(RL_ERR_STATE, "rl_init has not been called");
    RL_HIP(hipSetDevice(t->p.device));
    RL_HIP(hipStreamSynchronize(t->stream));
    const Ctx &c = t->ctx;
    const void *src = nullptr; size_t bytes = 0;
    switch (which) {
    case RL_ARR_LAMBDA: case RL_ARR_WEIGHT: {        // interleaved on the device: strided copy of one component
        bytes = (size_t)c.N * 8;
        if ((int64_t)bytes > cap_bytes) return fail(RL_ERR_INVALID, "output buffer too small");
        RL_HIP(hipMemcpy2D(out, 8, (const char *)c.lw + (which == RL_ARR_WEIGHT ? 8 : 0), 16, 8, (size_t)c.N, hipMemcpyDeviceToHost));
        return RL_OK;
    }
    case RL_ARR_SCORE: src = c.scores; bytes = (size_t)c.N * 8; break;
    case RL_ARR_VALID_SCORE: if (!t->has_valid) return fail(RL_ERR_STATE, "no validation set"); src = t->va.d_scores; bytes = (size_t)t->va.N * 8; break;
    case RL_ARR_NBINS: src = c.nthr; bytes = (size_t)c.F * 4; break;
    case RL_ARR_THRESHOLDS: src = c.thr; bytes = (size_t)c.F * c.TS * 4; break;
    case RL_ARR_ROOT_COUNT: src = c.cum_cnt; bytes = (size_t)c.F * c.TS * 4; break;
    case RL_ARR_QUANT: src = c.q; bytes = (size_t)c.N * 8; break;
    case RL_ARR_NDCG_PER_QUERY: src = c.ndcg_q; bytes = (size_t)c.Q * 8; break;
    case RL_ARR_GROW_STATS: src = c.grow_stats; bytes = 16; break;
    case RL_ARR_GROW_DOCS: src = c.grow_docs; bytes = 32; break;
    case RL_ARR_BUBBLES: src = c.grow_docs + 4; bytes = 32; break;
    case RL_ARR_PIECE_STATS: {
        const int64_t v[2] = {t->piece_rounds, t->piece_misses};
        if ((int64_t)sizeof(v) > cap_bytes) return fail(RL_ERR_INVALID, "output buffer too small");
        memcpy(out, v, sizeof(v));
        return RL_OK;
    }
    case RL_ARR_SPARSE_INFO: {
        const int64_t v[8] = {c.sp_on ? c.sp_ngroups : 0, t->sp_entries, c.sp_on ? c.numFG - c.sp_ngroups : c.numFG, t->sp_cols,
                              c.crows ? t->cr_groups : 0, (int64_t)t->cr_entries, (int64_t)t->cr_overflow, c.cr_stride};
        if (cap_bytes < (int64_t)sizeof(v)) return fail(RL_ERR_INVALID, "output buffer too small");
        memcpy(out, v, sizeof(v));
        return RL_OK;
    }
    case RL_ARR_TIE_STATS: {
        const int64_t v[10] = {t->tie_stalls, t->tie_nodes, t->tie_chain_nodes, t->tie_chain_docs, t->tie_us, t->tie_spec_segs, t->tie_spec_miss, t->tie_spec_serial,
                                t->tie_batches, t->tie_regrown};
        if (cap_bytes < (int64_t)sizeof(v)) return fail(RL_ERR_INVALID, "output buffer too small");
        memcpy(out, v, sizeof(v));
        return RL_OK;
    }
    case RL_ARR_STEP_LOG: {
        bytes = ((size_t)8 + 8 * kStepLogCap) * sizeof(int32_t);
        if ((int64_t)bytes > cap_bytes) return fail(RL_ERR_INVALID, "output buffer too small");
        if (c.steplog) { RL_HIP(hipMemcpy(out, c.steplog, bytes, hipMemcpyDeviceToHost)); RL_HIP(hipMemset(c.steplog, 0, 32)); }      // reading empties the log
        else memset(out, 0, bytes);
        return RL_OK;
    }
    case RL_ARR_PHASE_CLOCKS: src = c.clk; bytes = 64 * 32 * sizeof(long long); break;
    case RL_ARR_BLOCK_TRACE: if (!c.trace) return fail(RL_ERR_STATE, "no block trace (RLHIP_TRACE_TREE unset)"); src = c.trace; bytes = (size_t)64 * 3 * kTraceBlocks * kTraceStamps * sizeof(long long); break;
    case RL_ARR_CHAIN_STATS: {
        if (cap_bytes < 24) return fail(RL_ERR_INVALID, "output buffer too small");
        RL_HIP(hipMemcpy(out, (t->dist ? t->gchain : t->leaf_chain).stats, 12, hipMemcpyDeviceToHost));
        RL_HIP(hipMemcpy((char *)out + 12, t->metric_chain.stats, 12, hipMemcpyDeviceToHost));
        return RL_OK;
    }
    case RL_ARR_CHAIN_MISS: {
        const ChainBufs &b = (t->dist && !t->piece_chains) ? t->gchain : t->leaf_chain;
        bytes = (size_t)2 * b.maxseg * 4;
        if ((int64_t)bytes > cap_bytes) return fail(RL_ERR_INVALID, "output buffer too small");
        RL_HIP(hipMemcpy(out, b.miss, bytes, hipMemcpyDeviceToHost));
        return RL_OK;
    }
    case RL_ARR_BINS: {
        bytes = (size_t)c.F * c.N * 2;
        if ((int64_t)bytes > cap_bytes) return fail(RL_ERR_INVALID, "output buffer too small");
        RL_HIP(hipMemcpy2D(out, (size_t)c.N * 2, c.bins, (size_t)c.Npad * 2, (size_t)c.N * 2, c.F, hipMemcpyDeviceToHost));
        return RL_OK;
    }
    case RL_ARR_ROOT_SUM:
    case RL_ARR_ROOT_SUM_FIXED: {
        const bool fixed = which == RL_ARR_ROOT_SUM_FIXED;
        bytes = (size_t)c.F * c.TS * (fixed ? 16 : 8);
        if ((int64_t)bytes > cap_bytes) return fail(RL_ERR_INVALID, "output buffer too small");
        void *d = nullptr;
        RL_HIP(hipMalloc(&d, bytes));
        RL_HIP(hipMemsetAsync(d, 0, bytes, t->stream));      // same stream as the kernel: t->stream is non-blocking
        hipLaunchKernelGGL(k_debug_root_sum, dim3(c.F), dim3(kThreads), 0, t->stream, c, fixed ? (double *)nullptr : (double *)d,
                           fixed ? (long long *)d : (long long *)nullptr);
        RL_HIP(hipStreamSynchronize(t->stream));
        RL_HIP(hipMemcpy(out, d, bytes, hipMemcpyDeviceToHost));
        (void)hipFree(d);
        return RL_OK;
    }
    case RL_ARR_ROOT_SUM_JAVA: {
        if (!c.java) return fail(RL_ERR_STATE, "RL_ARR_ROOT_SUM_JAVA needs RL_FLAG_JAVA_ORDER");
        bytes = (size_t)c.F * c.TS * 8;
        if ((int64_t)bytes > cap_bytes) return fail(RL_ERR_INVALID, "output buffer too small");
        RL_HIP(hipStreamSynchronize(t->stream));
        RL_HIP(hipMemcpy(out, c.jcum, bytes, hipMemcpyDeviceToHost));       // node 0 = the root
        return RL_OK;
    }
    default: return fail(RL_ERR_INVALID, "unknown array id");
    }
    if ((int64_t)bytes > cap_bytes) return fail(RL_ERR_INVALID, "output buffer too small");
    RL_HIP(hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
    return RL_OK;
}

int rl_debug_exp(const double *x, int32_t n, double *out_fast, double *out_ref)
{
    if (!x || !out_fast || !out_ref || n < 0) return fail(RL_ERR_INVALID, "bad argument");
    if (n == 0) return RL_OK;
    double *d = nullptr;
    RL_HIP(hipMalloc((void **)&d, (size_t)n * 3 * sizeof(double)));
    RL_HIP(hipMemcpy(d, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_exp_probe, dim3((n + 255) / 256), dim3(256), 0, 0, (const double *)d, n, d + n, d + 2 * (size_t)n);
    RL_HIP(hipGetLastError());
    RL_HIP(hipDeviceSynchronize());
    RL_HIP(hipMemcpy(out_fast, d + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    RL_HIP(hipMemcpy(out_ref, d + 2 * (size_t)n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return RL_OK;
}

int rl_debug_rho(const double *x, const double *den, int32_t n, double *out_fast, double *out_ref)
{
    if (!x || !out_fast || !out_ref || n < 0) return fail(RL_ERR_INVALID, "bad argument");
    if (n == 0) return RL_OK;
    double *d = nullptr;
    RL_HIP(hipMalloc((void **)&d, (size_t)n * 4 * sizeof(double)));
    RL_HIP(hipMemcpy(d, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    if (den) {
        RL_HIP(hipMemcpy(d + 3 * (size_t)n, den, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_div_probe, dim3((n + 255) / 256), dim3(256), 0, 0, (const double *)d, (const double *)(d + 3 * (size_t)n), n, d + n, d + 2 * (size_t)n);
    } else
        hipLaunchKernelGGL(k_rho_probe, dim3((n + 255) / 256), dim3(256), 0, 0, (const double *)d, n, d + n, d + 2 * (size_t)n);
    RL_HIP(hipGetLastError());
    RL_HIP(hipDeviceSynchronize());
    RL_HIP(hipMemcpy(out_fast, d + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    RL_HIP(hipMemcpy(out_ref, d + 2 * (size_t)n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return RL_OK;
}

int rl_debug_float_chain(int32_t device, const double *x, int64_t n, const int64_t *seg_start, int32_t n_seg, float *out, int32_t *stats)
{
    if (!x || !seg_start || !out || n < 0 || n_seg < 1 || n > 2147483647 / 2) return fail(RL_ERR_INVALID, "bad argument");
    for (int i = 0; i < n_seg; i++) if (seg_start[i] > seg_start[i + 1]) return fail(RL_ERR_INVALID, "segments must be ascending");
    if (seg_start[0] != 0 || seg_start[n_seg] != n) return fail(RL_ERR_INVALID, "segments must cover [0, n)");
    RL_HIP(hipSetDevice(device));
    std::unique_ptr<rl_trainer> t(new rl_trainer());      // only the pool, the stream and the chain bookkeeping are used
    memset(&t->ctx, 0, sizeof(t->ctx)); memset(&t->ens, 0, sizeof(t->ens)); memset(&t->p, 0, sizeof(t->p));
    t->p.device = device;
    RL_HIP(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    RL_HIP(hipFuncSetAttribute((const void *)k_chain_stitch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RL_HIP(hipFuncSetAttribute((const void *)k_tie_finish, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    struct Guard { rl_trainer *t; ~Guard() { (void)hipStreamSynchronize(t->stream); (void)hipStreamDestroy(t->stream); for (void *q : t->pinned) (void)hipHostFree(q); } } guard{t.get()};
    ChainBufs b;
    int rc = alloc_chain(t.get(), b, n_seg, 1, n, true);
    if (rc) return rc;
    std::vector<int32_t> ss(n_seg + 1), st0(n_seg + 1);
    int32_t tiles = 0;
    for (int i = 0; i <= n_seg; i++) {
        ss[i] = (int32_t)seg_start[i]; st0[i] = tiles;
        if (i < n_seg) tiles += (int32_t)((seg_start[i + 1] - seg_start[i] + kChainTile - 1) / kChainTile);
    }
    ChainPlan plan{n_seg, tiles, tiles + n_seg, (int32_t)n};
    double *dx = nullptr;
    RL_HIP(t->pool.alloc(&dx, (size_t)std::max<int64_t>(n, 1)));
    RL_HIP(hipMemcpy(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    RL_HIP(hipMemcpy(b.seg_start, ss.data(), ss.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    RL_HIP(hipMemcpy(b.seg_tile0, st0.data(), st0.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    RL_HIP(hipMemcpy(b.plan, &plan, sizeof(plan), hipMemcpyHostToDevice));
    ChainSource src{dx, nullptr, nullptr, nullptr, nullptr, nullptr};
    enqueue_chain(t.get(), b, src);
    RL_HIP(hipGetLastError());
    RL_HIP(hipStreamSynchronize(t->stream));
    RL_HIP(hipMemcpy(out, b.result, (size_t)n_seg * sizeof(float), hipMemcpyDeviceToHost));
    if (stats) RL_HIP(hipMemcpy(stats, b.stats, 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return RL_OK;
}

int rl_debug_membench(int32_t device, int32_t mode, int64_t bytes, int32_t stride, int32_t iters, double *avg_ms, double *alg_bytes)
{
    if (!avg_ms || bytes < 4096 || iters < 1 || mode < 0 || mode > 9 || (mode == 3 && stride < 1)) return fail(RL_ERR_INVALID, "bad argument");
    RL_HIP(hipSetDevice(device));
    if (mode >= 4) {
        // LDS atomics (k_mb_lds_atomic): `bytes` = atomics per thread (rounded to 16), `stride` unused; alg_bytes returns the 64-bit atomics of one launch
        hipDeviceProp_t prop;
        RL_HIP(hipGetDeviceProperties(&prop, device));
        const int reps = (int)std::max<int64_t>(1, bytes / 16);
        const unsigned gridl = (unsigned)prop.multiProcessorCount * 3u;
        unsigned long long *sinkl = nullptr;
        RL_HIP(hipMalloc((void **)&sinkl, gridl * sizeof(unsigned long long)));
        struct G2 { unsigned long long *p; ~G2() { (void)hipFree(p); } } g2{sinkl};
        hipEvent_t e0, e1;
        RL_HIP(hipEventCreate(&e0)); RL_HIP(hipEventCreate(&e1));
        const size_t ldsb = (size_t)16 * kHistLdsStride * 12;
        for (int it = -1; it < iters; it++) {
            if (it == 0) RL_HIP(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(k_mb_lds_atomic, dim3(gridl), dim3(kThreads), ldsb, nullptr, mode - 4, reps, sinkl);
        }
        RL_HIP(hipEventRecord(e1, nullptr));
        RL_HIP(hipEventSynchronize(e1));
        RL_HIP(hipGetLastError());
        float ms = 0;
        RL_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *avg_ms = (double)ms / iters;
        if (alg_bytes) *alg_bytes = (double)gridl * kThreads * (double)reps * 16.0;
        return RL_OK;
    }
    const size_t n16 = (size_t)bytes / 16;
    uint4 *a = nullptr, *b = nullptr; int *idx = nullptr; unsigned *sink = nullptr;
    struct Guard { void **p[4]; ~Guard() { for (auto q : p) if (*q) (void)hipFree(*q); } } guard{{(void **)&a, (void **)&b, (void **)&idx, (void **)&sink}};
    hipStream_t s = nullptr;
    RL_HIP(hipMalloc((void **)&a, n16 * 16));
    RL_HIP(hipMemset(a, 1, n16 * 16));
    if (mode == 0) { RL_HIP(hipMalloc((void **)&b, n16 * 16)); RL_HIP(hipMemset(b, 0, n16 * 16)); }
    const unsigned grid = 256 * 16;
    RL_HIP(hipMalloc((void **)&sink, grid * sizeof(unsigned)));
    size_t n_idx = 0;
    if (mode == 3) {
        n_idx = (n16 / 2) / (size_t)stride;
        if (n_idx == 0) return fail(RL_ERR_INVALID, "buffer too small for this stride");
        RL_HIP(hipMalloc((void **)&idx, n_idx * sizeof(int)));
        hipLaunchKernelGGL(k_mb_fill_idx, dim3(1024), dim3(kThreads), 0, s, idx, n_idx, stride, stride);
    }
    hipEvent_t e0, e1;
    RL_HIP(hipEventCreate(&e0)); RL_HIP(hipEventCreate(&e1));
    double bytes_per = 0;
    for (int it = -1; it < iters; it++) {       // it == -1: warm-up
        if (it == 0) RL_HIP(hipEventRecord(e0, s));
        switch (mode) {
        case 0: hipLaunchKernelGGL(k_mb_copy, dim3(grid), dim3(kThreads), 0, s, (const uint4 *)a, b, n16); bytes_per = 2.0 * n16 * 16; break;
        case 1: hipLaunchKernelGGL(k_mb_read, dim3(grid), dim3(kThreads), 0, s, (const uint4 *)a, n16, sink); bytes_per = 1.0 * n16 * 16; break;
        case 2: hipLaunchKernelGGL(k_mb_write, dim3(grid), dim3(kThreads), 0, s, a, n16); bytes_per = 1.0 * n16 * 16; break;
        default: hipLaunchKernelGGL(k_mb_gather32, dim3(grid), dim3(kThreads), 0, s, (const uint4 *)a, (const int *)idx, n_idx, sink); bytes_per = 36.0 * n_idx; break;
        }
    }
    RL_HIP(hipEventRecord(e1, s));
    RL_HIP(hipEventSynchronize(e1));
    RL_HIP(hipGetLastError());
    float ms = 0;
    RL_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_ms = (double)ms / iters;
    if (alg_bytes) *alg_bytes = bytes_per;
    return RL_OK;
}

int rl_set_timing_flags(rl_trainer *t, int32_t flags)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    const int32_t mask = RL_FLAG_TIMING | RL_FLAG_TIMING_NODES;
    t->p.flags = (t->p.flags & ~mask) | (flags & mask);
    return RL_OK;
}

int rl_get_timing(rl_trainer *t, int32_t kernel, double *total_ms, int64_t *launches, double *alg_bytes)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    if (kernel < 0 || kernel >= RL_KERNEL_COUNT_) return fail(RL_ERR_INVALID, "unknown kernel id");
    if (total_ms) *total_ms = t->timing[kernel].ms;
    if (launches) *launches = t->timing[kernel].launches;
    if (alg_bytes) *alg_bytes = t->timing[kernel].bytes;
    return RL_OK;
}

int rl_reset_timing(rl_trainer *t)
{
    if (check_trainer(t)) return RL_ERR_INVALID;
    for (auto &s : t->timing) s = TimingSlot();
    return RL_OK;
}


// ---- scoring-only model (Ensemble loaded from RankLib model text) -----------------------------------
}  // extern "C"

struct rl_model {
    int32_t device = 0;
    std::vector<HostTree> trees;
    std::vector<int32_t> features;
    int32_t maxn = 1;
    bool uniform_weight = true;
    DevPool pool;
    EnsTree ens;
    float *d_w = nullptr;
    unsigned long long *d_pack = nullptr;   // packed nodes for k_model_eval_tiled (null when the model does not fit the packing)
    unsigned char *d_perm = nullptr;        // [tiles][kEvalTreeTile] trees of a tile by descending depth (255 = none): walker wavefront p takes ranks 8 p .. 8 p + 7
    unsigned char *d_gdepth = nullptr;      // [tiles][kEvalParts] deepest leaf among a walker's trees = its lockstep walk length
    int32_t maxcol = 0;                     // largest column any node reads
};

namespace rl {
// like k_ensemble_eval but with a weight per tree (Ensemble.weights)
__global__ __launch_bounds__(kThreads) void k_model_eval(const EnsTree e, const float *w, int MAXN, int nt, const float *X, int64_t n,
                                                          int stride, float *out)
{
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float *row = X + (size_t)i * stride;
        float s = 0.f;
        for (int t = 0; t < nt; t++) {
            const size_t o = (size_t)t * MAXN;
            int nd = 0;
            while (e.feat_idx[o + nd] != -1) {
                const int fc = e.feat_idx[o + nd];
                const float v = (fc < stride) ? row[fc] : 0.f;                 // -missingZero  DenseDataPoint.java:22-25
                nd = (v <= e.thr[o + nd]) ? e.left[o + nd] : e.right[o + nd];
            }
            s = (float)((double)s + (double)e.out[o + nd] * (double)w[t]);     // Ensemble.java:113
        }
        out[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// K10: Ensemble.eval for many trees.  One block scores a tile of kEvalDocs documents against
// the whole ensemble:
//   * the tile's feature rows are transposed into LDS once, sX[column][doc] (columns the rows do not have are zero:
//     -missingZero): a lane reads sX[c * kEvalDocs + doc], so the bank is doc % 32 whatever column the lane's path asks
//     for -- conflict-free;
//   * trees stream through LDS in tiles of kEvalTreeTile as packed 8-byte nodes, children adjacent (right = left + 1):
//       bits 0..31 threshold (or leaf output) float bits | 32..47 byte offset of the column in sX (0xFFFF = leaf)
//       | 48..63 byte offset of the left child in the tree
//     so a step is: load node, load value, compare, add -- 7 VALU + 2 LDS instructions;
//   * walker wavefront p (of kEvalParts) owns trees [p*kEvalPer, (p+1)*kEvalPer) of the tile for all documents: kEvalPer
//     chains per lane in lockstep (a leaf is a fixed point of the step, so finished trees idle in place).  The walk is
//     bound by instruction issue (13 per chain step), not by LDS latency.  Measured and dropped: refilling a finished chain
//     with the lane's next tree (fewer steps, but a leaf branch that some lane takes at almost every step: 8.2 M docs/s
//     against 14.8), and a branch-free step with all value loads issued first (22 instructions per step: 9.7 M docs/s);
//   * one more wavefront does nothing but Ensemble.eval's accumulation  s = (float)(s + out * weight)  in tree order
//     (learning/tree/Ensemble.java:110-116) for the PREVIOUS tile (leaf outputs double-buffered in LDS), so the serial
//     float chain of a document overlaps the walk of the next tile instead of stalling the walkers;
//   * the next tile of trees is fetched into registers while the current one is walked.
// Needs: column offsets and child offsets that fit 16 bits, and the LDS budget; otherwise k_model_eval runs.
// ------------------------------------------------------------------------------------------------
#ifndef RL_EVAL_PARTS
#define RL_EVAL_PARTS 4
#endif
#ifndef RL_EVAL_PER
#define RL_EVAL_PER 8
#endif
constexpr int kEvalDocs = 64, kEvalParts = RL_EVAL_PARTS, kEvalPer = RL_EVAL_PER, kEvalTreeTile = kEvalParts * kEvalPer;
constexpr int kEvalThreads = kEvalDocs * (kEvalParts + 1), kEvalPrefetch = 8;     // 8-byte words each thread prefetches per tile
// Phases of a walker's walk (round 6; see the loop): chains still walking after the first phase, the second, .. and in the last one.  Same box,
// alternating libraries, 30 M rows x 10 000 trees (profiles/r06w_ab_infer_phased_walk.txt): one loop of eight chains 26.8 M docs/s | 8 -> 4: 27.9 |
// 8 -> 4 -> 2: 27.5 | 8 -> 6 -> 4 -> 2: 28.3 - 28.5 | a staircase 8 -> 7 -> .. -> 1: 23.3.
#ifndef RL_EVAL_PHASES
#define RL_EVAL_PHASES 3
#endif
#if RL_EVAL_PHASES == 1
constexpr int kEvalPhases = 1, kEvalPh1 = 4, kEvalPh2 = 2, kEvalPhLast = 4;
#elif RL_EVAL_PHASES == 2
constexpr int kEvalPhases = 2, kEvalPh1 = 4, kEvalPh2 = 2, kEvalPhLast = 2;
#else
constexpr int kEvalPhases = 3, kEvalPh1 = 6, kEvalPh2 = 4, kEvalPhLast = 2;
#endif
constexpr int kEvalMetaDepths = kEvalParts * (1 + kEvalPhases);      // per tile: every walker's steps, then the steps at which its phases end

static inline size_t eval_tiled_lds(int cols, int maxn)
{
    return (size_t)cols * kEvalDocs * 4 + (size_t)kEvalTreeTile * maxn * 8 + (size_t)2 * kEvalTreeTile * kEvalDocs * 4 + 2 * kEvalTreeTile * 4 + 2 * (kEvalTreeTile + kEvalMetaDepths);
}

// cols = max(row_stride, largest column any node reads + 1)
//
// The walk (round 4).  A leaf is packed as a node that loops onto itself: it "reads" column 0 -- no RankLib feature has id 0; the staged tile holds
// -infinity there -- so `x <= value` is always true and its left-child offset is its own.  A chain step is then the same eight instructions for
// every node (and, add, ds_read_b32, shift, compare, select, add3, ds_read_b64) with no leaf test and no branch, so the compiler issues the eight
// chains' feature loads back to back and their node loads back to back: the wavefront waits for LDS twice per step of EIGHT chains instead of
// twice per chain (the branchy version spent half of its time in those waits: 2.5 walker wavefronts per SIMD cannot hide them).  The trees of a
// tile, sorted by depth, are dealt round the walkers (perm / gdepth, built with the packing), deepest first inside a walker; a walker's deepest tree sets its
// number of steps, and since round 6 its chains drop out in phases as their trees end -- eight chains to the 7th tree's depth, six to the 5th's, four to the
// 3rd's, two to the deepest's (see kEvalPhases) -- instead of all eight idling on their leaves to the last step: 26.8 -> 28.4 M docs/s.  The accumulator
// adds the outputs in the ensemble's own order whatever walker produced them.
__global__ __launch_bounds__(kEvalThreads) void k_model_eval_tiled(const unsigned long long *nodes, const float *w, int MAXN, int nt,
                                                                   const float *X, int64_t n, int stride, int cols, float *out,
                                                                   const unsigned char *perm, const unsigned char *gdepth)
{
    extern __shared__ unsigned char ev_raw[];
    float *sX = (float *)ev_raw;                                               // [cols][kEvalDocs]
    unsigned long long *sT = (unsigned long long *)(sX + (size_t)cols * kEvalDocs);   // [kEvalTreeTile][MAXN]
    float *sO = (float *)(sT + (size_t)kEvalTreeTile * MAXN);                  // [2][kEvalTreeTile][kEvalDocs] leaf outputs (double buffer)
    float *sW = sO + 2 * kEvalTreeTile * kEvalDocs;                            // [2][kEvalTreeTile] tree weights
    unsigned char *sP = (unsigned char *)(sW + 2 * kEvalTreeTile);             // [2][kEvalTreeTile + kEvalMetaDepths] the tile's walker assignment and walk lengths
    const int tid = threadIdx.x, doc = tid & (kEvalDocs - 1), part = tid / kEvalDocs;
    const bool walker = part < kEvalParts;
    const int tile_words = kEvalTreeTile * MAXN;                               // <= kEvalThreads * kEvalPrefetch (checked by the host)
    const unsigned char *sXb = (const unsigned char *)sX + doc * 4;
    for (int64_t tile = blockIdx.x; tile * kEvalDocs < n; tile += gridDim.x) {
        const int64_t d0 = tile * kEvalDocs;
        const int nd = (int)min((int64_t)kEvalDocs, n - d0);
        __syncthreads();
        const float *src = X + (size_t)d0 * stride;                            // the tile is one contiguous range of X
        for (int e = tid; e < nd * stride; e += kEvalThreads) { const int dd = e / stride, c = e - dd * stride; sX[c * kEvalDocs + dd] = src[e]; }
        for (int e = tid; e < (cols - stride) * kEvalDocs; e += kEvalThreads) sX[stride * kEvalDocs + e] = 0.f;
        float s = 0.f;                                                         // the accumulator wavefront's running Ensemble.eval sum
        unsigned long long pre[kEvalPrefetch];
#pragma unroll
        for (int u = 0; u < kEvalPrefetch; u++) { const int e = tid + u * kEvalThreads; pre[u] = (e < min(tile_words, nt * MAXN)) ? nodes[e] : 0ull; }
        __syncthreads();
        if (tid < kEvalDocs) sX[tid] = -__builtin_inff();                      // column 0: what a leaf "reads" (after the staging pass wrote the rows' column 0)
        int k = 0, tt_prev = 0;
        for (int t0 = 0; t0 < nt; t0 += kEvalTreeTile, k++) {
            const int tt = min(kEvalTreeTile, nt - t0);
            const int cb = k & 1;
            __syncthreads();                                                   // tile k-1 walked (its outputs complete), sT free
#pragma unroll
            for (int u = 0; u < kEvalPrefetch; u++) { const int e = tid + u * kEvalThreads; if (e < tile_words) sT[e] = pre[u]; }
            if (tid < tt) sW[cb * kEvalTreeTile + tid] = w[t0 + tid];
            if (tid < kEvalTreeTile) sP[cb * (kEvalTreeTile + kEvalMetaDepths) + tid] = perm[(size_t)k * kEvalTreeTile + tid];
            else if (tid < kEvalTreeTile + kEvalMetaDepths) sP[cb * (kEvalTreeTile + kEvalMetaDepths) + tid] = gdepth[(size_t)k * kEvalMetaDepths + (tid - kEvalTreeTile)];
            __syncthreads();
            {   // next tile -> registers (in flight during the walk)
                const size_t nb = (size_t)(t0 + kEvalTreeTile) * MAXN;
                const long long left = (long long)nt * MAXN - (long long)nb;
#pragma unroll
                for (int u = 0; u < kEvalPrefetch; u++) { const int e = tid + u * kEvalThreads; pre[u] = (e < tile_words && e < left) ? nodes[nb + e] : 0ull; }
            }
            if (walker) {
                const unsigned char *pp = sP + cb * (kEvalTreeTile + kEvalMetaDepths);
                const int depth = __builtin_amdgcn_readfirstlane((int)pp[kEvalTreeTile + part]);      // wave-uniform: a scalar loop bound
                if (depth > 0) {
                    float *so = sO + (size_t)cb * kEvalTreeTile * kEvalDocs + doc;
                    const unsigned char *tb[kEvalPer];
                    unsigned long long v[kEvalPer];
                    int li[kEvalPer];
#pragma unroll
                    for (int u = 0; u < kEvalPer; u++) {
                        li[u] = __builtin_amdgcn_readfirstlane((int)pp[part * kEvalPer + u]);         // 255: no such tree in this (last) tile -- the chain walks tree 0 again, unstored
                        tb[u] = (const unsigned char *)(sT + (size_t)(li[u] < tt ? li[u] : 0) * MAXN);
                        v[u] = *(const unsigned long long *)tb[u];
                    }
                    // The walker's trees come deepest first.  All eight chains walk for as many steps as the walker's (kEvalPh1 + 1)-th tree has levels, then the
                    // kEvalPh1 deepest for as many as the (kEvalPh2 + 1)-th has, ... : a chain that has reached its leaf in every lane stops costing instructions
                    // (in ONE loop to the deepest tree's depth the shallow chains idled on their leaves, at the cost of their instructions).
                    int step = 0;
#define RL_EVAL_PHASE(NCH, UNTIL)                                                                                                          \
                    for (; step < (UNTIL); step++) {                                                                                       \
                        float x[NCH];                                                                                                      \
                        _Pragma("unroll") for (int u = 0; u < NCH; u++) x[u] = *(const float *)(sXb + ((unsigned)(v[u] >> 32) & 0xffffu)); \
                        _Pragma("unroll") for (int u = 0; u < NCH; u++) {       /* Split.eval: value <= threshold goes left (Split.java:118); a leaf stays where it is */ \
                            const unsigned off = (unsigned)(v[u] >> 48) + ((x[u] <= __uint_as_float((unsigned)v[u])) ? 0u : 8u);           \
                            v[u] = *(const unsigned long long *)(tb[u] + off);                                                             \
                        }                                                                                                                  \
                    }
                    const unsigned char *pd = pp + kEvalTreeTile + kEvalParts + part * kEvalPhases;
                    const int end0 = __builtin_amdgcn_readfirstlane((int)pd[0]);           // (scalar loop bounds, read once)
                    [[maybe_unused]] const int end1 = __builtin_amdgcn_readfirstlane((int)pd[kEvalPhases >= 2 ? 1 : 0]);
                    [[maybe_unused]] const int end2 = __builtin_amdgcn_readfirstlane((int)pd[kEvalPhases >= 3 ? 2 : 0]);
                    RL_EVAL_PHASE(kEvalPer, end0)
#if RL_EVAL_PHASES >= 2
                    RL_EVAL_PHASE(kEvalPh1, end1)
#endif
#if RL_EVAL_PHASES >= 3
                    RL_EVAL_PHASE(kEvalPh2, end2)
#endif
                    RL_EVAL_PHASE(kEvalPhLast, depth)
#undef RL_EVAL_PHASE
                    if (doc < nd) {
#pragma unroll
                        for (int u = 0; u < kEvalPer; u++) if (li[u] < tt) so[li[u] * kEvalDocs] = __uint_as_float((unsigned)v[u]);
                    }
                }
            } else if (k > 0 && doc < nd) {                                    // accumulate the previous tile while this one is walked
                const float *po = sO + (size_t)(cb ^ 1) * kEvalTreeTile * kEvalDocs + doc, *pw = sW + (cb ^ 1) * kEvalTreeTile;
                for (int t = 0; t < tt_prev; t++) s = (float)((double)s + (double)po[t * kEvalDocs] * (double)pw[t]);   // Ensemble.java:113
            }
            tt_prev = tt;
        }
        __syncthreads();
        if (!walker && doc < nd) {
            if (k > 0) {
                const int cb = (k - 1) & 1;
                const float *po = sO + (size_t)cb * kEvalTreeTile * kEvalDocs + doc, *pw = sW + cb * kEvalTreeTile;
                for (int t = 0; t < tt_prev; t++) s = (float)((double)s + (double)po[t * kEvalDocs] * (double)pw[t]);
            }
            out[d0 + doc] = s;
        }
    }
}
}  // namespace rl

extern "C" {

int rl_model_from_text(const char *text, int32_t device, rl_model **out)
{
    if (!text || !out) return fail(RL_ERR_INVALID, "null argument");
    *out = nullptr;
    std::unique_ptr<rl_model> m(new rl_model());
    std::string err;
    if (!model_from_text(text, m->trees, err)) return fail(RL_ERR_INVALID, "Error in Emsemble(xmlRepresentation): " + err);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RL_ERR_NO_DEVICE, "no HIP device visible: librlhip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(RL_ERR_INVALID, "device ordinal out of range");
    m->device = device;
    RL_HIP(hipSetDevice(device));
    std::map<int32_t, int> fids;
    for (auto &t : m->trees) { m->maxn = std::max(m->maxn, t.n_nodes); for (int f : t.feature) if (f != -1) fids[f] = 0; }
    for (auto &kv : fids) m->features.push_back(kv.first);
    const size_t nt = m->trees.size(), en = std::max<size_t>(1, nt * m->maxn);
    std::vector<int32_t> fi(en, -1), le(en, -1), ri(en, -1);
    std::vector<float> th(en, 0.f), ou(en, 0.f), w(std::max<size_t>(1, nt), 0.f);
    for (size_t i = 0; i < nt; i++) {
        const HostTree &t = m->trees[i];
        w[i] = t.weight;
        for (int j = 0; j < t.n_nodes; j++) {
            const size_t o = i * m->maxn + j;
            fi[o] = t.feature[j]; le[o] = t.left[j]; ri[o] = t.right[j]; th[o] = t.threshold[j]; ou[o] = t.output[j];
        }
    }
    memset(&m->ens, 0, sizeof(m->ens));
    RL_HIP(m->pool.alloc(&m->ens.feat_idx, en)); RL_HIP(m->pool.alloc(&m->ens.left, en)); RL_HIP(m->pool.alloc(&m->ens.right, en));
    RL_HIP(m->pool.alloc(&m->ens.thr, en)); RL_HIP(m->pool.alloc(&m->ens.out, en)); RL_HIP(m->pool.alloc(&m->d_w, w.size()));
    RL_HIP(hipMemcpy(m->ens.feat_idx, fi.data(), en * 4, hipMemcpyHostToDevice));
    RL_HIP(hipMemcpy(m->ens.left, le.data(), en * 4, hipMemcpyHostToDevice));
    RL_HIP(hipMemcpy(m->ens.right, ri.data(), en * 4, hipMemcpyHostToDevice));
    RL_HIP(hipMemcpy(m->ens.thr, th.data(), en * 4, hipMemcpyHostToDevice));
    RL_HIP(hipMemcpy(m->ens.out, ou.data(), en * 4, hipMemcpyHostToDevice));
    RL_HIP(hipMemcpy(m->d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    {   // packed nodes (see k_model_eval_tiled): breadth-first renumbering puts siblings next to each other
        int maxcol = 0;
        for (int32_t f : m->features) maxcol = std::max(maxcol, f);
        m->maxcol = maxcol;
        bool ok = nt > 0 && (size_t)m->maxn * 8 < 0x10000 && (size_t)(maxcol + 1) * kEvalDocs * 4 < 0xffff &&
                  (size_t)kEvalTreeTile * m->maxn <= (size_t)kEvalThreads * kEvalPrefetch;
        for (int32_t f : m->features) ok = ok && f >= 1;        // column 0 is what a packed leaf reads (no RankLib feature has id 0: learning/DataPoint.java:33)
        if (ok) {
            std::vector<unsigned long long> pk(en, 0ull);           // (padding: a leaf at offset 0 with value +0.0)
            std::vector<int> order, newid, lvl;
            std::vector<int> tdepth(nt, 0);
            for (size_t i = 0; i < nt && ok; i++) {
                const HostTree &t = m->trees[i];
                order.assign(1, 0); newid.assign(t.n_nodes, -1); newid[0] = 0; lvl.assign(1, 0);
                for (size_t h = 0; h < order.size(); h++) {
                    const int j = order[h];
                    if (t.feature[j] == -1) continue;
                    if (t.left[j] < 0 || t.right[j] < 0 || t.left[j] >= t.n_nodes || t.right[j] >= t.n_nodes || (int)order.size() + 2 > t.n_nodes) { ok = false; break; }
                    newid[t.left[j]] = (int)order.size(); order.push_back(t.left[j]); lvl.push_back(lvl[h] + 1);
                    newid[t.right[j]] = (int)order.size(); order.push_back(t.right[j]); lvl.push_back(lvl[h] + 1);
                    tdepth[i] = std::max(tdepth[i], lvl[h] + 1);
                }
                for (size_t h = 0; h < order.size() && ok; h++) {
                    const int j = order[h];
                    const bool leaf = t.feature[j] == -1;
                    uint32_t bits; const float fv = leaf ? t.output[j] : t.threshold[j];
                    memcpy(&bits, &fv, 4);
                    if (leaf && std::isnan(fv)) { ok = false; break; }       // a leaf loops through `-inf <= value`: NaN outputs take the generic kernel
                    // leaf: reads column 0 (-infinity in the staged tile) and its left child is itself
                    const unsigned long long co = leaf ? 0ull : (unsigned long long)t.feature[j] * kEvalDocs * 4;
                    const unsigned long long lo = leaf ? (unsigned long long)h * 8 : (unsigned long long)newid[t.left[j]] * 8;
                    pk[i * m->maxn + h] = (unsigned long long)bits | (co << 32) | (lo << 48);
                }
                if (tdepth[i] > 250) ok = false;
            }
            if (ok) {
                // the trees of a tile go to the walker wavefronts by descending depth (stable), eight each
                const size_t ntl = (nt + kEvalTreeTile - 1) / kEvalTreeTile;
                std::vector<unsigned char> pm(ntl * kEvalTreeTile, 255), gd(ntl * kEvalMetaDepths, 0);       // gd: per tile the walkers' steps, then per walker the steps at which its phases end
                std::vector<int> idx;
                for (size_t tl = 0; tl < ntl; tl++) {
                    const size_t t0 = tl * kEvalTreeTile, tt = std::min<size_t>(kEvalTreeTile, nt - t0);
                    idx.resize(tt);
                    for (size_t q = 0; q < tt; q++) idx[q] = (int)q;
                    std::stable_sort(idx.begin(), idx.end(), [&](int a2, int b2) { return tdepth[t0 + a2] > tdepth[t0 + b2]; });
                    // The sorted trees are dealt ROUND the walkers (walker p: ranks p, p + 4, p + 8, ..; deepest first inside a walker as the phases need it): every
                    // walker spans the tile's whole range of depths, so its chains drop out early and the four walkers reach the tile's barrier together.
                    // With eight consecutive ranks each (rounds 4 - 5, RLHIP_EVAL_DEAL=0) walker 0 held the eight deepest trees -- little to drop, and the others
                    // waited for it: 28.2 against 28.8 M docs/s (profiles/r06w_ab_infer_phased_walk.txt).
                    static const bool deal_rr = !(getenv("RLHIP_EVAL_DEAL") && atoi(getenv("RLHIP_EVAL_DEAL")) == 0);
                    if (deal_rr && tt == (size_t)kEvalTreeTile) {
                        std::vector<int> rr(tt);
                        for (size_t q = 0; q < tt; q++) rr[(q % kEvalParts) * kEvalPer + q / kEvalParts] = idx[q];
                        idx = rr;
                    }
                    for (size_t q = 0; q < tt; q++) {
                        pm[tl * kEvalTreeTile + q] = (unsigned char)idx[q];
                        unsigned char &g = gd[tl * kEvalMetaDepths + q / kEvalPer];
                        g = std::max<unsigned char>(g, (unsigned char)std::max(tdepth[t0 + idx[q]], 1));      // (a single-leaf tree still stores its output: one step)
                    }
                    // a phase of n chains ends when the walker's (n' + 1)-th tree (n' = the next phase's chains) is done: that tree's depth.  A walker with fewer
                    // trees: 0 (the phase is skipped).  RLHIP_EVAL_PHASED=0: every phase runs to the walker's full depth (one loop, rounds 4 - 5).
                    static const bool phased = !(getenv("RLHIP_EVAL_PHASED") && atoi(getenv("RLHIP_EVAL_PHASED")) == 0);
                    const int next_ch[3] = {kEvalPhases >= 2 ? kEvalPh1 : kEvalPhLast, kEvalPhases >= 3 ? kEvalPh2 : kEvalPhLast, kEvalPhLast};
                    for (int p = 0; p < kEvalParts; p++)
                        for (int ph = 0; ph < kEvalPhases; ph++) {
                            const size_t q = (size_t)p * kEvalPer + next_ch[ph];
                            unsigned char &e = gd[tl * kEvalMetaDepths + kEvalParts + p * kEvalPhases + ph];
                            e = !phased ? gd[tl * kEvalMetaDepths + p] : (q < tt ? (unsigned char)std::max(tdepth[t0 + idx[q]], 1) : 0);
                        }
                }
                RL_HIP(m->pool.alloc(&m->d_perm, pm.size())); RL_HIP(m->pool.alloc(&m->d_gdepth, gd.size()));
                RL_HIP(hipMemcpy(m->d_perm, pm.data(), pm.size(), hipMemcpyHostToDevice));
                RL_HIP(hipMemcpy(m->d_gdepth, gd.data(), gd.size(), hipMemcpyHostToDevice));
                RL_HIP(m->pool.alloc(&m->d_pack, en));
                RL_HIP(hipMemcpy(m->d_pack, pk.data(), en * 8, hipMemcpyHostToDevice));
                RL_HIP(hipFuncSetAttribute((const void *)k_model_eval_tiled, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
        }
    }
    *out = m.release();
    return RL_OK;
}

void rl_model_destroy(rl_model *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    delete m;
}

int rl_model_num_trees(const rl_model *m, int32_t *n)
{
    if (!m) return fail(RL_ERR_INVALID, "null model");
    if (n) *n = (int32_t)m->trees.size();
    return RL_OK;
}

int rl_model_features(const rl_model *m, int32_t *ids, int32_t cap, int32_t *n)
{
    if (!m) return fail(RL_ERR_INVALID, "null model");
    if (n) *n = (int32_t)m->features.size();
    if (ids) for (int i = 0; i < cap && i < (int)m->features.size(); i++) ids[i] = m->features[i];
    return RL_OK;
}

static int model_eval_launch(rl_model *m, const float *dX, int64_t n_docs, int32_t row_stride, float *dO, hipStream_t s)
{
    const int cols = std::max(row_stride, m->maxcol + 1);
    const size_t lds = eval_tiled_lds(cols, m->maxn);
    static const bool force_generic = getenv("RLHIP_EVAL_GENERIC") != nullptr;       // cross-checks in the tests
    if (m->d_pack && lds <= (size_t)160 * 1024 && !force_generic) {
        const int64_t tiles = (n_docs + kEvalDocs - 1) / kEvalDocs;
        hipLaunchKernelGGL(k_model_eval_tiled, dim3((unsigned)std::min<int64_t>(tiles, 256 * 256)), dim3(kEvalThreads), lds, s,
                           (const unsigned long long *)m->d_pack, (const float *)m->d_w, m->maxn, (int)m->trees.size(), dX, n_docs, row_stride, cols, dO,
                           (const unsigned char *)m->d_perm, (const unsigned char *)m->d_gdepth);
    } else {
        hipLaunchKernelGGL(k_model_eval, dim3((unsigned)std::min<int64_t>(8192, (n_docs + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, m->ens,
                           (const float *)m->d_w, m->maxn, (int)m->trees.size(), dX, n_docs, row_stride, dO);
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int rl_model_predict(rl_model *m, const float *X, int64_t n_docs, int32_t row_stride, float *out)
{
    if (!m) return fail(RL_ERR_INVALID, "null model");
    if (!X || !out || n_docs < 0 || row_stride < 1) return fail(RL_ERR_INVALID, "bad argument");
    if (n_docs == 0) return RL_OK;
    RL_HIP(hipSetDevice(m->device));
    float *dX = nullptr, *dO = nullptr;
    RL_HIP(hipMalloc((void **)&dX, (size_t)n_docs * row_stride * sizeof(float)));
    RL_HIP(hipMalloc((void **)&dO, (size_t)n_docs * sizeof(float)));
    RL_HIP(hipMemcpy(dX, X, (size_t)n_docs * row_stride * sizeof(float), hipMemcpyHostToDevice));
    int rc = model_eval_launch(m, dX, n_docs, row_stride, dO, 0);
    if (rc == RL_OK) { RL_HIP(hipDeviceSynchronize()); RL_HIP(hipMemcpy(out, dO, (size_t)n_docs * sizeof(float), hipMemcpyDeviceToHost)); }
    (void)hipFree(dX); (void)hipFree(dO);
    return rc;
}

int rl_model_predict_device(rl_model *m, const float *dX, int64_t n_docs, int32_t row_stride, float *dOut, void *stream)
{
    if (!m) return fail(RL_ERR_INVALID, "null model");
    if (!dX || !dOut || n_docs < 0 || row_stride < 1) return fail(RL_ERR_INVALID, "bad argument");
    if (n_docs == 0) return RL_OK;
    RL_HIP(hipSetDevice(m->device));
    return model_eval_launch(m, dX, n_docs, row_stride, dOut, (hipStream_t)stream);
}

}  // extern "C"
