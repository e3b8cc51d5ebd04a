// rl_internal.h -- shared host/device declarations of librlhip (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/rlhip.h"

namespace rl {

constexpr int kThreads = 256;        // 4 wavefronts of 64
constexpr int kWave = 64;
constexpr int kFinThreads = 320;     // k_hist_finish / k_hist_reduce blocks: 5 wavefronts cover the 257 bins of -tc 256 in one pass
constexpr int kLog2Chunk = 14;       // max docs accumulated into one int64 LDS accumulator (fixes the fixed-point exponent)
constexpr int kChunk = 1 << kLog2Chunk;   // chunk of the root histogram
#ifndef RL_NODE_CHUNK
#define RL_NODE_CHUNK 8192
#endif
constexpr int kNodeChunk = RL_NODE_CHUNK;     // largest chunk of a child-node histogram
constexpr int kMinChunk = 512;       // smallest chunk a (small) node is cut into
#ifndef RL_HIST_DOCS
#define RL_HIST_DOCS 2
#endif
constexpr int kHistDocs = RL_HIST_DOCS;         // samples per thread and iteration of the histogram kernel
#ifndef RL_LOOKBACK2
#define RL_LOOKBACK2 1
#endif
#ifndef RL_HIST_PREFETCH
#define RL_HIST_PREFETCH 0
#endif
constexpr int kHistFG = 16;          // features per group of the histogram layout gbins[group][doc][kHistFG]
#ifndef RL_PART_TILE
#define RL_PART_TILE 2048
#endif
constexpr int kPartTile = RL_PART_TILE;      // docs per partition tile (256 threads x 8)
constexpr int kMaxBins = 4096;       // bin stride limit (thresholds per feature incl. MAX_VALUE); 8 * bin must fit uint16
constexpr int kHistLdsStride = 264;  // compile-time LDS row stride of the histogram kernels when TS <= 264 (the -tc 256 case: 257)
constexpr int kHistLdsBytes = 64 * 1024;
#ifndef RL_KSPEC
#define RL_KSPEC 4
#endif
constexpr int kSpec = RL_KSPEC;      // nodes split (speculatively, in queue order) per growth step
constexpr int kLambdaWaveCap = 384;  // docs/query handled by the wave-per-query lambda kernel
constexpr int kLambdaBlockCap = 5000;
#ifndef RL_RANK_SHORT
#define RL_RANK_SHORT 384
#endif
constexpr int kRankShort = RL_RANK_SHORT;      // lists up to this length get a launch of k_rank_wave with less LDS per wavefront (more wavefronts per CU)
constexpr int kLambdaFusedMaxK = 16;  // NDCG@k up to this k uses the LDS-resident fused lambda kernel

// A tree node while the tree is being grown (device resident).
// Mirrors Split + its FeatureHistogram scalars (learning/tree/Split.java:22-38,
// learning/tree/FeatureHistogram.java:36-44).
struct NodeRec {
    int32_t start, count;    // sample range [start, start+count) in index buffer `buf`
    int32_t buf, parent;
    int32_t left, right;     // node ids, -1 while a leaf
    int32_t best_f, best_t;  // best split of THIS node (feature index, threshold index); -1 = none
    double  best_S;          // -1 = no admissible candidate          (FeatureHistogram.java:311)
    double  deviance;        // Split.deviance
    double  sum_response;    // node total of lambda, converted from the exact fixed-point total
    double  sq_response;     // node total of lambda^2
    long long tot_hi; unsigned long long tot_lo;   // exact 128-bit fixed-point sum of lambda
    long long sq;            // exact fixed-point sum of lambda^2
    float   output; int32_t gcount;   // gcount: samples of the node over ALL ranks (== count on one GPU)
    // speculative growth: a node is PREPARED once its samples are partitioned and its children (pl, pr) have their
    // histograms and best splits; it becomes internal (left/right set) only when RegressionTree.fit's loop pops it.
    int32_t pl, pr;          // prepared children, -1 while unprepared
    int32_t fid;             // id in the exported tree (creation order of the COMMITTED splits), -1 = not part of the tree
    int32_t prepared;
    int32_t best_cl, tie;    // cumulative histogram entry at the best split: count ...; tie: other candidates reach best_S exactly --
                             // 1 = only inside the winning feature (an empty-bin plateau), 2 = in several features (see rl_tie.inc)
    unsigned long long ph;   // path hash of the node: keys the seeded feature draw of its split attempt (rl_params.seed)
    long long best_hi; unsigned long long best_lo;   // ... and exact fixed-point sum (= the left child's totals)
};

// one node being split in the current growth step
struct SlotRec {
    int32_t node, left, right, build_left;   // parent, its prepared children, which child is histogrammed from samples
    int32_t pstart, pcount, pbuf, f, t;      // parent's local sample range and split (copied: saves a dependent load)
    int32_t tile0, ntiles;                   // partition tiles [tile0, tile0+ntiles) of this step's grid
    int32_t chunk0, nchunks;                 // histogram chunks [chunk0, chunk0+nchunks) (upper bound) of this step's grid
    int32_t nleft;                           // local size of the left child when known in advance (one GPU), else -1
    int32_t cs, skip_hist;                   // cs: documents per histogram chunk of the built child when the step's chunks were balanced (balance_slots), 0 = chunk_docs' rule;
                                             // skip_hist: the split that fills the leaf budget (k_select2) -- partitioned, no child histogram is accumulated or finished
    long long sq_left;                       // fixed-point sum of lambda^2 over the BUILT child (k_part_scatter; the sibling's is parent - built)
};

struct TreeState {
    int32_t n_nodes, nslots, done, taken;
    int32_t qsize, n_leaves, E, E2;
    int32_t n_splits, error, tiles_total, chunks_total;
    int32_t arrive, epoch, step, tree_seq;    // finish blocks arrived (last one runs select_step); growth step counter (look-back tags);
                                              // select_step calls in this tree; trees started (both mirrored by the host, see Ctx::progress)
    unsigned long long maxabs_bits;   // max |lambda| of the round (bit pattern, monotone for x >= 0)
    long long root_sq;                // sum over all docs of rint(lambda^2 * 2^E2)
    // lazy tie-break (rl_tie.inc): nodes select_step wanted to partition whose exactly tied best split the Java's rounding noise decides; while
    // stall_n > 0 the step has no slots (every growth kernel is a no-op) and the host runs the resolution kernels
    int32_t stall_n, stall_node[kSpec], defer_any, stall_pad[2 + (4 - kSpec % 4) % 4];      // defer_any: the tree holds nodes whose stored threshold awaits the batched tie-break
    SlotRec slot[kSpec];
    int32_t arrive1[kSpec][16];        // first-level arrival counters of k_hist_finish (<= 16 feature groups per slot)
};

// one kept tree of the ensemble, nodes in creation order
struct TreeSlot {
    int32_t n_nodes, pad;
};

struct FeatBest;
struct Ctx {
    int32_t N, Npad, Q, F, TS, L, MAXN, NC, mls, k, maxChunks, nTiles, FG, numFG;   // MAXN = 2L-1 tree nodes, NC = node records incl. speculation
    float lr;
    int32_t rank, n_ranks;
    int32_t sharded;              // a communicator exists (rl_dist_init*), whatever its size: local sample ranges are not known in advance, histograms are all-reduced (rl_dist.inc)
    int32_t Nglobal;              // documents over ALL ranks (== N on one GPU): fixes the lambda^2 exponent, which every rank must share
    int32_t sub_child;            // features per block of the child-node histogram passes (16, or 8 / 4: RLHIP_SUB_CHILD)
    int32_t hist_nt;              // threads per block of the child-node histogram passes (256 / 512 / 1024; launch_hist)
    int32_t node_div, node_min, node_chunk;   // child-node histograms: target chunks per node, smallest / largest chunk (see chunk_docs)
    const int32_t *live_nthr;      // [n_live] thresholds of live[i] (saves k_hist_finish a dependent load)
    int32_t n_live; const int32_t *live;   // unsharded runs: features with more than one distinct value (the others can never split); k_hist_finish
                                  // is launched over these only (a third of the Yahoo-shape columns are empty)
    int32_t limb_words;           // sharded runs: int64 words per bin in the all-reduced histogram: 3 = (sum >> 44, sum & (2^44-1), count),
                                  // 2 = ((sum >> 44) << 32 | count, low limb) when the data set has fewer than 2^25 documents
    const int32_t *fcol;          // [F] feature sampling: the real column whose key feature f carries; bit 31 set on the later runs of a split threshold table
    int32_t fs_on;                // feature sampling is active (fs_size real features are drawn per split attempt)
    int32_t fs_size;              // features a split attempt looks at: F, or (int)(rate * F) with feature sampling (Random Forests)
    unsigned long long seed;      // rl_params.seed
    int32_t tie_on;          // bit 0: lazy Java-order tie-break (rl_tie.inc): no feature sampling, not the strict mode, not RL_FLAG_FIRST_TIE;
                             // bit 1: ties over several features that share one cut are deferred to the end of the tree (one GPU)
    int32_t mart, metric;    // MART leaf rule (learning/tree/MART.java); RL_METRIC_* of the train metric
    long long *dist_buf;     // [kSpec][F*TS*3 + 4] int64 limbs of the histograms being all-reduced (multi-GPU only)
    // static per data set
    const uint16_t *bins;   // [F][Npad]  feature-major: single-column scans (partition)
    const uint16_t *gbins;  // [numFG][Npad][kHistFG]  group-major: one 32-byte row per document and group (histograms)
    const uint16_t *dbins;  // [Npad][numFG][kHistFG]  document-major: ALL groups of a document adjacent (numFG x 32 bytes).  A sparse node's
                            // sample list touches one 128-byte memory line per (document, group) in gbins -- 4 x the bytes it uses -- but only
                            // the document's own ~numFG/4 lines here; the feature-group blocks of one chunk run on one XCD and share them in L2
    // packed rows (rl_kernels_round.inc pbin8_of; only when the threshold tables have at most 257 entries): one byte per bin + a 16-bit mask per
    // (document, group) for bin 256
    int32_t p8, pd_stride;          // 0 = off, 1 = root pass, 2 = child passes too; bytes per document row of pdbins (16 numFG low bytes, then 2 numFG mask bytes, padded to 64)
    const unsigned char *pbins;     // [numFG][Npad][16]  group-major low bytes (root pass)
    const uint16_t *phib;           // [numFG][Npad]      group-major masks
    const unsigned char *pdbins;    // [Npad][pd_stride]  document-major rows (child passes)
    const uint8_t *cr_grp;        // [numFG] 1 = the group's child passes read compact rows (block-uniform choice; the others keep the 32-byte rows)
    const uint4 *crows; int32_t cr_stride;   // compact rows [Npad][cr_stride] of sparse data (k_compact_rows; null = off): the child passes read these instead of dbins
    int32_t dm_gstride;        // groups per document row of dbins: numFG rounded up to a multiple of 4 (rows start on 128-byte lines)
    int32_t dm_root, dm_div;   // document-major rows for the root pass (0 / 1); for a node of cnt samples when cnt * dm_div <= N (0 = never, 1 = every child)
    int32_t sub;            // features of a group handled per histogram block (kHistFG unless the threshold table is huge)
    const int32_t *mode;    // [F] most populated bin of every feature: never accumulated, rebuilt as total - others
    const uint32_t *runs;   // [numFG] bit j: feature 16 g + j comes in runs of equal bins (query-level columns): quad-folded atomics in k_hist<.., RUNS>
    int32_t balance, balance_cap, balance_target, balance_min;        // balance_slots (RLHIP_BALANCE=0: chunk_docs' per-node rule always)
    int32_t any_runs;       // some column does: the RUNS instantiation of k_hist is launched
    int32_t skip_last;      // k_select2: the step of the split that fills the leaf budget skips its child histograms (RLHIP_SKIP_LAST=0: off)
    const float *thr;       // [F][TS]
    const int32_t *nthr;    // [F]
    const int32_t *feature_ids;
    const int32_t *vcol;    // [F] column of the row matrix behind histogram feature f (null = identity): a real feature whose threshold table has more than
                            // 4095 entries is several virtual features (rl_init)
    const float *labels;    // [N]
    const int32_t *qoff;    // [Q+1]
    const double *ideal0, *ideal1;  // [Q] ideal DCG used by swapChange in round 0 / later (NDCGScorer cache quirk)
    const double *disc;     // discount table, >= maxq+2 entries
    // per round
    double *scores, *ndcg_q;
    double2 *lw;             // [N] (lambda, weight) of the round, interleaved: the leaf chains gather both with one 16-byte load
    long long *q, *r;
    int32_t *idx[2];
    long long *ql[2];        // fixed-point lambda in sample-list order, valid for the ranges of BUILT children only (written by the partition
                             // that creates them; the lists themselves carry document ids alone)
    unsigned long long *tile_desc;   // [nTiles] look-back descriptors of the single-pass partition
    unsigned long long *tile_gdesc;  // [nTiles / 64 + kSpec + 1] totals of the groups of 64 tiles (lookback2_exclusive)
    NodeRec *nodes;
    TreeState *st;
    int32_t *queue;
    long long *cum_hi; unsigned long long *cum_lo; int32_t *cum_cnt;   // [MAXN][F][TS] cumulative
    int32_t *cum_cnt_loc;     // sharded runs (round 6): THIS rank's cumulative counts of every live node (k_hist_reduce) -- the local size of a node's left child is
                              // cum_cnt_loc[node][best_f][best_t], so the single-pass partition and exact chunks need no count pass (null: unsharded)
    long long *part_sum; int32_t *part_cnt;                            // [maxChunks][F][TS]
    long long *part_tot;                                               // [maxChunks] sum of q over the chunk's samples
    struct FeatBest *fb;                                               // [kSpec][2][F] per-feature best split of each new node (rl_kernels_round.inc)
    unsigned long long *fb_root;                                       // [2] exact root total of the round
    unsigned long long *fb_sq;                                         // [kSpec] lambda^2 sum of each slot's left child (published by k_hist_finish block (0, slot))
    int32_t *tile_cnt;                                                 // [nTiles]
    long long *tile_sq;                                                // [nTiles] lambda^2 partial of each partition tile's left members
    int32_t *leaf_node, *leaf_start;                                   // [MAXN], [MAXN+1]
    uint16_t *leaf_of;                                                 // [N] leaf (position in the leaf table) of every document, written with the leaf sums' gather (k_chain_prefix): k_score_stream
    int32_t *grow_stats;                                               // [4] cumulative: growth steps, nodes prepared, splits committed, trees
    unsigned long long *grow_docs;                                     // [4] cumulative documents: accumulated into child histograms (the smaller child of every
                                                                       // PREPARED node), partitioned (prepared nodes), left children of COMMITTED splits (what the
                                                                       // Java accumulates: rho of SURVEY.md 8d), committed split nodes (nu)
    float *round_metric;                                               // [n_trees][2]
    // host-visible (pinned, fine-grained) growth progress: tree_seq << 32 | stalled << 31 | deferred ties << 30 | select_step calls in the tree << 1 | done.
    // A HINT only: the host uses it to stop enqueuing growth steps of a finished tree (extra steps are no-ops).
    unsigned long long *progress;
    // sparse-column path of the root pass (rl_csc.inc, BASELINE.json configs[3])
    int32_t sp_on, sp_ngroups;
    const uint8_t *sp_grp;          // [numFG] 1 = the group's root histogram comes from its entry lists: k_hist<true> skips it
    const int32_t *sp_glist;        // [sp_ngroups] the sparse groups
    const uint32_t *sp_ent;         // packed entries (document-in-chunk << 16 | column-in-group << 12 | bin), ordered by (chunk, sparse group, document)
    const int32_t *sp_off;          // [root chunks * sp_ngroups + 1] first entry of (chunk, sparse group)
    // RL_FLAG_JAVA_ORDER (rl_java_order.inc): the f64 histogram RankLib itself would hold -- per-(feature, bin) sums accumulated
    // sequentially in ascending document order, sequential prefix, right = parent - left -- decides the splits
    int32_t java;
    double *jl;              // [N]        lambda in sample-list order of the nodes being built (left children / the root)
    uint16_t *jb;            // [F][Npad]  bins in the same order, feature-major
    double *jbin;            // [kSpec][F][TS] per-bin Java-order sums of each slot's LEFT child (slot 0: the root pass)
    double *jtot;            // [kSpec][2]  sumResponse, sqSumResponse of the same nodes (FeatureHistogram.java:133-137,182-186)
    double *jcum;            // [NC][F][TS] cumulative Java-order sums of every live node
    const uint16_t *jmap, *jinv; const uint32_t *jone;      // k_jhist2: bin -> owning (wavefront, lane), its inverse, single-bin wavefronts (null: k_jhist)
    int32_t *steplog;        // debug (RLHIP_STEPLOG=1, RL_ARR_STEP_LOG): [0] = entries written, then 8-int entries: growth-step slots and committed tied splits
    int32_t trace_tree;      // -DRL_PHASE_CLOCKS builds: the tree (TreeState::tree_seq while it grows) whose growth kernels record their spans (RLHIP_TRACE_TREE)
    long long *trace;        // [64][3][kTraceBlocks][2] entry / exit stamps of the growth kernels' blocks (null unless RLHIP_TRACE_TREE is set)
    long long *clk;          // [64][32] wall-clock stamps (10 ns units) of the finish / select phases of the last 64 growth steps; only
                             // written by builds with -DRL_PHASE_CLOCKS (tools/phase_clocks.py), RL_ARR_PHASE_CLOCKS reads it
};

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

}  // namespace rl

#define RL_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return rl::fail(RL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " at " + \
                                            __FILE__ + ":" + std::to_string(__LINE__));              \
    } while (0)
