"""-m gpu: the lazy Java-order tie-break of the default path (ranklib_amd/csrc/rl_tie.inc, HISTORY.md 4.13).

The parity suites already demand the oracle's stored (feature, threshold) pairs everywhere; these tests aim at the machinery itself: the
speculative evaluation of long chains against the literal walk, the deferred batch, the first-candidate flag, and the statistics array."""
import numpy as np
import pytest

import oracle_ffi as O
from ranklib_amd import _native as N
from ranklib_amd import synth
from tree_equiv import assert_equivalent

pytestmark = pytest.mark.gpu


def run(X, lab, qoff, rounds, leaves, flags=0, oracle=True, **kw):
    g = N.Trainer(n_trees=rounds, n_leaves=leaves, flags=flags, **kw)
    g.set_train(X, lab, qoff)
    g.init()
    o = None
    if oracle:
        o = O.Oracle(X, lab, qoff, n_trees=rounds, n_leaves=leaves, n_threads=16, mls=kw.get("min_leaf_support", 1))
        o.init()
    trees, ties = [], 0
    for r in range(rounds):
        tg, tmg, _, _ = g.boost_round()
        trees.append(tg.trimmed())
        if o is not None:
            to, tmo, _, _ = o.round()
            ties += assert_equivalent(to, tg, X, "round %d" % r)
            assert np.float32(tmo).view(np.uint32) == np.float32(tmg).view(np.uint32), r
    return trees, ties, g.array("TIE_STATS"), g.array("SCORE")


def same_trees(a, b):
    for x, y in zip(a, b):
        for k in ("feature", "left", "right", "count"):
            assert np.array_equal(x[k], y[k]), k
        assert np.array_equal(x["threshold"].view(np.uint32), y["threshold"].view(np.uint32))
        assert np.array_equal(x["output"].view(np.uint32), y["output"].view(np.uint32))


def test_long_chains_speculative_evaluation_equals_the_literal_walk_and_the_oracle(monkeypatch):
    """150 k documents, deep trees with a large minimum leaf support removed: derivation chains of 10^5 documents are cut into many chunks
    (windows of 2048 values), run from 256 candidate states, stitched -- and must give what the literal walk and the oracle give"""
    X, lab, qoff = synth.make_dataset(150000, 12, "mslr", seed_offset=11)
    rounds, leaves = 10, 48
    trees, ties, st, sc = run(X, lab, qoff, rounds, leaves)
    assert ties == 0
    assert st[0] > 0 and st[3] > 200000 and st[5] > 100, st         # resolutions ran, on chains of > 10^5 documents, through many speculative chunks
    monkeypatch.setenv("RLHIP_TIE_WALK", "1")
    trees_w, _, st_w, sc_w = run(X, lab, qoff, rounds, leaves, oracle=False)
    assert st_w[5] == 0 and st_w[0] == st[0]                         # the same resolutions, none of them speculative
    same_trees(trees, trees_w)
    assert np.array_equal(sc.view(np.int64), sc_w.view(np.int64))


def test_first_tie_flag_keeps_the_first_candidate_and_the_same_function():
    """RL_FLAG_FIRST_TIE: no resolution runs; the trees are the same function on the training set (equivalent, scores bit-identical) but some stored
    (feature, threshold) pairs differ from the oracle's -- which is exactly what the lazy tie-break is for"""
    X, lab, qoff = synth.make_dataset(2500, 5, "mslr", seed_offset=4)
    trees, ties, st, sc = run(X, lab, qoff, 4, 31)
    assert ties == 0 and st[0] > 0 and st[1] >= st[0]
    trees_f, ties_f, st_f, sc_f = run(X, lab, qoff, 4, 31, flags=N.RL_FLAG_FIRST_TIE)
    assert st_f[0] == 0 and ties_f > 0
    assert np.array_equal(sc.view(np.int64), sc_f.view(np.int64))


def test_deferred_plateau_ties_are_decided_in_one_batch_per_tree():
    """ties whose candidates all cut off the same documents do not stall the tree: fewer resolutions than nodes, and with the deferral defeated by
    the environment switch of the walk (every tie then stalls as it is met? no -- the switch only changes HOW the sums are made) the same trees"""
    X, lab, qoff = synth.make_dataset(4000, 6, "mslr", seed_offset=9)
    trees, ties, st, _ = run(X, lab, qoff, 6, 31)
    assert ties == 0
    assert st[1] > st[0] > 0, st                                      # batches: more nodes than calls


def test_mirrored_cut_of_another_feature_is_a_tie():
    """feature 2 = -feature 1: every cut of one is the mirrored cut of the other (equal true gain, f64 S equal only up to the rounding of
    total - sumLeft).  The partition key sees the tie, the Java-order evaluation decides it as the oracle does"""
    X, lab, qoff = synth.make_dataset(3000, 4, "ns", seed_offset=21)
    X = X.copy()
    X[:, 1] = -X[:, 0]
    trees, ties, st, _ = run(X, lab, qoff, 5, 16)
    assert ties == 0 and st[0] > 0


def test_huge_threshold_tables_take_the_walk():
    """-tc -1 with thousands of distinct values: the stable sort's cursors would not fit the LDS, the resolution uses the literal walk"""
    X, lab, qoff = synth.make_dataset(3000, 6, "ns", seed_offset=3)
    g = N.Trainer(n_trees=3, n_leaves=48, n_threshold=-1, ranker="MART")
    g.set_train(X, lab, qoff); g.init()
    o = O.Oracle(X, lab, qoff, n_trees=3, n_leaves=48, n_threshold=-1, ranker="MART"); o.init()
    for r in range(3):
        tg, _, _, _ = g.boost_round(); to, _, _, _ = o.round()
        assert assert_equivalent(to, tg, X, "round %d" % r) == 0
    st = g.array("TIE_STATS")
    assert st[0] > 0 and st[5] == 0, st


def duplicate_columns(seed, n_docs=6000, n_feat=8):
    """columns 4.. repeat columns 0..3 (scaled: other thresholds, the same cuts): ties over several features whose candidates all cut a node the same way"""
    X, lab, qoff = synth.make_dataset(n_docs, n_feat, "mslr", seed_offset=seed)
    X = X.copy()
    for j in range(4, n_feat):
        X[:, j] = 2.0 * X[:, j - 4] + 1.0
    return X, lab, qoff


def test_ties_over_several_features_that_share_one_cut_are_deferred_not_stalled(monkeypatch):
    """duplicated columns: every best split ties across two features over one and the same cut.  The tree is not stalled on them: they join the per-tree
    batch (k_tie_verify checks the cuts document by document), the oracle's feature is stored, and the switch that stalls on them instead gives the
    same trees with many more resolutions"""
    X, lab, qoff = duplicate_columns(31)
    trees, ties, st, sc = run(X, lab, qoff, 5, 31)
    assert ties == 0
    assert st[8] > 0 and st[9] == 0, st                                # batches at the end of trees, no tree grown twice
    assert st[1] > 5 * st[0], st                                       # dozens of nodes per batch
    monkeypatch.setenv("RLHIP_TIE_NO_XDEFER", "1")
    trees_s, ties_s, st_s, sc_s = run(X, lab, qoff, 5, 31)
    assert ties_s == 0 and st_s[0] > 3 * st[0], (st, st_s)
    same_trees(trees, trees_s)
    assert np.array_equal(sc.view(np.int64), sc_s.view(np.int64))


def test_a_tree_whose_deferred_tie_fails_the_check_is_grown_again(monkeypatch):
    """the verification's miss path, forced: every batch with a tie over several features reports another cut, the tree is grown again from its root
    histogram with the shortcut off (stalls instead) -- same trees, same scores, the counter says how often"""
    X, lab, qoff = duplicate_columns(32)
    trees, ties, st, sc = run(X, lab, qoff, 4, 24)
    assert ties == 0 and st[9] == 0
    monkeypatch.setenv("RLHIP_TIE_FORCE_REGROW", "1")
    trees_r, ties_r, st_r, sc_r = run(X, lab, qoff, 4, 24)
    assert ties_r == 0 and st_r[9] > 0, st_r
    same_trees(trees, trees_r)
    assert np.array_equal(sc.view(np.int64), sc_r.view(np.int64))


def test_equal_lambdas_on_opposite_sides_are_caught_by_the_check():
    """two candidates of two features with equal (left count, exact left sum) that are NOT the same documents: lists whose documents all carry one label
    have lambda = 0, and two such documents swapped between the sides leave count and sum unchanged.  Built on purpose: feature 0 takes two values and
    carries the labels, feature 1 repeats it with two zero-lambda documents swapped -- the root's best split ties over the two features with
    different cuts.  The device defers it (equal keys), the check finds the two documents, the tree is grown again with a stall, and the stored trees
    are the oracle's"""
    rng = np.random.default_rng(5)
    n_q, per = 60, 20
    n = n_q * per
    qoff = np.arange(0, n + 1, per, dtype=np.int32)
    right = rng.random(n) < 0.5
    lab = np.where(right, rng.integers(1, 4, n), rng.integers(0, 2, n)).astype(np.float32)
    lab[:2 * per] = 1.0                                               # two single-label lists: every lambda in them is exactly 0
    a, b = 3, per + 5
    right[a], right[b] = False, True
    X = (rng.standard_normal((n, 4)) * 0.01).astype(np.float32)
    X[:, 0] = np.where(right, -0.75, -2.0)
    X[:, 1] = X[:, 0]
    X[a, 1], X[b, 1] = X[b, 0], X[a, 0]                               # the copy disagrees on the two zero-lambda documents
    trees, ties, st, _ = run(X, lab, qoff, 3, 8)
    assert ties == 0, st
    assert st[9] > 0, st                                              # at least the first tree was grown twice


@pytest.mark.parametrize("leaves,mls,n_docs,n_feat", [(1200, 1, 60000, 17), (-1, 4, 9000, 24)])
def test_wide_trees_resolve_every_deferred_tie_without_a_limit(leaves, mls, n_docs, n_feat):
    """Trees of a thousand leaves (and -leaf -1 with a small minimum leaf support): the deferred batch at the end of a tree holds hundreds of right
    children with plateau ties, i.e. hundreds of chain nodes and (chain node, feature) pairs in ONE resolution.  The pinned read-back buffer grows
    with the batch (it used to be sized for 512 chain nodes and the round failed with RL_ERR_UNSUPPORTED beyond: ADVICE r03); the stored
    (feature, threshold) pairs are the oracle's."""
    X, lab, qoff = synth.make_dataset(n_docs, n_feat, "mslr", seed_offset=23)
    X = X.copy()
    X[:, ::2] = np.floor(X[:, ::2] * 6.0)            # low-cardinality columns: small nodes tie all the time
    trees, ties, st, _ = run(X, lab, qoff, 2, leaves, min_leaf_support=mls)
    assert ties == 0
    n_leaves = int((trees[0]["feature"] == -1).sum())
    assert n_leaves >= 600, n_leaves
    assert st[0] > 0 and st[2] > 512, st              # resolutions ran, over more chain nodes than the old fixed capacity
