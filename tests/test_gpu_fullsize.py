"""-m gpu: BASELINE.json's full sizes (c1, c2): the oracle itself on all host threads (test_fullsize_oracle_parity: a few
seconds of init + ~2 s per round on the GPU box's 256 threads) and size-independent properties.

c1 = MSLR-WEB10K-shape (1.2 M docs), c2 = MSLR-WEB30K-shape (3.77 M docs), both 136 features / 31 leaves.
Checked per round:
  * conservation: every feature's last cumulative root bin is the SAME exact 128-bit sum and equals sum(q);
    root counts end at N; node counts: parent = left + right, leaves partition the training set
  * the exact parallel float chains: every leaf output equals the literal serial Java float running sums of the
    leaf's lambdas / weights (serial C loop from the test infrastructure), same for the per-round metric mean
  * score update: every document moved by exactly lr * (output of the leaf it fell into)
  * determinism: a second run from scratch produces byte-identical trees / scores
"""
import os
import time

import numpy as np
import pytest

import oracle_ffi as O
from ranklib_amd import _native as N
from ranklib_amd import synth
from tree_equiv import assert_equivalent, node_members

pytestmark = pytest.mark.gpu


def run(shape, rounds, keep_arrays):
    n_docs, n_feat, kind, _, n_leaves = synth.SHAPES[shape]
    X, lab, qoff, Q = synth.make_shard(n_docs, n_feat, kind, 0, 1)
    g = N.Trainer(n_trees=rounds, n_leaves=n_leaves)
    g.set_train(X, lab, qoff)
    g.init()
    out = []
    prev = np.zeros(n_docs)
    for r in range(rounds):
        t, tm, _, _ = g.boost_round()
        rec = dict(tree=t.trimmed(), tm=tm, score=g.array("SCORE"))
        if keep_arrays:
            rec.update(lam=g.array("LAMBDA"), w=g.array("WEIGHT"), q=g.array("QUANT"), fixed=g.array("ROOT_SUM_FIXED"),
                       cnt=g.array("ROOT_COUNT"), nb=g.array("NBINS"), ndcg=g.array("NDCG_PER_QUERY"), prev=prev)
        prev = rec["score"]
        out.append(rec)
    return X, lab, qoff, out, g.array("CHAIN_STATS")


@pytest.mark.parametrize("shape,rounds", [("c1", 3), ("c2", 2)])
def test_fullsize_invariants(shape, rounds):
    X, lab, qoff, recs, stats = run(shape, rounds, True)
    n = X.shape[0]
    lr = np.float64(np.float32(0.1))
    for r, rec in enumerate(recs):
        tr = rec["tree"]
        # --- exact conservation of the fixed-point histogram
        tot = int(rec["q"].astype(object).sum())
        for f in range(X.shape[1]):
            T = rec["nb"][f]
            hi, lo = int(rec["fixed"][f, T - 1, 0]), int(rec["fixed"][f, T - 1, 1]) & 0xFFFFFFFFFFFFFFFF
            assert hi * 2 ** 64 + lo == tot, (r, f)
            assert rec["cnt"][f, T - 1] == n
        # --- tree bookkeeping
        internal = np.nonzero(tr["feature"] != -1)[0]
        for i in internal:
            assert tr["count"][i] == tr["count"][tr["left"][i]] + tr["count"][tr["right"][i]]
        leaves = np.nonzero(tr["feature"] == -1)[0]
        assert tr["count"][leaves].sum() == n and tr["count"][0] == n and len(leaves) == 31
        mem = node_members(tr, X)
        for i in range(len(tr["feature"])):
            assert len(mem[i]) == tr["count"][i], (r, i)
        # --- leaf outputs == literal Java float running sums (LambdaMART.java:401-413)
        leaf_of = np.zeros(n, np.int32)
        for i in leaves:
            idx = mem[i].astype(np.int32)
            s1, s2 = O.float_chain(rec["lam"], idx), O.float_chain(rec["w"], idx)
            exp = np.float32(0) if s2 == 0 else np.float32(s1 / s2)
            assert np.float32(tr["output"][i]).view(np.uint32) == exp.view(np.uint32), (r, i, len(idx))
            leaf_of[idx] = i
        # --- score update: modelScores[k] += learningRate * output   (:208)
        exp_scores = rec["prev"] + lr * tr["output"][leaf_of].astype(np.float64)
        assert np.array_equal(exp_scores.view(np.int64), rec["score"].view(np.int64)), r
        # --- per-round metric: float running sum of the per-query NDCG values, / Q in float (:469-483)
        nd = rec["ndcg"]
        assert nd.min() >= 0.0 and nd.max() <= 1.0
        s = O.float_chain(nd)
        assert np.float32(rec["tm"]).view(np.uint32) == np.float32(s / np.float32(len(nd))).view(np.uint32), r
    assert stats[2] == 0 and stats[5] == 0, "serial finish of a float chain was needed: %s" % stats
    print("%s: float chains evaluated %d, window misses repaired %d, metric chains %d (misses %d)" %
          (shape, stats[0], stats[1], stats[3], stats[4]))


def test_fullsize_determinism_c1():
    _, _, _, a, _ = run("c1", 3, False)
    _, _, _, b, _ = run("c1", 3, False)
    for x, y in zip(a, b):
        assert x["tm"] == y["tm"]
        assert np.array_equal(x["score"].view(np.int64), y["score"].view(np.int64))
        for k in ("feature", "threshold", "left", "right", "output", "count"):
            assert np.array_equal(x["tree"][k], y["tree"][k])


@pytest.mark.parametrize("shape,rounds,strict", [("c1", 3, False), ("c2", 3, False), ("c1", 2, True), ("c2", 2, True), ("c3", 2, False)])
def test_fullsize_oracle_parity(shape, rounds, strict):
    """c1 / c2 / c3 (BASELINE.json configs[1..3] at full size; c3 = 473 k x 700 sparse columns: entry lists in the root pass, compact rows in the
    child passes) against the CPU oracle run with RankLib's MyThreadPool work split on every host thread
    (learning/tree/LambdaMART.java:169-272): thresholds, bins, root counts at init; lambda, weight, scores, per-round metric
    bit for bit; trees through tree_equiv, and NO split may store another (feature, threshold) than the oracle's -- with the default flags
    (exact ties are re-decided in the Java's summation order, rl_tie.inc) and in the strict mode (RL_FLAG_JAVA_ORDER)."""
    n_docs, n_feat, kind, _, n_leaves = synth.SHAPES[shape]
    X, lab, qoff, Q = synth.make_shard(n_docs, n_feat, kind, 0, 1)
    t0 = time.time()
    o = O.Oracle(X, lab, qoff, n_trees=rounds, n_leaves=n_leaves, n_threads=os.cpu_count() or 8)
    o.init()
    t_oinit = time.time() - t0
    g = N.Trainer(n_trees=rounds, n_leaves=n_leaves, **({"flags": N.RL_FLAG_JAVA_ORDER} if strict else {}))
    g.set_train(X, lab, qoff)
    g.init()
    nb, thr, bins, cnt = g.array("NBINS"), g.array("THRESHOLDS"), g.array("BINS"), g.array("ROOT_COUNT")
    for f in range(n_feat):
        T = o.n_bins(f)
        assert nb[f] == T, f
        assert np.array_equal(thr[f, :T].view(np.uint32), o.thresholds(f).view(np.uint32)), f
        assert np.array_equal(bins[f].astype(np.int32), o.bins(f)), f
        assert np.array_equal(cnt[f, :T], o.root_count(f)), f
    del bins
    if shape == "c3":
        info = g.array("SPARSE_INFO")
        assert info[0] > 0 and info[4] >= 30, "the sparse paths (root: entry lists, children: compact rows) must be the ones that run at c3: %s" % info.tolist()
    stats = {}
    t_or = 0.0
    for r in range(rounds):
        t0 = time.time()
        to, tmo, _, _ = o.round()
        t_or += time.time() - t0
        tg, tmg, _, _ = g.boost_round()
        assert np.array_equal(g.array("LAMBDA").view(np.int64), o.lambdas().view(np.int64)), r
        assert np.array_equal(g.array("WEIGHT").view(np.int64), o.weights().view(np.int64)), r
        assert_equivalent(to, tg, X, "%s round %d trace %s" % (shape, r, o.split_trace()), stats)
        assert np.float32(tmo).view(np.uint32) == np.float32(tmg).view(np.uint32), r
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r
    print("\n[fullsize parity] %s%s: %d rounds, splits compared %d, tie-resolved differently %d; oracle init %.1f s, %.2f s/round on %d threads"
          % (shape, " (java-order)" if strict else "", rounds, stats.get("splits", 0), stats.get("plateau", 0), t_oinit,
             t_or / rounds, os.cpu_count() or 8))
    print("[fullsize parity] lazy tie-break: %s (resolutions, nodes, chain nodes, chain documents)" % g.array("TIE_STATS").tolist())
    assert stats.get("plateau", 0) == 0


def test_c1_forty_rounds_against_the_oracle():
    """The boosting loop learning/tree/LambdaMART.java:169-272 followed for 40 rounds at BASELINE.json configs[1]'s full size (1.2 M x 136, 31 leaves,
    default flags) against the oracle -- where the driver's own GPU run can see it (tools/long_parity.py follows 1000 rounds, profiles/r05t_* / r06*):
    every round's tree (no split may store another (feature, threshold)), the float train metric and the scores of all documents bit for bit;
    lambda and weight of all documents every tenth round.  Late rounds differ from the first three: chain-like trees, 15+ growth steps, lambdas
    whose magnitudes span many binades."""
    rounds = 40
    n_docs, n_feat, kind, _, n_leaves = synth.SHAPES["c1"]
    X, lab, qoff, Q = synth.make_shard(n_docs, n_feat, kind, 0, 1)
    o = O.Oracle(X, lab, qoff, n_trees=rounds, n_leaves=n_leaves, n_threads=os.cpu_count() or 8)
    o.init()
    g = N.Trainer(n_trees=rounds, n_leaves=n_leaves)
    g.set_train(X, lab, qoff)
    g.init()
    stats = {}
    t_or = 0.0
    for r in range(rounds):
        t0 = time.time()
        to, tmo, _, _ = o.round()
        t_or += time.time() - t0
        tg, tmg, _, _ = g.boost_round()
        if r % 10 == 0 or r == rounds - 1:
            assert np.array_equal(g.array("LAMBDA").view(np.int64), o.lambdas().view(np.int64)), r
            assert np.array_equal(g.array("WEIGHT").view(np.int64), o.weights().view(np.int64)), r
        assert_equivalent(to, tg, X, "c1 round %d" % r, stats)
        assert np.float32(tmo).view(np.uint32) == np.float32(tmg).view(np.uint32), r
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r
    print("\n[c1 x %d rounds] splits compared %d, tie-resolved differently %d; oracle %.2f s/round; growth steps per tree %.1f"
          % (rounds, stats.get("splits", 0), stats.get("plateau", 0), t_or / rounds, float(g.array("GROW_STATS")[0]) / rounds))
    assert stats.get("splits", 0) == 30 * rounds and stats.get("plateau", 0) == 0


def test_ensemble_eval_10k_trees_1m_rows():
    """BASELINE.json configs[4] in the -m gpu suite at 1/100 of its rows: a 10 000-tree, 31-leaf ensemble (100 trained rounds tiled, as bench.py
    --workload infer builds it) scores 1 000 003 device-resident rows (a partial last tile).  The oracle's Ensemble.eval (learning/tree/
    Ensemble.java:110-116, Split.java:115-125) on samples of the first, a middle and the last rows must give the same bits; a second pass over the
    same rows must give the same bits again; rows with NaN and +-Infinity cells (NaN compares false: right child, Split.java:118) included."""
    import torch
    F, L, rounds, nt = 136, 31, 100, 10000
    X, lab, qoff = synth.make_dataset(60000, F, "mslr")
    g = N.Trainer(n_trees=rounds, n_leaves=L)
    g.set_train(X, lab, qoff)
    g.init()
    g.boost_rounds_async(rounds)
    g.sync()
    g.finish()
    trees = [g.get_tree(i).trimmed() for i in range(rounds)]
    text = g.model_text()
    head, body = text.split("<ensemble>\n", 1)
    blocks = [b.split(">", 1)[1] for b in body.rsplit("</ensemble>", 1)[0].split("\t</tree>\n")[:-1]]
    assert len(blocks) == rounds
    text = head + "<ensemble>\n" + "".join("\t<tree id=\"%d\" weight=\"0.1\">%s\t</tree>\n" % (i + 1, blocks[i % rounds]) for i in range(nt)) + "</ensemble>\n"
    m = N.Model(text)
    assert m.num_trees() == nt
    n, stride = 1000003, F + 1
    gen = torch.Generator(device="cuda").manual_seed(7)
    dX = torch.empty((n, stride), dtype=torch.float32, device="cuda")
    for a in range(0, n, 1 << 18):
        b = min(n, a + (1 << 18))
        u = torch.rand((b - a, stride), generator=gen, device="cuda")
        k = torch.arange(stride, device="cuda") % 4
        dX[a:b] = torch.where(k == 1, torch.floor(u * 21.0), torch.where(k == 2, u, torch.where(k == 3, torch.exp(4.0 * u), torch.where(u < 0.7, torch.zeros_like(u), u))))
    dX[:, 0] = 0
    dX[5, 3] = float("nan"); dX[6, 7] = float("inf"); dX[7, 9] = float("-inf"); dX[n - 2, 17] = float("nan")
    dO = torch.zeros(n, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time()
    m.predict_device(dX.data_ptr(), n, stride, dO.data_ptr())
    torch.cuda.synchronize()
    dt = time.time() - t0
    first = dO.cpu().numpy().copy()
    dO.zero_()
    m.predict_device(dX.data_ptr(), n, stride, dO.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(first.view(np.uint32), dO.cpu().numpy().view(np.uint32)), "two passes over the same rows differ"
    all_trees = [trees[i % rounds] for i in range(nt)]
    threads = os.cpu_count() or 8
    for a, b in ((0, 1500), (n // 2 - 700, n // 2 + 700), (n - 1500, n)):
        ref = O.eval_flat_model(all_trees, dX[a:b].cpu().numpy(), n_threads=threads)
        assert np.array_equal(ref.view(np.uint32), first[a:b].view(np.uint32)), (a, b)
    print("\n[infer] %d trees x %d rows: %.2f s first pass (%.1f M docs/s incl. launch), 4 400 sampled rows equal the oracle's Ensemble.eval" % (nt, n, dt, n / dt / 1e6))
