#!/usr/bin/env python
"""Random small configurations, GPU against the oracle: rankers x metrics x cut-offs x leaves x min leaf support x threshold
candidates x feature sampling x list-length mixes x validation.  usage (GPU box): python tools/fuzz_parity.py [n_cases] [seed] [scale]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (initialise torch's HIP runtime first, see tests/conftest.py)
if torch.cuda.is_available():
    torch.cuda.init()
from fractions import Fraction  # noqa: E402

import oracle_ffi as O  # noqa: E402
from ranklib_amd import _native as N  # noqa: E402
from tree_equiv import assert_equivalent, node_members  # noqa: E402


sys.path.insert(0, os.path.join(ROOT, "tools"))
from parity_classify import classify  # noqa: E402


JAVA = bool(os.environ.get("FUZZ_JAVA"))      # RL_FLAG_JAVA_ORDER: trees must be IDENTICAL to the oracle's (no tie may be resolved differently)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # multiplies the number of lists (bigger data: several chunks / tiles per node)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = ties = skipped = 0
other_splits = compared_splits = 0
big_other = big_compared = big_cases = 0   # the same counts in cases where a threshold table has more than 4095 entries (virtual features: exact ties keep the first candidate)
                                          # splits that store another (feature, threshold) than the oracle's while the trees stay equivalent (exact-tie plateaus): 0 with
reasons = {}                              # the lazy Java-order tie-break (rl_tie.inc) unless features are sampled
for case in range(n_cases):
    F = int(rng.choice([3, 8, 17, 40]))
    kind = rng.choice(["tiny", "mixed", "long"])
    if kind == "tiny":
        sizes = rng.integers(1, 17, int(rng.integers(20, 400)))
    elif kind == "mixed":
        sizes = np.concatenate([rng.integers(1, 17, 100), rng.integers(17, 200, 30), rng.integers(200, 500, 3)])
    else:
        sizes = rng.integers(100, 700, int(rng.integers(3, 12)))
    if scale > 1:
        sizes = np.concatenate([sizes] * scale)
    rng.shuffle(sizes)
    qoff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    n = int(qoff[-1])
    X = rng.random((n, F)).astype(np.float32)
    X[:, ::3] = np.floor(X[:, ::3] * rng.integers(2, 30))                  # low-cardinality columns, ties
    if F > 8:
        X[:, 5] = 0.0                                                       # a dead column
    if rng.random() < 0.3:                                                  # query-level columns (one value per list): k_hist<.., RUNS>
        for f in rng.choice(F, size=min(F, int(rng.integers(1, 4))), replace=False):
            vals = rng.random(len(sizes)).astype(np.float32) if rng.random() < 0.5 else np.floor(rng.random(len(sizes)) * 4).astype(np.float32)
            X[:, f] = np.repeat(vals, sizes)
    z = X[:, 0] * 0.3 + X[:, 1 % F] * X[:, 2 % F] + 0.5 * rng.random(n)
    lab = np.floor(np.clip(z / z.max() * 5, 0, 4)).astype(np.float32)
    ranker = rng.choice(["LAMBDAMART", "LAMBDAMART", "MART"])
    metric = rng.choice(["NDCG", "NDCG", "DCG", "MAP", "ERR"])
    k = int(rng.choice([1, 3, 10, 16, 25])) if metric != "MAP" else int(rng.choice([0, 5]))
    leaves = int(rng.choice([2, 3, 7, 10, 31, 64]))
    mls = int(rng.choice([1, 1, 5, 50]))
    tc = int(rng.choice([256, 256, 10, -1]))
    frate = float(rng.choice([1.0, 1.0, 0.5, 0.3]))
    lr = float(rng.choice([0.1, 0.05, 1.0]))
    rounds = int(rng.integers(2, 6))
    seed = int(rng.integers(0, 2 ** 31))
    with_valid = bool(rng.random() < 0.3) and not os.environ.get("FUZZ_DIST")      # a validation set + early stopping
    estop = int(rng.choice([1, 2, 100]))
    if with_valid:
        vs = rng.integers(1, 60, int(rng.integers(5, 60)))
        vqoff = np.concatenate([[0], np.cumsum(vs)]).astype(np.int32)
        Xv = rng.random((int(vqoff[-1]), F)).astype(np.float32)
        Xv[:, ::3] = np.floor(Xv[:, ::3] * 7)
        labv = np.floor(rng.random(int(vqoff[-1])) * 5).astype(np.float32)
    if os.environ.get("FUZZ_ONLY") and str(case) not in os.environ["FUZZ_ONLY"].split(","):
        continue                                   # (the random stream has been drawn: the listed cases see the data they saw in the full run)
    desc = dict(case=case, n=n, F=F, kind=str(kind), ranker=str(ranker), metric=str(metric), k=k, leaves=leaves, mls=mls, tc=tc, frate=frate, lr=lr, rounds=rounds, valid=with_valid, estop=estop)
    big_case = False
    try:
        o = O.Oracle(X, lab, qoff, n_trees=rounds, n_leaves=leaves, lr=lr, n_threshold=tc, mls=mls, k=k, ranker=str(ranker), metric=str(metric),
                     n_threads=3, frate=frate, seed=seed, early_stop=estop)
        g = N.Trainer(n_trees=rounds, n_leaves=leaves, learning_rate=lr, n_threshold=tc, min_leaf_support=mls, metric_k=k, metric=str(metric),
                      ranker=str(ranker), feature_sampling_rate=frate, seed=seed, early_stop_rounds=estop,
                      flags=N.RL_FLAG_JAVA_ORDER if JAVA else 0)
        g.set_train(X, lab, qoff)
        if with_valid:
            o.set_validation(Xv, labv, vqoff); g.set_validation(Xv, labv, vqoff)
        if os.environ.get("FUZZ_DIST"):        # the sharded code path (count + scatter, limb reduce, finish<.,true>, gathered chains) with one rank
            g.dist_init_callback(0, 1, lambda arr, op: None, lambda src: src.copy())
        o.init(); g.init()
        big_case = g.hist_features()[0] != F        # a threshold table beyond 4095 entries: histogrammed as runs (rl_init), first candidate wins an exact tie
        big_cases += 1 if big_case else 0
        same_trees = True                 # validation rows may take another branch at a tie-resolved (equivalent, not identical) split
        ended = False                     # left the case at a tie / a diverged run
        for m in range(rounds):
            to, tmo, vmo, so_ = o.round()
            tg, tmg, vmg, sg_ = g.boost_round()
            lam = o.lambdas().copy()
            if not np.isfinite(o.scores()).all() or not np.isfinite(lam).all():
                skipped += 1                      # the run has diverged to infinite scores: out of contract
                ended = True
                break
            assert np.array_equal(g.array("LAMBDA"), lam), "lambda, round %d" % m
            try:
                nt = assert_equivalent(to, tg, X, ctx="round %d" % m)
                if frate >= 1.0 and not os.environ.get("FUZZ_DIST"):
                    nsp = int((to.trimmed()["feature"] != -1).sum())
                    if big_case: big_other += nt; big_compared += nsp
                    else: other_splits += nt; compared_splits += nsp
                    if nt and os.environ.get("FUZZ_VERBOSE"):
                        print("  case %d round %d: %d split(s) store another (feature, threshold) %s" % (case, m, nt, desc), flush=True)
                        a_, b_ = to.trimmed(), tg.trimmed()
                        for i_ in range(len(a_["feature"])):
                            if a_["feature"][i_] != b_["feature"][i_] or a_["threshold"][i_] != b_["threshold"][i_]:
                                print("    node %d (count %d): oracle (f %d, %r) gpu (f %d, %r)" % (i_, a_["count"][i_], a_["feature"][i_], float(a_["threshold"][i_]), b_["feature"][i_], float(b_["threshold"][i_])), flush=True)
                                if os.environ.get("FUZZ_ONLY"):
                                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                                    from tie_diag import java_restatement
                                    java_restatement(a_, node_members(a_, X), i_, lam, g.array("BINS"), g.array("NBINS"), g.array("THRESHOLDS"), mls)
                        print("    TIE_STATS", g.array("TIE_STATS").tolist(), flush=True)
                if JAVA:
                    assert nt == 0, "RL_FLAG_JAVA_ORDER: %d splits store another (feature, threshold) than the oracle's, round %d" % (nt, m)
            except AssertionError:
                if JAVA:
                    raise
                why = classify(to, tg, X, lam, frate < 1.0)
                if why:
                    ties += 1
                    reasons[why] = reasons.get(why, 0) + 1
                    if frate >= 1.0 and os.environ.get("FUZZ_VERBOSE"):
                        print("  case %d ended in round %d: %s %s" % (case, m, why, desc), flush=True)
                    ended = True
                    break
                print("  unexplained:", getattr(classify, "last", None), flush=True)
                raise
            assert np.array_equal(g.array("SCORE"), o.scores()), "scores, round %d" % m
            assert np.float32(tmg) == np.float32(tmo), "metric, round %d" % m
            ta, tb = to.trimmed(), tg.trimmed()
            same_trees = same_trees and np.array_equal(ta["feature"], tb["feature"]) and np.array_equal(ta["threshold"].view(np.uint32), tb["threshold"].view(np.uint32))
            if with_valid and same_trees:
                assert np.float32(vmg) == np.float32(vmo), "validation metric, round %d" % m
                assert so_ == sg_, "early stop flag, round %d" % m
            if so_ or sg_:
                break                             # early stop (LambdaMART.java:248)
        if not ended:
            so, vo = o.finish(); sg, vg = g.finish()
            if not with_valid or same_trees:
                assert so == sg, "final metric"
                if with_valid:
                    assert vo == vg and o.trees_kept() == g.num_trees(), "validation: final metric / kept trees (rollback)"
            # the saved model, loaded back and scored by the inference kernel == Ensemble.eval of the oracle's trees on the training rows
            if not with_valid or same_trees:      # (with equivalent-but-not-identical trees the validation rows may pick another best round)
                mdl = N.Model(g.model_text())
                rows = np.concatenate([np.zeros((n, 1), np.float32), X], axis=1)
                got, want = mdl.predict_rows(rows), o.predict(X)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "model scores"
                mdl.close()
    except N.RankLibError as ex:
        if "rlhip status -4" in str(ex):          # a documented limit (a threshold table beyond 4095 entries together with feature sampling, sharding or the strict mode)
            skipped += 1
        else:
            bad += 1
            print("MISMATCH", desc, "->", repr(ex)[:300], flush=True)
    except Exception as ex:       # noqa: BLE001
        bad += 1
        print("MISMATCH", desc, "->", repr(ex)[:300], flush=True)
print("%d cases: %d mismatches, %d ended at a tie %s, %d skipped (documented limits); without feature sampling %d of %d compared splits store another "
      "(feature, threshold) than the oracle's; %d cases with a threshold table of more than 4095 entries (first tied candidate wins there): %d of %d"
      % (n_cases, bad, ties, reasons, skipped, other_splits, compared_splits, big_cases, big_other, big_compared))
sys.exit(1 if bad else 0)
