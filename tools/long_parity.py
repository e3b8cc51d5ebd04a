#!/usr/bin/env python
"""The DEFAULT path followed for the whole length BASELINE.json states, against the CPU oracle (test infrastructure: the checker, never the product).

Trains a BASELINE.json shape (c1 x 1000 trees = configs[1] as stated; c2 x 300 rounds) with default flags on the GPU and with the oracle side by
side -- the boosting loop learning/tree/LambdaMART.java:169-272 -- with a HELD-OUT set (a fifth of the training set's size, the same generator
continued past the training documents, labelled with the training set's cuts) passed through rl_set_validation (LambdaMART.java:228-250).
Every round: the tree (tree_equiv: identical, or equivalent with an exact tie resolved differently), the scores of every training document (bits),
the float train metric and the float validation metric (bits); every 10th round lambda and weight of every document (bits).  The first round whose
tree / scores / metrics differ is classified with tools/parity_classify.py and the comparison stops there; both sides then run to the end on their
own and the final train and held-out NDCG@10 of both are printed (MetricScorer.score over Ensemble.eval, LambdaMART.java:259, eval/Evaluator.java:669-708).

usage (GPU box; ~0.7 s a round at c1 and ~2.2 s at c2 on the box's host threads, i.e. 12 + 11 minutes):
    python tools/long_parity.py c1 1000
    python tools/long_parity.py c2 300
The last line is one JSON object; the c1 x 1000 one is committed as profiles/r05_long_parity_c1.json and read by bench.py (config.c1_heldout_run)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_ffi as O  # noqa: E402
from parity_classify import classify  # noqa: E402
from ranklib_amd import _native as N, synth  # noqa: E402
from tree_equiv import assert_equivalent  # noqa: E402


held_out = synth.make_heldout


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "c1"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else synth.SHAPES[shape][3]
    estop = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30      # (-estop: the run is followed over its whole length; the rollback at the end stays)
    lam_every = int(os.environ.get("LONG_PARITY_LAMBDA_EVERY", "10"))
    n_docs, n_feat, kind, _, leaves = synth.SHAPES[shape]
    threads = os.cpu_count() or 8
    X, lab, qoff, _ = synth.make_shard(n_docs, n_feat, kind, 0, 1)
    Xv, labv, qv = held_out(shape)
    t0 = time.time()
    o = O.Oracle(X, lab, qoff, n_trees=rounds, n_leaves=leaves, n_threads=threads, early_stop=estop)
    o.set_validation(Xv, labv, qv)
    o.init()
    g = N.Trainer(n_trees=rounds, n_leaves=leaves, early_stop_rounds=estop)
    g.set_train(X, lab, qoff)
    g.set_validation(Xv, labv, qv)
    g.init()
    print("[long parity] %s: %d x %d training documents, %d held-out documents in %d lists, %d rounds, %d leaves, default flags; oracle on %d threads; init %.1f s"
          % (shape, n_docs, n_feat, len(labv), len(qv) - 1, rounds, leaves, threads, time.time() - t0), flush=True)
    stats = {}
    first_div, why = None, None
    compared = 0
    t_o = t_g = 0.0
    tl = time.time()
    for r in range(rounds):
        t1 = time.time()
        to, tmo, vmo, so = o.round()
        t2 = time.time()
        tg, tmg, vmg, sg = g.boost_round()
        t3 = time.time()
        t_o += t2 - t1; t_g += t3 - t2
        if first_div is None:
            what = None
            try:
                assert_equivalent(to, tg, X, "%s round %d" % (shape, r), stats)
            except AssertionError as ex:
                what = "tree: %s" % (str(ex)[:300],)
            if what is None:
                so_, sg_ = o.scores(), g.array("SCORE")
                if not np.array_equal(so_.view(np.int64), sg_.view(np.int64)):
                    what = "scores: %d documents differ, max |d| %.3g" % (int((so_ != sg_).sum()), float(np.abs(so_ - sg_).max()))
                elif np.float32(tmo).view(np.uint32) != np.float32(tmg).view(np.uint32):
                    what = "train metric %r vs %r" % (float(tmo), float(tmg))
                elif np.float32(vmo).view(np.uint32) != np.float32(vmg).view(np.uint32):
                    what = "validation metric %r vs %r" % (float(vmo), float(vmg))
                elif r % lam_every == 0 or r == rounds - 1:
                    if not (np.array_equal(o.lambdas().view(np.int64), g.array("LAMBDA").view(np.int64)) and
                            np.array_equal(o.weights().view(np.int64), g.array("WEIGHT").view(np.int64))):
                        what = "lambda / weight"
            if what is None:
                compared = r + 1
            else:
                first_div = r
                why = None
                if what.startswith("tree"):
                    try:
                        why = classify(to, tg, X, o.lambdas(), False)
                    except Exception as ex:       # noqa: BLE001
                        why = "classification failed: %r" % (ex,)
                    if why is None:
                        why = "UNEXPLAINED %s" % (getattr(classify, "last", None),)
                print("[long parity] %s: FIRST DIVERGENT ROUND %d -- %s -- classified: %s" % (shape, r, what, why), flush=True)
                print("[long parity] oracle split trace of that round: %s" % (o.split_trace(),), flush=True)
        if (r + 1) % 50 == 0 or r == rounds - 1:
            now = time.time()
            print("[long parity] round %4d: train %.4f / %.4f  held-out %.4f / %.4f (oracle / gpu); %s; %.2f s a round (oracle %.2f, gpu+checks %.3f); tie-break %s"
                  % (r + 1, tmo, tmg, vmo, vmg, "identical so far, %d splits compared, %d stored another (feature, threshold)" % (stats.get("splits", 0), stats.get("plateau", 0))
                     if first_div is None else "diverged at round %d" % first_div, (now - tl) / (50 if (r + 1) % 50 == 0 else max(1, (r + 1) % 50)), t_o / (r + 1), t_g / (r + 1),
                     g.array("TIE_STATS").tolist()[:4]), flush=True)
            tl = now
        if so or sg:
            print("[long parity] early stop at round %d (oracle %s, gpu %s)" % (r, so, sg), flush=True)
            break
    fo_t, fo_v = o.finish()
    fg_t, fg_v = g.finish()
    res = dict(shape=shape, rounds=rounds, rounds_compared_identical=compared, first_divergent_round=first_div, divergence=why,
               splits_compared=stats.get("splits", 0), splits_storing_another_candidate=stats.get("plateau", 0),
               ndcg10_train_oracle=fo_t, ndcg10_train_gpu=fg_t, ndcg10_heldout_oracle=fo_v, ndcg10_heldout_gpu=fg_v,
               trees_kept_oracle=o.trees_kept(), trees_kept_gpu=g.num_trees(), best_validation_round_oracle=o.best_valid()[0], best_validation_round_gpu=g.best_validation()[0],
               oracle_threads=threads, oracle_s_per_round=round(t_o / max(1, r + 1), 3), tie_stats=g.array("TIE_STATS").tolist())
    print("[long parity] %s: final NDCG@10 train %.6f / %.6f, held-out %.6f / %.6f (oracle / gpu): |d| %.2g / %.2g; trees kept %d / %d"
          % (shape, fo_t, fg_t, fo_v, fg_v, abs(fo_t - fg_t), abs(fo_v - fg_v), res["trees_kept_oracle"], res["trees_kept_gpu"]), flush=True)
    print(json.dumps(res))
    return 0 if (abs(fo_t - fg_t) <= 1e-4 and abs(fo_v - fg_v) <= 1e-4) else 1


if __name__ == "__main__":
    sys.exit(main())
