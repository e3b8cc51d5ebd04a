#!/usr/bin/env python
"""What the growth steps of a run work on, and which committed splits had an exactly tied best candidate (RLHIP_STEPLOG=1, RL_ARR_STEP_LOG).
  python tools/step_stats.py [shape] [rounds] [skip]
Answers two design questions with numbers: how many growth steps hold only small nodes (a fused one-launch step would serve them), and how
often / on how large a Java-order derivation chain the lazy tie-break (rl_tie.inc) has to run."""
import os
import sys

os.environ["RLHIP_STEPLOG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ranklib_amd import _native as N
from ranklib_amd import synth


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    n_docs, n_feat, kind, _, leaves = synth.SHAPES[shape]
    X, lab, qoff = synth.make_dataset(n_docs, n_feat, kind)
    g = N.Trainer(n_trees=rounds + skip, n_leaves=leaves)
    g.set_train(X, lab, qoff)
    g.init()
    if skip:
        g.boost_rounds_async(skip); g.sync()
    g.array("STEP_LOG")                      # reading empties the log
    g.boost_rounds_async(rounds); g.sync()
    log = g.array("STEP_LOG")
    n = min(int(log[0]), 8192)
    e = log[8:8 + 8 * n].reshape(n, 8)
    steps, ties = e[e[:, 1] == 0], e[e[:, 1] == 1]
    trees = len(np.unique(e[:, 0]))
    print("%s: %d docs, %d rounds logged (%d entries%s)" % (shape, n_docs, trees, len(e), ", LOG FULL" if int(log[0]) > 8192 else ""))
    # per growth step: the largest split node and the documents accumulated
    key = steps[:, 0].astype(np.int64) * 1000 + steps[:, 2]
    uk = np.unique(key)
    pmax = np.array([steps[key == k, 4].max() for k in uk]); bsum = np.array([steps[key == k, 5].sum() for k in uk]); nsl = np.array([np.sum(key == k) for k in uk])
    print("growth steps per tree: %.2f, slots per step: %.2f" % (len(uk) / trees, nsl.mean()))
    for lim in (1024, 2048, 4096, 8192, 16384, 65536, 262144):
        sel = pmax <= lim
        print("  steps whose largest split node has <= %6d documents: %5.1f %% of the steps (%.2f per tree); nodes <= that size: %5.1f %% of the prepared nodes" %
              (lim, 100.0 * sel.mean(), sel.sum() / trees, 100.0 * np.mean(steps[:, 4] <= lim)))
    si = steps[:, 2]
    for k in range(1, int(si.max()) + 1):
        m = si == k
        if m.sum() == 0:
            continue
        print("  step %2d: in %4.0f %% of the trees, slots %.1f, split node median %8d max %8d, accumulated child median %8d" %
              (k, 100.0 * len(np.unique(steps[m, 0])) / trees, m.sum() / max(len(np.unique(steps[m, 0])), 1), np.median(steps[m, 4]), steps[m, 4].max(), np.median(steps[m, 5])))
    ts = g.array("TIE_STATS")
    print("lazy tie-break over the whole run (skipped rounds included): %d resolutions, %d nodes, %d chain nodes, %d chain documents; %.1f ms on the host, "
          "%d speculative segments, %d window misses, %d serial segments" % (int(ts[0]), int(ts[1]), int(ts[2]), int(ts[3]), ts[4] / 1e3, int(ts[5]), int(ts[6]), int(ts[7])))
    print("committed splits with a tied best candidate: %d in %d rounds (%.2f per round)" % (len(ties), trees, len(ties) / max(trees, 1)))
    if len(ties):
        same_cut = np.sum(((ties[:, 2] & 3) == 2) & ((ties[:, 2] & 4) != 0))      # several features, one cut (deferred to the end of the tree)
        ties = ties.copy(); ties[:, 2] &= 3
        need = ties[(ties[:, 2] == 2) | ((ties[:, 3] & 1) == 1)]
        print("  tie kinds: %d plateaus of one feature (%d of them in right children), %d across features (lanes of 32 holding a tied feature: median %d max %d)" %
              (np.sum(ties[:, 2] == 1), np.sum((ties[:, 2] == 1) & ((ties[:, 3] & 1) == 1)), np.sum(ties[:, 2] == 2),
               np.median(ties[ties[:, 2] == 2, 3] >> 1) if np.any(ties[:, 2] == 2) else 0, (ties[ties[:, 2] == 2, 3] >> 1).max() if np.any(ties[:, 2] == 2) else 0))
        print("  across features with every tied candidate cutting off the same (count, exact sum): %d" % same_cut)
        print("  of those the Java's noise decides (several features tie, or a plateau in a right child): %d (%.2f per round)" % (len(need), len(need) / trees))
        for name, col in (("documents of the node", 4), ("largest node of the derivation chain", 5), ("nodes in the chain", 6), ("documents in the chain", 7)):
            if len(need):
                v = need[:, col]
                print("    %-38s median %8d  p90 %8d  max %8d" % (name, np.median(v), np.percentile(v, 90), v.max()))
        if len(need):
            for lim in (4096, 65536, 1 << 20):
                print("    chains with <= %7d documents in total: %5.1f %%" % (lim, 100.0 * np.mean(need[:, 7] <= lim)))


if __name__ == "__main__":
    main()
