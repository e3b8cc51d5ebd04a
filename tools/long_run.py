#!/usr/bin/env python
"""Long training run on the GPU: 1000 boosting rounds at a BASELINE.json shape, chain / growth statistics and the metric curve.
usage (GPU box): python tools/long_run.py [shape] [rounds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ranklib_amd import _native as N, synth  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "c1"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n_docs, n_feat, kind, _, leaves = synth.SHAPES[shape]
X, lab, qoff, _ = synth.make_shard(n_docs, n_feat, kind, 0, 1)
g = N.Trainer(n_trees=rounds, n_leaves=leaves)
g.set_train(X, lab, qoff)
g.init()
t0 = time.perf_counter()
tl = t0
done = 0
while done < rounds:
    k = min(100, rounds - done)
    g.boost_rounds_async(k)
    g.sync()
    done += k
    now = time.perf_counter()
    print("round %4d  %.1f rounds/s  train metric %.4f  chain stats %s  grow stats %s" % (done, k / (now - tl), g.round_metrics(done - 1)[0],
                                                                                       g.array("CHAIN_STATS").tolist(), g.array("GROW_STATS").tolist()), flush=True)
    tl = time.perf_counter()
el = time.perf_counter() - t0
final, _ = g.finish()
m = [g.round_metrics(i)[0] for i in range(rounds)]
print("%s: %d rounds in %.2f s = %.1f rounds/s; final metric %.4f; metric monotone-ish: first %.4f last %.4f min %.4f" %
      (shape, rounds, el, rounds / el, final, m[0], m[-1], min(m)))
assert np.isfinite(m).all() and np.isfinite(g.array("SCORE")).all()
