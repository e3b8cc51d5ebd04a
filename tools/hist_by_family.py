#!/usr/bin/env python
"""GPU box: root / child histogram time when ALL 136 columns come from ONE of the four synthetic feature families
(ranklib_amd/synth.py: 0 = counts 0..20, 1 = continuous, 2 = heavy-tailed, 3 = 70 % zeros): which bin-occupancy shapes cost the LDS atomics most."""
import os
import sys
import numpy as np
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R)
from ranklib_amd import _native as N, synth  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
X, lab, qoff, Q = synth.make_shard(n, 136, 'mslr', 0, 1)
for fam in (-1, 0, 1, 2, 3):
    Xf = X if fam < 0 else np.ascontiguousarray(X[:, [fam + 4 * (j % 34) for j in range(136)]])
    g = N.Trainer(n_trees=12, n_leaves=31, flags=N.RL_FLAG_TIMING | N.RL_FLAG_TIMING_NODES)
    g.set_train(Xf, lab, qoff); g.init()
    g.boost_rounds_async(2); g.sync(); g.reset_timing()
    gd0 = g.array("GROW_DOCS").astype(np.float64)
    g.boost_rounds_async(10); g.sync()
    built = (g.array("GROW_DOCS").astype(np.float64) - gd0)[0] / 10
    r_ms, r_n, _ = g.timing("HIST_ROOT"); n_ms, n_n, _ = g.timing("HIST_NODE")
    print("family %2d: root %.3f ms (%.2f G pairs/s), nodes %.3f ms per round for %.2f M docs (%.2f G pairs/s)" %
          (fam, r_ms / r_n, n * 136 / (r_ms / r_n) / 1e6, n_ms / 10, built / 1e6, built * 136 / (n_ms / 10) / 1e6), flush=True)
    g.close()
