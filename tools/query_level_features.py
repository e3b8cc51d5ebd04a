"""What query-level features cost the histogram kernels: MSLR's query-dependent columns (query length, IDF sums, ...) hold ONE value for all
documents of a query, so the 64 consecutive documents a wavefront accumulates hit the SAME bin of such a column -- a 64-way same-address LDS
atomic unless that bin is the column's mode.  The synthetic shapes of bench.py have no such column.
    python tools/query_level_features.py [n_docs] [n_query_level_columns ...]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
from ranklib_amd import _native as N  # noqa: E402
from ranklib_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
ks = [int(a) for a in sys.argv[2:]] or [0, 8, 32]
X0, lab, qoff = synth.make_dataset(n, 136, "mslr", seed_offset=0)
rng = np.random.default_rng(1)
for k in ks:
    X = X0.copy()
    for f in range(k):
        vals = rng.random(len(qoff) - 1).astype(np.float32)
        X[:, 4 * f + 1] = np.repeat(vals, np.diff(qoff))
    g = N.Trainer(n_trees=40, n_leaves=31, flags=N.RL_FLAG_TIMING | N.RL_FLAG_TIMING_NODES)
    g.set_train(X, lab, qoff)
    g.init()
    g.boost_rounds_async(5); g.sync(); g.reset_timing()
    t0 = time.perf_counter()
    g.boost_rounds_async(30); g.sync()
    dt = time.perf_counter() - t0
    ms_root, n_root, _ = g.timing("HIST_ROOT")
    ms_node, n_node, _ = g.timing("HIST_NODE")
    print("%3d query-level columns of 136: %6.1f rounds/s, root histogram %.3f ms, child histograms %.3f ms per round"
          % (k, 30 / dt, ms_root / max(n_root, 1), ms_node / 30), flush=True)
    g.close()
