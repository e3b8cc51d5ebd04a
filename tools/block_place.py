#!/usr/bin/env python
"""Raw per-block stamps of ONE tree's growth kernels, with the hardware place of every block (stamp 6: XCD << 32 | HW_ID; -DRL_PHASE_CLOCKS build
selected with RLHIP_LIB): which CU ran which block when -- the load balance of the child-histogram passes (HISTORY.md 10.5).
usage (GPU box): RLHIP_LIB=... python tools/block_place.py out.npy [shape] [tree]   -> int64 [steps<=16][3 kernels][2048 blocks][8 stamps] (10 ns ticks)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = sys.argv[1]
shape = sys.argv[2] if len(sys.argv) > 2 else "c2"
tree = int(sys.argv[3]) if len(sys.argv) > 3 else 20
os.environ["RLHIP_TRACE_TREE"] = str(tree + 1)
from ranklib_amd import _native as N, synth  # noqa: E402

n_docs, n_feat, kind, _, leaves = synth.SHAPES[shape]
X, lab, qoff, _ = synth.make_shard(n_docs, n_feat, kind, 0, 1)
g = N.Trainer(n_trees=tree + 1, n_leaves=leaves)
g.set_train(X, lab, qoff)
g.init()
g.boost_rounds_async(tree + 1)
g.sync()
tr = g.array("BLOCK_TRACE")
np.save(out, np.ascontiguousarray(tr.reshape(64, 3, -1, 8)[:16]))
print("saved", out, tr.dtype, tr.shape)
