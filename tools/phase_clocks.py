#!/usr/bin/env python
"""Where the time of k_hist_finish / select_step goes: device wall-clock stamps of one tree's growth steps.
Needs a library built with -DRL_PHASE_CLOCKS (make -C ranklib_amd/csrc EXTRA=-DRL_PHASE_CLOCKS LIB=...), selected with RLHIP_LIB.
usage (GPU box): RLHIP_LIB=... python tools/phase_clocks.py [shape] [rounds]
build: cd ranklib_amd/csrc && hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DRL_PHASE_CLOCKS -shared -o ../lib/variants/clk.so -x hip rl_trainer.hip -x hip rl_model.cpp"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ranklib_amd import _native as N, synth  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n_docs, n_feat, kind, _, leaves = synth.SHAPES[shape]
X, lab, qoff, _ = synth.make_shard(n_docs, n_feat, kind, 0, 1)
g = N.Trainer(n_trees=rounds, n_leaves=leaves)
g.set_train(X, lab, qoff)
g.init()
g.boost_rounds_async(rounds)
g.sync()
clk = g.array("PHASE_CLOCKS").astype(np.float64) * 0.01          # us
names = ["block(0,0): entry", "chunk sums", "mode bin", "prefix", "gain scan", "block best", "publish + waitcnt", "arrive",
         "LAST block: select entry", "A best feature", "B children of the step", "C fit loop", "D next slots", "copy back"]
print("step  " + "  ".join("%-9s" % n[:9] for n in names[1:8]) + " | " + "  ".join("%-9s" % n[:9] for n in names[9:]) + " | blk0->select  total")
for step in range(64):
    r = clk[step]
    if r[0] == 0 or r[13] == 0 or r[13] < r[0]:
        continue
    a = [r[k] - r[k - 1] for k in range(1, 8)]
    b = [r[k] - r[k - 1] for k in range(9, 14)]
    print("%4d  " % step + "  ".join("%9.2f" % v for v in a) + " | " + "  ".join("%9.2f" % v for v in b) + " | %9.2f  %9.2f" % (r[8] - r[7], r[13] - r[0]) +
          " | select entry -> state copied %.2f, -> records reduced %.2f" % (r[8] - r[14], r[15] - r[8]) +
          " (prefetch issued %.2f, share reduced %.2f, half wave %.2f, tie pass %.2f, stored %.2f)" % (r[16] - r[8], r[17] - r[16], r[18] - r[17], r[19] - r[18], r[15] - r[19]))
