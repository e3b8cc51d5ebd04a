#!/usr/bin/env python
"""Does the SHARDED code path (one-rank communicator) grow the plain path's trees at full size?  (round 6: the 9 000-document tests said yes,
the c2 bench said no.)  usage (GPU box): python tools/dist_scale_check.py [mode=rccl1|cb1] [rounds] [sizes ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: F401  (torch's HIP runtime first, tests/conftest.py)
if torch.cuda.is_available():
    torch.cuda.init()
from test_gpu_dist import single  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "rccl1"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sizes = [int(v) for v in sys.argv[3:]] or [9000, 60000, 300000, 1200000, 3770000]
for n in sizes:
    cfg = (n, 136, "mslr", 3, 31, rounds)
    a = single(*cfg)
    b = single(*cfg, dist_mode=mode)
    msg = "identical"
    for i, (x, y) in enumerate(zip(a[0], b[0])):
        bad = [k for k in ("feature", "left", "right", "count") if not np.array_equal(x[k], y[k])]
        bad += [k for k in ("threshold", "output") if not np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32))]
        if not np.array_equal(x["deviance"].view(np.int64), y["deviance"].view(np.int64)):
            bad.append("deviance")
        if bad:
            nd = [j for j in range(len(x["feature"])) if j >= len(y["feature"]) or x["feature"][j] != y["feature"][j] or x["count"][j] != y["count"][j]]
            msg = "tree %d differs in %s; first node %s: plain (f %s, count %s) sharded (f %s, count %s)" % (
                i, bad, nd[:1], x["feature"][nd[0]] if nd else "-", x["count"][nd[0]] if nd else "-",
                y["feature"][nd[0]] if nd and nd[0] < len(y["feature"]) else "-", y["count"][nd[0]] if nd and nd[0] < len(y["count"]) else "-")
            break
    print("%8d documents x 136, 31 leaves, %d rounds, %s: %s; metrics %s / %s" % (n, rounds, mode, msg, a[1][-1], b[1][-1]), flush=True)
