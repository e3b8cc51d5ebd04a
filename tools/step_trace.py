#!/usr/bin/env python
"""Where a growth step's time goes, kernel by kernel: entry / exit wall-clock stamps of every working block of the partition / child-histogram /
finish kernels of ONE tree.  Needs a library built with -DRL_PHASE_CLOCKS (see tools/phase_clocks.py), selected with RLHIP_LIB.
usage (GPU box): RLHIP_LIB=... python tools/step_trace.py [shape] [tree]      (tree = 0-based boosting round to trace)
Per step and kernel: gap = the previous kernel's last working block leaving .. this kernel's first working block entering (launch boundary +
dispatch), span = first entry .. last exit, med / max = median / longest single block, n = working blocks (at most 2048 are recorded)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
tree = int(sys.argv[2]) if len(sys.argv) > 2 else 20
os.environ["RLHIP_TRACE_TREE"] = str(tree + 1)          # TreeState::tree_seq while tree `tree` grows
from ranklib_amd import _native as N, synth  # noqa: E402

n_docs, n_feat, kind, _, leaves = synth.SHAPES[shape]
X, lab, qoff, _ = synth.make_shard(n_docs, n_feat, kind, 0, 1)
g = N.Trainer(n_trees=tree + 1, n_leaves=leaves)
g.set_train(X, lab, qoff)
g.init()
g.boost_rounds_async(tree + 1)
g.sync()
clk = g.array("PHASE_CLOCKS").astype(np.float64) * 0.01          # us
tr = g.array("BLOCK_TRACE").astype(np.float64) * 0.01
if os.environ.get("RLHIP_TRACE_DUMP"):                   # raw stamps [step][kernel][block][stamp] for offline analysis
    np.save(os.environ["RLHIP_TRACE_DUMP"], tr.astype(np.float32) - np.float32(0) if False else (tr - tr[tr > 0].min()).astype(np.float32))
print("%s, tree %d: per growth step, microseconds" % (shape, tree))
print("step | part: gap  span   med   max    n | hist: gap  span   med   max    n | finish: gap  span   med   max    n | select tail | step total")
prev_end = None
rows = []
for s in range(64):
    end = clk[s][28]
    ks = []
    for k in range(3):
        b = tr[s, k]
        m = b[:, 7] > 0
        if not m.any():
            ks = None
            break
        t0, t1 = b[m, 0], b[m, 7]
        ks.append((t0.min(), t1.max(), float(np.median(t1 - t0)), float((t1 - t0).max()), int(m.sum())))
    if ks is None or end == 0:
        continue
    (p0, p1, pmed, pmax, pn), (h0, h1, hmed, hmax, hn), (f0, f1, fmed, fmax, fn) = ks
    gap_p = (p0 - prev_end) if prev_end is not None else float("nan")
    total = (end - prev_end) if prev_end is not None else float("nan")
    print("%4d | %9.1f %5.1f %5.1f %5.1f %4d | %9.1f %5.1f %5.1f %5.1f %4d | %11.1f %5.1f %5.1f %5.1f %4d | %11.1f | %8.1f" %
          (s, gap_p, p1 - p0, pmed, pmax, pn, h0 - p1, h1 - h0, hmed, hmax, hn, f0 - h1, f1 - f0, fmed, fmax, fn, end - f1, total))
    if prev_end is not None:
        rows.append([gap_p, p1 - p0, h0 - p1, h1 - h0, f0 - h1, f1 - f0, end - f1, total])
    prev_end = end
sel2 = [(s_, clk[s_][20:26] - clk[s_][20]) for s_ in range(64) if clk[s_][25] > clk[s_][20] > 0]
if sel2:       # k_select2 (rl_step2.inc): stamps of its phases since the kernel's first instruction
    print("bookkeeping kernel (k_select2), microseconds since its entry: loads landed, A best feature, B children, C fit loop, D next slots")
    for s_, r in sel2:
        print("%4d " % s_ + " ".join("%7.2f" % v for v in r[1:]))
    print("mean " + " ".join("%7.2f" % v for v in np.mean(np.array([r for _, r in sel2]), axis=0)[1:]))
if rows:
    m = np.mean(np.array(rows), axis=0)
    print("mean over %d steps: part gap %.1f span %.1f | hist gap %.1f span %.1f | finish gap %.1f span %.1f | select tail %.1f | step %.1f" % ((len(rows),) + tuple(m)))

# phase stamps inside the blocks (median over the working blocks of the step, microseconds since the block's entry)
names = {0: ["slot table", "idx + bins loaded", "ballots + sync", "look-back done", "stores issued", "", "exit"],
         1: ["slot + modes", "LDS zeroed", "first ids", "loop done", "totals", "", "exit"],
         2: ["step tables", "chunk sums", "mode bin + prefix", "gain scan", "block best + ties", "", "published"]}
for k in (0, 1, 2):
    print("kernel %d (%s): median stamp since block entry, by step" % (k, ["partition", "child histogram", "finish"][k]))
    print("step " + " ".join("%-18s" % n for n in names[k] if n))
    for s_ in range(64):
        b = tr[s_, k]
        m = b[:, 7] > 0
        if not m.any():
            continue
        rel = b[m] - b[m, 0:1]
        med = np.median(rel, axis=0)
        print("%4d " % s_ + " ".join("%-18.2f" % med[i + 1] for i, n in enumerate(names[k]) if n))
