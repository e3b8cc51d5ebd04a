#!/usr/bin/env python
"""Where do the bytes of the child-histogram passes come from?    Runs `bench.py --plain` a few rounds under rocprofv3, ONE
counter set per pass (kernel trace only, as MI355X_MICROARCH.md prescribes), joins the passes launch by launch (the run is deterministic: launch i of
k_hist<false> is the same growth step in every pass) and prints, for the launches that fill the chip and for the small ones apart:
  FETCH_SIZE (bytes the L2 fetched from memory), WRITE_SIZE, TCP_TCC_READ_REQ (requests L1 -> L2), TCC_REQ / TCC_HIT / TCC_MISS (L2 look-ups),
  TCC_EA_RDREQ / TCC_EA_RDREQ_32B (requests L2 -> memory, how many of them 32 bytes),
beside the pass's algorithmic bytes (documents accumulated x (2 F + 12)) from the growth-step log of the same schedule.
usage (GPU box): python tools/pmc_child_hist_cache.py [shape] [rounds]  > gpurun_out/r06_pmc_child_hist_cache.txt"""
import glob
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
SETS = ["FETCH_SIZE", "WRITE_SIZE", "TCP_TCC_READ_REQ_sum", "TCC_REQ_sum", "TCC_READ_sum", "TCC_READ_SECTORS_sum", "TCC_HIT_sum", "TCC_MISS_sum",
        "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]       # (gfx950 names: rocprofv3 --list-avail)


def one_pass(ctr):
    """per tree (the launches between two root passes, in dispatch order): the counter of every k_hist<false> launch"""
    d = tempfile.mkdtemp(prefix="rlhip_pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--shape", shape, "--steps", str(rounds - 1), "--warmup", "1", "--plain"]
    pr = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    if pr.returncode != 0:
        return None
    con = sqlite3.connect(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0])
    rows = con.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name=? group by kernel_name, dispatch_id order by dispatch_id", (ctr,)).fetchall()
    con.close()
    shutil.rmtree(d, ignore_errors=True)
    trees = []
    for name, _, v in rows:
        if "k_hist<true" in name:
            trees.append([])
        elif "k_hist<false" in name and trees:
            trees[-1].append(float(v))
    return trees


def step_log():
    """documents accumulated by every growth step of the same schedule (RLHIP_STEPLOG: one entry per slot)"""
    os.environ["RLHIP_STEPLOG"] = "1"
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from ranklib_amd import _native as N, synth
    n_docs, n_feat, kind, _, leaves = synth.SHAPES[shape]
    X, lab, qoff, _ = synth.make_shard(n_docs, n_feat, kind, 0, 1)
    g = N.Trainer(n_trees=rounds, n_leaves=leaves)
    g.set_train(X, lab, qoff)
    g.init()
    g.boost_rounds_async(rounds); g.sync()
    log = g.array("STEP_LOG")
    n = int(log[0])
    e = log[8:8 + 8 * n].reshape(n, 8)
    e = e[e[:, 1] == 0]                       # growth-step slots: tree, 0, step, slot, parent docs, built docs, tie, slots
    steps = {}
    for tree, _, step, _, pdocs, bdocs, _, _ in e:
        key = (int(tree), int(step))
        a = steps.setdefault(key, [0, 0])
        a[0] += int(bdocs); a[1] += int(pdocs)
    return n_docs, n_feat, steps


vals = {}
for c in SETS:
    v = one_pass(c)
    if v is None or len(v) == 0:
        print("# pass %s gave nothing (counter not available on this box?)" % c)
        continue
    vals[c] = v
n_docs, F, steps = step_log()
# launch i of a tree is its growth step i; the launches behind a tree's last logged step are the steps the host had enqueued beyond its end (no work)
ntree = min(len(v) for v in vals.values())
per_tree = {}
for (tree, step), a in steps.items():
    per_tree.setdefault(tree, []).append((step, a))
tree_ids = sorted(per_tree)
print("# %s, %d rounds: %d trees in every pass, %d in the step log; child-pass launches per tree %s, logged steps per tree %s"
      % (shape, rounds, ntree, len(tree_ids), [len(t) for t in next(iter(vals.values()))][:ntree], [len(per_tree[t]) for t in tree_ids]))
built = []
sel_idx = []          # (tree, launch) of every logged step
for ti, t in enumerate(tree_ids[:ntree]):
    for li, (step, a) in enumerate(sorted(per_tree[t])):
        if all(li < len(v[ti]) for v in vals.values()):
            built.append(a[0]); sel_idx.append((ti, li))
built = np.array(built, dtype=np.float64)
alg = built * (2.0 * F + 12.0)
m = len(built)
nl = m
work = np.arange(m)
vals = {c: np.array([v[ti][li] for ti, li in sel_idx], dtype=np.float64) for c, v in vals.items()}
big = built >= 400000          # ~ the steps whose chunks are balanced over two blocks per CU
for label, sel in (("steps that fill the chip (>= 400 k documents accumulated)", big), ("small steps", ~big), ("all steps", np.ones(m, bool))):
    k = int(sel.sum())
    if k == 0:
        continue
    print("\n%s: %d launches, %.0f k documents a launch, algorithmic bytes %.2f MB a launch" % (label, k, built[sel].mean() / 1e3, alg[sel].mean() / 1e6))
    for c, v in vals.items():
        x = v[:nl][work][sel]
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            print("  %-22s %10.2f MB a launch   = %.2f x algorithmic" % (c, x.mean() * 1024 / 1e6, x.sum() * 1024 / alg[sel].sum()))
        elif c == "TCP_TCC_READ_REQ_sum":
            print("  %-22s %10.0f k requests   (x 64 B = %.2f MB = %.2f x algorithmic: what the CUs ask the L2 for)" % (c, x.mean() / 1e3, x.mean() * 64 / 1e6, x.sum() * 64 / alg[sel].sum()))
        elif c == "TCC_READ_SECTORS_sum":
            print("  %-22s %10.0f k sectors    (x 32 B = %.2f MB = %.2f x algorithmic: what the L2 delivers)" % (c, x.mean() / 1e3, x.mean() * 32 / 1e6, x.sum() * 32 / alg[sel].sum()))
        elif c.startswith("TCC_EA0_RDREQ"):
            print("  %-22s %10.0f k requests" % (c, x.mean() / 1e3))
        else:
            print("  %-22s %10.0f k" % (c, x.mean() / 1e3))
    if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
        h, mi = vals["TCC_HIT_sum"][:nl][work][sel].sum(), vals["TCC_MISS_sum"][:nl][work][sel].sum()
        print("  L2 hit rate %.3f" % (h / max(h + mi, 1.0)))
    if all(("TCC_EA0_RDREQ%s_sum" % q) in vals for q in ("", "_32B", "_64B", "_128B")):
        g = lambda q: vals["TCC_EA0_RDREQ%s_sum" % q][:nl][work][sel].sum()
        a, b32, b64, b128 = g(""), g("_32B"), g("_64B"), g("_128B")
        byts = b32 * 32 + b64 * 64 + b128 * 128
        print("  memory reads by size: %.0f %% 32 B, %.0f %% 64 B, %.0f %% 128 B of %.0f k requests a launch = %.2f MB = %.2f x algorithmic"
              % (100.0 * b32 / max(a, 1.0), 100.0 * b64 / max(a, 1.0), 100.0 * b128 / max(a, 1.0), a / k / 1e3, byts / k / 1e6, byts / alg[sel].sum()))
