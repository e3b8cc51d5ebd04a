"""One MI355X, a training set ten times MSLR-WEB30K: N documents x 136 features held in HBM (rows, bins and both histogram layouts),
delivered through rl_set_rows in blocks (what the JNI shim does), a few boosting rounds timed, and -- with --oracle-rounds -- thresholds,
root counts, lambdas, weights, trees (tree_equiv), scores and the per-round metric compared with the CPU oracle on every host core.

    python tools/big_train.py --docs 40000000 --rounds 20 --oracle-rounds 2          (on the GPU box, via gpurun)

Every index in the library that is a product of documents and features must be 64-bit for this to pass (N x F = 5.4e9 > 2^32)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ranklib_amd import _native as N  # noqa: E402
from ranklib_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=40_000_000)
    ap.add_argument("--features", type=int, default=136)
    ap.add_argument("--kind", default="mslr")
    ap.add_argument("--leaves", type=int, default=31)
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--oracle-rounds", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=2_000_000)
    ap.add_argument("--java-order", action="store_true")
    a = ap.parse_args()
    n, F = a.docs, a.features
    sp = a.kind == "yahoo"
    t0 = time.time()
    qoff = synth.query_sizes(n, a.kind, synth.SEED_QSIZE)
    ns = min(262144, n)
    _, cuts = synth.labels_from(synth.features(ns, F, 0, synth.SEED_DATA, sparse=sp), 0, synth.SEED_LABEL)
    keep = a.oracle_rounds > 0
    X = np.empty((n, F), np.float32) if keep else None
    lab = np.empty(n, np.float32)
    for d0 in range(0, n, a.chunk):           # pass 1: labels (and the host copy the oracle needs)
        d1 = min(n, d0 + a.chunk)
        blk = X[d0:d1] if keep else np.empty((d1 - d0, F), np.float32)
        synth.features(d1 - d0, F, d0, synth.SEED_DATA, out=blk, sparse=sp)
        lab[d0:d1], _ = synth.labels_from(blk, d0, synth.SEED_LABEL, cuts=cuts)
    print("[big] %d documents x %d features, %d queries generated in %.0f s (%.1f GB of rows)" % (n, F, len(qoff) - 1, time.time() - t0, n * F * 4 / 1e9), flush=True)

    g = N.Trainer(n_trees=max(a.rounds, a.oracle_rounds, 1), n_leaves=a.leaves, flags=N.RL_FLAG_JAVA_ORDER if a.java_order else 0)
    t0 = time.time()
    g.N, g.F, g.Q = n, F, len(qoff) - 1
    qo = np.ascontiguousarray(qoff, np.int32)
    N.check(N.lib().rl_set_train(g.h, None, n, F, lab.ctypes.data, qo.ctypes.data, g.Q, None, None))
    for d0 in range(0, n, a.chunk):           # pass 2: rows, block by block
        d1 = min(n, d0 + a.chunk)
        blk = X[d0:d1] if keep else synth.features(d1 - d0, F, d0, synth.SEED_DATA, sparse=sp)
        N.check(N.lib().rl_set_rows(g.h, 0, d0, d1 - d0, blk.ctypes.data))
    t_up = time.time() - t0
    t0 = time.time()
    g.init()
    t_init = time.time() - t0
    used = total = 0.0
    try:
        import re
        import subprocess
        txt = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True, timeout=60).stdout
        total = float(re.search(r"VRAM Total Memory \(B\): (\d+)", txt).group(1))
        used = float(re.search(r"VRAM Total Used Memory \(B\): (\d+)", txt).group(1))
    except Exception:
        pass
    print("[big] upload %.0f s, rl_init %.1f s, HBM in use %.1f of %.0f GB" % (t_up, t_init, used / 1e9, total / 1e9), flush=True)

    if a.oracle_rounds > 0:
        import oracle_ffi as O
        from tree_equiv import assert_equivalent
        t0 = time.time()
        o = O.Oracle(X, lab, qo, n_trees=a.oracle_rounds, n_leaves=a.leaves, n_threads=os.cpu_count() or 8)
        o.init()
        t_oinit = time.time() - t0
        nb, thr, cnt = g.array("NBINS"), g.array("THRESHOLDS"), g.array("ROOT_COUNT")
        for f in range(F):
            T = o.n_bins(f)
            assert nb[f] == T, f
            assert np.array_equal(thr[f, :T].view(np.uint32), o.thresholds(f).view(np.uint32)), f
            assert np.array_equal(cnt[f, :T], o.root_count(f)), f
        stats, t_or = {}, 0.0
        for r in range(a.oracle_rounds):
            t0 = time.time()
            to, tmo, _, _ = o.round()
            t_or += time.time() - t0
            tg, tmg, _, _ = g.boost_round()
            assert np.array_equal(g.array("LAMBDA").view(np.int64), o.lambdas().view(np.int64)), r
            assert np.array_equal(g.array("WEIGHT").view(np.int64), o.weights().view(np.int64)), r
            assert_equivalent(to, tg, X, "round %d" % r, stats)
            assert np.float32(tmo).view(np.uint32) == np.float32(tmg).view(np.uint32), r
            assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r
        print("[big] oracle parity over %d rounds: thresholds, root counts, lambda, weight, scores, metric bit for bit; splits compared %d, "
              "tie-resolved differently %d; oracle init %.0f s, %.1f s/round on %d threads"
              % (a.oracle_rounds, stats.get("splits", 0), stats.get("plateau", 0), t_oinit, t_or / a.oracle_rounds, os.cpu_count() or 8), flush=True)
        if a.java_order:
            assert stats.get("plateau", 0) == 0
    done = g.round if hasattr(g, "round") else a.oracle_rounds
    left = a.rounds - a.oracle_rounds
    if left > 0:
        g.boost_rounds_async(min(3, left)); g.sync()
        left -= min(3, left)
    if left > 0:
        t0 = time.perf_counter()
        g.boost_rounds_async(left); g.sync()
        dt = time.perf_counter() - t0
        m = g.round_metrics(a.rounds - 1)
        cs, cm = g.array("CHAIN_STATS"), g.array("CHAIN_MISS")
        print("[big] float chains of the last round: segments %d, window misses repaired %d, finished serially %d; misses per leaf chain (lambda | weight): %s | %s"
              % (cs[0], cs[1], cs[2], cm[0][:a.leaves].tolist(), cm[1][:a.leaves].tolist()), flush=True)
        print("[big] %d rounds in %.2f s: %.2f boosting rounds/s (%.1f ms per round), train NDCG@10 %.4f" % (left, dt, left / dt, 1000 * dt / left, float(m[0])), flush=True)


if __name__ == "__main__":
    main()
