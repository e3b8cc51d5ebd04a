#!/usr/bin/env python
"""bench.py -- boosting rounds/sec of the MI355X LambdaMART path (BASELINE.json metric).

A "step" is one boosting round (one iteration of learning/tree/LambdaMART.java:180-251: lambdas,
root histogram, a 31-leaf tree grown best-first, leaf outputs, score update, train NDCG@10) on synthetic
MSLR-shaped data that is already resident in HBM when the timed region starts (init() -- binning,
H2D -- is reported separately, never inside `value`).

Workload: the shape BASELINE.json's metric is quoted on, MSLR-WEB30K-shape (configs[2]: 3.77 M documents x 136
features, ~31.5 k queries, 31 leaves); it fits one GPU, so N = 1 runs the whole set and N > 1 shards the SAME
set's queries contiguously over the ranks ("scaling": "strong"; one exact histogram all-reduce per split over
RCCL).  `--shape c1` runs configs[1] (MSLR-WEB10K-shape, 1.2 M documents).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...    (queries sharded by rank)

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the root histogram, rl::k_hist<true>):
algorithmic bytes per launch / HIP-event time of that launch measured live on the library's own stream.
`cpu_baseline` is the CPU oracle (java-exact restatement with RankLib's thread split) timed on this box's
host cores on the same data for a bounded number of rounds.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured float4 copy


def pmc_traffic(shape, world):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC pass (profiles/r01_pmc_root_hist.json:
    2 x FETCH_SIZE (gfx950 half-count correction) + WRITE_SIZE); null when no pass exists for this workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_root_hist.json")) as f:
            d = json.load(f)
        return float(d[shape]["traffic_bytes"]) if world == 1 and shape in d else None
    except Exception:
        return None


def bench_infer(args):
    """--workload infer (BASELINE.json configs[4] shape): Ensemble.eval of a 10 000-tree model on rows resident
    in HBM.  Trees: `--infer-train-rounds` real boosting rounds on MSLR-shaped data, tiled to `--trees`; rows: generated on
    the device with the same column kinds as ranklib_amd.synth.  A step scores the whole batch once."""
    import numpy as np
    import torch
    from ranklib_amd import _native as N
    from ranklib_amd import synth
    torch.cuda.set_device(0)
    F, L = 136, 31
    X, lab, qoff = synth.make_dataset(200000, F, "mslr")
    g = N.Trainer(n_trees=args.infer_train_rounds, n_leaves=L)
    g.set_train(X, lab, qoff)
    g.init()
    g.boost_rounds_async(args.infer_train_rounds)
    g.sync()
    g.finish()
    trees = [g.get_tree(i).trimmed() for i in range(args.infer_train_rounds)]
    text = g.model_text()
    head, body = text.split("<ensemble>\n", 1)
    blocks = body.rsplit("</ensemble>", 1)[0].split("\t</tree>\n")[:-1]
    reps = (args.trees + len(blocks) - 1) // len(blocks)
    tiled = []
    for r in range(reps):
        for b in blocks:
            if len(tiled) < args.trees:
                tiled.append(b.split(">", 1)[1])        # drop the <tree id=.. weight=..> opening, re-numbered below
    text = head + "<ensemble>\n" + "".join("\t<tree id=\"%d\" weight=\"0.1\">%s\t</tree>\n" % (i + 1, t) for i, t in enumerate(tiled)) + "</ensemble>\n"
    t0 = time.time()
    m = N.Model(text)
    t_load = time.time() - t0
    nt = m.num_trees()
    # mean path length on the training distribution from the per-node training counts
    visits = float(np.mean([t["count"][t["feature"] != -1].sum() / t["count"][0] for t in trees]))
    n = args.docs
    stride = F + 1
    gen = torch.Generator(device="cuda").manual_seed(20240601)
    dX = torch.empty((n, stride), dtype=torch.float32, device="cuda")
    chunk = 1 << 20
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        u = torch.rand((b - a, stride), generator=gen, device="cuda")
        k = torch.arange(stride, device="cuda") % 4                   # column kinds of synth.make_dataset (feature id f = column f)
        cnt = torch.floor(u * 21.0)
        heavy = torch.exp(4.0 * u)
        sparse = torch.where(torch.rand((b - a, stride), generator=gen, device="cuda") < 0.7, torch.zeros_like(u), u)
        dX[a:b] = torch.where(k == 1, cnt, torch.where(k == 2, u, torch.where(k == 3, heavy, sparse)))
        dX[a:b, 0] = 0
    dO = torch.empty(n, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        m.predict_device(dX.data_ptr(), n, stride, dO.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        m.predict_device(dX.data_ptr(), n, stride, dO.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / args.steps
    docs_per_s = n * args.steps / elapsed
    out = {
        "metric": "documents scored/sec (Ensemble.eval, %d trees x %d leaves)" % (nt, L), "value": docs_per_s, "unit": "docs/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 compares, f32 score chain (as Ensemble.eval)",
        "data": "synthetic",
        "config": {"workload": "c4-shape inference: %d documents x %d features resident in HBM, %d trees (%d trained rounds tiled), "
                               "31 leaves" % (n, F, nt, args.infer_train_rounds),
                   "mean_node_visits_per_tree": visits, "node_visits_per_s": docs_per_s * nt * visits,
                   "model_parse_seconds": round(t_load, 2)},
        "roofline": {"kernel": "rl::k_model_eval_tiled", "bound": "hbm", "achieved": n * stride * 4.0 / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": n * stride * 4.0 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "note": "HBM is not the limiter here: every row is read once and visited %d x %.1f times in LDS; the kernel is bound by "
                             "LDS / VALU issue (DESIGN.md 4.8), node_visits_per_s is the figure of merit" % (nt, visits)},
    }
    if args.cpu_rounds > 0:
        import oracle_ffi as O
        threads = args.cpu_threads or (os.cpu_count() or 1)
        all_trees = [trees[i % len(trees)] for i in range(nt)]
        # a short probe sizes the timed sample to ~3 s of wall time on all host threads (thread start-up dominates shorter runs)
        probe = min(n, threads * 64)
        tc = time.perf_counter()
        O.eval_flat_model(all_trees, dX[:probe].cpu().numpy(), n_threads=threads)
        t_probe = max(time.perf_counter() - tc, 1e-3)
        ns = int(min(n, max(probe, probe * 3.0 / t_probe)))
        rows = dX[:ns].cpu().numpy()
        tc = time.perf_counter()
        ref = O.eval_flat_model(all_trees, rows, n_threads=threads)
        t_cpu = time.perf_counter() - tc
        same = bool(np.array_equal(ref.view(np.uint32), dO[:ns].cpu().numpy().view(np.uint32)))
        out["cpu_baseline"] = {"value": ns / t_cpu, "unit": "docs/s", "cores": threads, "kind": "port",
                               "sample": "the first %d rows of the same batch, same %d trees, C restatement of Ensemble.eval with the rows split "
                                         "over %d threads; scores identical to the GPU's: %s" % (ns, nt, threads, same)}
        out["speedup_vs_cpu_baseline"] = docs_per_s / (ns / t_cpu)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", default="c2", help="c0 | c1 | c1ns | c2 (ranklib_amd.synth.SHAPES)")
    ap.add_argument("--cpu-rounds", type=int, default=8, help="rounds timed for the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores (RankLib's default -thread)")
    ap.add_argument("--no-timing", action="store_true", help="do not record HIP events around the dominant kernel")
    ap.add_argument("--workload", default="train", help="train (default, the BASELINE.json metric) | infer (configs[4]: Ensemble.eval)")
    ap.add_argument("--trees", type=int, default=10000, help="infer: trees in the scored model")
    ap.add_argument("--docs", type=int, default=10000000, help="infer: rows per step (54.8 GB at the 100 M of configs[4])")
    ap.add_argument("--infer-train-rounds", type=int, default=100)
    args = ap.parse_args()
    if args.workload == "infer":
        if not any(a.startswith("--steps") for a in sys.argv):
            args.steps, args.warmup = 3, 1
        return bench_infer(args)

    import numpy as np
    import torch
    from ranklib_amd import _native as N
    from ranklib_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RLHIP_BENCH_SAME_GPU"):        # debugging aid: several ranks on one device (RCCL may refuse this)
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        # control plane only (rendezvous, unique-id broadcast, timing max): gloo.  The data path (histogram
        # all-reduce, gathers) is the library's own RCCL communicator over xGMI.
        import torch.distributed as dist
        dist.init_process_group("gloo")

    n_docs, n_feat, kind, n_trees, n_leaves = synth.SHAPES[args.shape]
    t0 = time.time()
    X, lab, qoff, q_total = synth.make_shard(n_docs, n_feat, kind, rank, world)
    t_gen = time.time() - t0

    flags = 0 if args.no_timing else N.RL_FLAG_TIMING
    total_rounds = args.warmup + args.steps
    g = N.Trainer(n_trees=max(total_rounds, 1), n_leaves=n_leaves, device=local_rank, flags=flags)
    t0 = time.time()
    g.set_train(X, lab, qoff)
    if world > 1:
        box = [g.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        g.dist_init(box[0], rank, world)
    g.init()
    t_init = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        g.boost_rounds_async(args.warmup)
        g.sync()
    g.reset_timing()
    barrier()
    t0 = time.perf_counter()
    g.boost_rounds_async(args.steps)
    g.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ndcg_t = float(g.round_metrics(total_rounds - 1)[0])
    ms_root, n_root, bytes_root = g.timing("HIST_ROOT")
    ms_node, n_node, _ = g.timing("HIST_NODE")
    ms_lam, n_lam, _ = g.timing("LAMBDA")
    gs = g.array("GROW_STATS")

    if rank != 0:
        return
    rounds_per_s = args.steps / elapsed
    out = {
        "metric": "boosting rounds/sec (LambdaMART -ranker 6, NDCG@10)",
        "value": rounds_per_s,
        "unit": "rounds/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "int64 fixed-point histograms + f64 lambdas/scores (f32 leaf chains as in the Java)",
        "data": "synthetic",
        "config": {
            "workload": "%s: synthetic %s, %d docs x %d features, %d queries (%s docs/query) in total, queries sharded "
                        "contiguously over %d GPU(s); LambdaMART -ranker 6, %d leaves, lr 0.1, -tc 256, -mls 1, NDCG@10" %
                        (args.shape, {"c2": "MSLR-WEB30K-shape (BASELINE.json configs[2], the shape the metric is quoted on)",
                                      "c1": "MSLR-WEB10K-shape (BASELINE.json configs[1])"}.get(args.shape, args.shape),
                         n_docs, n_feat, q_total, "~120 log-normal" if kind == "mslr" else "5..15", world, n_leaves),
            "docs_total": n_docs, "docs_rank0": int(X.shape[0]), "features": n_feat, "queries_total": q_total, "leaves": n_leaves,
            "ndcg10_train_after_%d_rounds" % total_rounds: ndcg_t,
            "init_seconds": round(t_init, 3), "datagen_seconds": round(t_gen, 3),
            "growth_steps_per_tree": round(float(gs[0]) / max(int(gs[3]), 1), 2),
            "nodes_prepared_per_tree": round(float(gs[1]) / max(int(gs[3]), 1), 2),
            "splits_per_tree": round(float(gs[2]) / max(int(gs[3]), 1), 2),
        },
    }
    if n_root > 0:
        per_launch_ms = ms_root / n_root
        alg_bytes = bytes_root / n_root
        achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9
        out["roofline"] = {
            "kernel": "rl::k_hist<true,16> (root histogram, FeatureHistogram.update)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(args.shape, world),
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": per_launch_ms, "launches": int(n_root),
            "note": "algorithmic bytes = N_local*(F*2 B bin ids + 8 B fixed-point lambda); HIP events on the library stream",
        }
        out["kernel_ms_per_round"] = {"hist_root": ms_root / args.steps, "lambda": ms_lam / args.steps}

    if args.cpu_rounds > 0 and world == 1:
        import oracle_ffi as O
        threads = args.cpu_threads or (os.cpu_count() or 1)
        o = O.Oracle(X, lab, qoff, n_trees=args.cpu_rounds, n_leaves=n_leaves, n_threads=threads)
        tc = time.perf_counter()
        o.init()
        t_cpu_init = time.perf_counter() - tc
        tc = time.perf_counter()
        for _ in range(args.cpu_rounds):
            o.round()
        t_cpu = time.perf_counter() - tc
        out["cpu_baseline"] = {
            "value": args.cpu_rounds / t_cpu, "unit": "rounds/s", "cores": threads, "kind": "port",
            "sample": "same data set, %d boosting rounds of the java-exact C oracle with RankLib's MyThreadPool work split "
                      "(init %.1f s not counted)" % (args.cpu_rounds, t_cpu_init),
        }
        out["speedup_vs_cpu_baseline"] = rounds_per_s / (args.cpu_rounds / t_cpu)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
