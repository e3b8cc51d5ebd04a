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

  python bench.py --gpus N --steps K --warmup W      N > 1: the script starts its own N ranks (torch.distributed.run, one per GPU)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...    (the same, launched from outside)
  --scaling weak: N x the shape's documents (the same documents per rank at every N) instead of the same set sharded

Prints ONE JSON line on rank 0.  `roofline` is for the time-dominant kernel (the child-node histogram passes, rl::k_hist<false>):
algorithmic bytes per launch / HIP-event time of that launch measured live on the library's own stream, with the counter traffic
(`traffic`, `frac_traffic`) beside SURVEY.md 8d's figure; `roofline.root_pass` is the root histogram (rl::k_hist<true>) likewise,
`roofline.lds_atomic_calibration` the measured LDS atomic rates both kernels' atomics are a fraction of.
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


# rocprofv3 FETCH_SIZE -> bytes on gfx950, calibrated on known byte counts with this library's own access patterns
# (tools/calib_fetch.sh, profiles/r02_fetch_calibration.txt): the counter tallies half of the bytes of 16-byte-per-lane streams AND of
# 32-byte row gathers; WRITE_SIZE is exact.  Both in KB.
FETCH_FACTOR_WIDE = 2.0
FETCH_FACTOR_GATHER32 = 2.0


def live_pmc(shape, rounds=4):
    """HBM traffic of the histogram kernels, MEASURED by this run: bench.py re-runs itself (a few rounds, --plain) under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes) and
    reads the per-dispatch counters.  Returns None when rocprofv3 is not usable (the field is then null, never a stale number)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    res = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="rlhip_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--shape", shape, "--steps", str(rounds - 1), "--warmup", "1", "--plain"]
            # its own process group: on a timeout the profiler AND the benchmark under it are killed (by exact pgid)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                if pr.wait(timeout=150) != 0:
                    raise RuntimeError("rocprofv3 exited with %d" % pr.returncode)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(pr.pid, signal.SIGKILL)
                pr.wait()
                raise
            con = sqlite3.connect(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0])
            for name, n, tot in con.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
                if "k_hist<true" in name:
                    res["root_" + ctr] = tot / n * 1024.0                  # per launch
                elif "k_hist<false" in name:
                    res["node_" + ctr] = tot / rounds * 1024.0            # per round (all growth steps)
            con.close()
            shutil.rmtree(d, ignore_errors=True)
        return res if "root_FETCH_SIZE" in res and "root_WRITE_SIZE" in res else None
    except Exception as ex:      # noqa: BLE001
        sys.stderr.write("live PMC pass failed: %r\n" % (ex,))
        return None


def bench_infer(args):
    """--workload infer (BASELINE.json configs[4] shape): Ensemble.eval of a 10 000-tree model on rows resident
    in HBM.  Trees: `--infer-train-rounds` real boosting rounds on MSLR-shaped data, tiled to `--trees`; rows: generated on
    the device with the same column kinds as ranklib_amd.synth.  A step scores the whole batch once."""
    import numpy as np
    import torch
    from ranklib_amd import _native as N
    from ranklib_amd import synth
    # N > 1 (torch.distributed.run): independent replicas, every rank scores its own `--docs` rows with the same model ("scaling": "weak";
    # there is nothing to exchange in Ensemble.eval).  gloo only carries the barrier and the max of the elapsed times.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return respawn(args.gpus)
    claim_stdout()
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        dist = init_control_plane()
    torch.cuda.set_device(local_rank)
    F, L = 136, 31
    X, lab, qoff = synth.make_dataset(200000, F, "mslr")
    g = N.Trainer(n_trees=args.infer_train_rounds, n_leaves=L, device=local_rank)
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
    m = N.Model(text, device=local_rank)
    t_load = time.time() - t0
    nt = m.num_trees()
    # mean path length on the training distribution from the per-node training counts
    visits = float(np.mean([t["count"][t["feature"] != -1].sum() / t["count"][0] for t in trees]))

    def depth_of(t):        # levels below the root of the deepest leaf
        d = np.zeros(len(t["feature"]), np.int32)
        for i in range(len(t["feature"])):      # (trimmed trees list a parent before its children)
            if t["feature"][i] != -1:
                d[int(t["left"][i])] = d[int(t["right"][i])] = d[i] + 1
        return int(d.max())
    # chain steps per tree as the kernel walks them: the trees of a 32-tree tile, sorted by depth, are dealt round its four walker wavefronts; a walker runs
    # all eight chains for as many steps as its 7th tree has levels, then six to its 5th tree's depth, four to its 3rd's, two to its deepest's (round 6;
    # until then: all eight to the deepest -- `lockstep_steps_per_tree`)
    dep = np.array([max(depth_of(trees[i % len(trees)]), 1) for i in range(args.trees)])
    steps_sum = lock_sum = 0
    for a in range(0, len(dep), 32):
        ds_ = np.sort(dep[a:a + 32])[::-1]
        # a full tile's sorted trees are dealt round the four walkers (walker p: ranks p, p + 4, ..), a partial last tile eight consecutive ranks each
        groups = [ds_[p::4] for p in range(4)] if len(ds_) == 32 else [ds_[q:q + 8] for q in range(0, len(ds_), 8)]
        for grp in groups:
            g = [int(v) for v in grp]
            lock_sum += g[0] * len(g)
            prev, done = 0, 0
            for n_ch, nxt in ((8, 6), (6, 4), (4, 2), (2, 0)):        # chains walking, index of the tree whose depth ends the phase
                end = g[nxt] if nxt < len(g) else 0
                if nxt == 0:
                    end = g[0]
                if end > prev:
                    done += min(n_ch, len(g)) * (end - prev)
                    prev = end
            steps_sum += done
    walk_steps = steps_sum / float(len(dep))
    lock_steps = lock_sum / float(len(dep))
    n = args.docs
    stride = F + 1
    gen = torch.Generator(device="cuda").manual_seed(20240601 + rank)
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
    if dist is not None:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        m.predict_device(dX.data_ptr(), n, stride, dO.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms = e0.elapsed_time(e1) / args.steps
    docs_per_s = n * world * args.steps / elapsed
    # --infer-ab VAR=v1,v2,..: the same rows with an environment knob of the kernel set to each value (one warm-up + `--steps` timed passes each), and
    # whether every value left the same score bits (RLHIP_EVAL_COMPACT=1,0: rank-coded against float cells)
    variants = {}
    if args.infer_ab:
        var, vals = args.infer_ab.split("=", 1)
        ref_bits = None
        for v in vals.split(","):
            os.environ[var] = v
            for kv in v.split("+")[1:]:          # "1+OTHER=3": further knobs for this value
                os.environ[kv.split(":")[0]] = kv.split(":")[1]
            m.predict_device(dX.data_ptr(), n, stride, dO.data_ptr())
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(args.steps):
                m.predict_device(dX.data_ptr(), n, stride, dO.data_ptr())
            torch.cuda.synchronize()
            tv = time.perf_counter() - tv
            bi = dO.view(torch.int32).to(torch.int64)
            bits = (int(bi.sum().item()), int((bi * (torch.arange(n, device="cuda") % 8191 + 1)).sum().item()))
            if ref_bits is None:
                ref_bits = bits
            variants["%s=%s" % (var, v)] = {"docs_per_s": n * args.steps / tv, "same_bits_as_first": bits == ref_bits}
        os.environ.pop(var, None)
    if rank != 0:
        return
    out = {
        "metric": "documents scored/sec (Ensemble.eval, %d trees x %d leaves)" % (nt, L), "value": docs_per_s, "unit": "docs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 compares, f32 score chain (as Ensemble.eval)",
        "data": "synthetic",
        "config": {"workload": "c4 (BASELINE.json configs[4]) inference: %d documents x %d features per GPU resident in HBM, %d GPU(s) as independent replicas, "
                               "%d trees (%d trained rounds tiled), 31 leaves" % (n, F, world, nt, args.infer_train_rounds),
                   "mean_node_visits_per_tree": visits, "node_visits_per_s": docs_per_s * nt * visits,
                   "model_parse_seconds": round(t_load, 2), "same_rows_ab": variants},
        "roofline": {
            # the walk is bound by instruction issue, not by bytes: a chain step is 5 vector-ALU + 2 LDS wave-instructions for 64 lanes (HISTORY.md 4.8).
            # peak = what the four SIMDs of a CU issue if they did nothing else: 4 SIMDs x 64 lanes / (5 VALU x 4 cycles) lane-steps per CU and clock.
            "kernel": "rl::k_model_eval_tiled", "bound": "valu_issue",
            "achieved": docs_per_s / world * nt * walk_steps / (256 * 2.4e9), "peak": 4 * 64 / (5 * 4.0), "unit": "lane-steps per CU and clock",
            "frac": docs_per_s / world * nt * walk_steps / (256 * 2.4e9) / (4 * 64 / (5 * 4.0)),
            "lds_pipe_peak": 64 / ((256 + 512) / 128.0),
            "walk_steps_per_tree": walk_steps, "lockstep_steps_per_tree": lock_steps, "mean_path_nodes_per_tree": visits, "lane_use": visits / walk_steps,
            "useful_node_visits_per_cu_clock": docs_per_s / world * nt * visits / (256 * 2.4e9),
            "hbm": {"achieved": n * stride * 4.0 / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": n * stride * 4.0 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "traffic": None,
            "note": "every row is read from HBM once and walked %d x %.1f steps in LDS: HBM is at a fraction of a percent.  A lane-step = one lane advancing one node "
                    "(a leaf repeats itself until its chain's phase ends: eight chains to the walker's 7th-deepest tree, six to the 5th, four to the 3rd, two to the deepest); lds_pipe_peak = 64 lanes / (768 B per wave-step / 128 B per clock); "
                    "counters: profiles/r04_infer_*" % (nt, walk_steps)},
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
    emit(json.dumps(out))


_REAL_STDOUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too -- gloo announces its connections, RCCL prints a version banner through the C
    library's buffered stdout when a communicator is created (it would be flushed BEHIND the JSON line at exit) -- so for the life of the process file
    descriptor 1 points at stderr and only emit() writes to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(line)
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, (line + "\n").encode())


def init_control_plane():
    """gloo process group for rendezvous, the unique-id broadcast and the timing max.  Gloo announces its connections on STDOUT ("[Gloo] Rank 0 is
    connected to ..."), which would break the one-JSON-line contract: file descriptor 1 points at stderr while the group is set up."""
    import torch.distributed as dist
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        dist.init_process_group("gloo")
        dist.barrier()                       # (the full mesh is connected -- and announced -- no later than the first collective)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    return dist


def respawn(n):
    """`python bench.py --gpus N` as typed: re-exec this script as N ranks (one per GPU) under torch.distributed.run on a free local port.
    Rank 0 prints the one JSON line; stdout / stderr / exit code are the launcher's."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def scaling_model(docs_per_rank, world, steps_tree=10.8, ar_bytes=None):
    """MODELLED (never fitted to a multi-GPU run) milliseconds per round of a sharded run, from the ONE-RANK run of the sharded code path at c2 of round 6
    (profiles/r06f_sharded_one_rank_kernel_stats.txt / _timeline.txt: everything of the N > 1 path but the wire, 385 rounds/s against 429 of the plain path):
    kernels whose work is per document scale with the shard; a growth step (single-pass partition from local counts, child histograms, limb reduce,
    all-reduce, k_fin2<DIST>, k_select2) is ~112 us at 3.77 M documents per rank, of which ~70 us are the latency floor of its five dependent launches
    whatever the shard size, and pays one all-reduce (ASSUMED 20 us + bytes at 40 GB/s effective ring bandwidth per rank over xGMI); the leaves' float
    chains run on every rank's own pieces (rl_dist.inc piece mode, profiles/r06i_*: 0.30 ms at 3.77 M documents per rank + ~60 us of small launches and the
    host's look at the walk's result + three small all-gathers, ASSUMED 15 us each)."""
    share = docs_per_rank / 3.77e6
    if ar_bytes is None:
        ar_bytes = 2.2 * 559e3          # ~2.2 slots of 559 KB per step on average at F = 136
    t_doc = (0.355 + 0.133 + 0.30 + 0.015) * share                 # lambdas, ranking + per-query metric, root pass + reduce + finish, score update
    t_step = steps_tree * (0.070 + 0.042 * share + (0.020 + ar_bytes / 40e9 * 1e3 if world > 1 else 0.0))
    # leaf sums (piece mode): gather in leaf order + the chain pipeline on the rank's own pieces scale with the shard; tables, totals and drifts are all-gathered
    t_leaf = 0.30 * share + 0.06 + (0.045 if world > 1 else 0.0)
    return {"modelled_ms_per_round": t_doc + t_step + t_leaf, "modelled_rounds_per_s": 1000.0 / (t_doc + t_step + t_leaf), "world": world,
            "docs_per_rank": docs_per_rank, "growth_steps_per_tree": steps_tree, "allreduce_bytes_per_call": ar_bytes,
            "note": "modelled = per-document kernels x shard share + growth steps x (70 us floor + 42 us x share + one all-reduce) + leaf sums on every rank's own pieces (0.30 ms x share + 0.1)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", default="c2", help="c0 | c1 | c1ns | c2 | c2ns | c3 (ranklib_amd.synth.SHAPES)")
    ap.add_argument("--cpu-rounds", type=int, default=8, help="rounds timed for the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores (RankLib's default -thread)")
    ap.add_argument("--no-timing", action="store_true", help="do not record HIP events around the dominant kernel")
    ap.add_argument("--sustain", type=int, default=300, help="extra rounds after the timed region for config.sustained_rounds_per_s (0 = skip)")
    ap.add_argument("--node-rounds", type=int, default=10, help="extra rounds with HIP events around every child-node histogram launch (0 = skip)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--sharded-one-rank", action="store_true", help="N = 1 only: run the sharded (multi-GPU) code path on a one-rank RCCL communicator")
    ap.add_argument("--plain", action="store_true", help="timed region only: no CPU baseline, sustained run, node timing, membench or PMC passes")
    ap.add_argument("--metric", default="NDCG", help="train metric: NDCG (the BASELINE.json metric) | DCG | ERR | MAP (RankLib's own default is ERR@10)")
    ap.add_argument("--first-tie", action="store_true", help="RL_FLAG_FIRST_TIE: exact ties keep the first candidate (no lazy Java-order tie-break)")
    ap.add_argument("--java-order", action="store_true", help="RL_FLAG_JAVA_ORDER: the strict mode (split gains from the Java's own f64 summation order)")
    ap.add_argument("--workload", default="train", help="train (default, the BASELINE.json metric) | infer (configs[4]: Ensemble.eval)")
    ap.add_argument("--scaling", default="strong", help="strong (default: the SAME data set sharded over --gpus ranks, what BASELINE.json configs[2] states) | "
                                                       "weak (--gpus x the shape's documents: the same documents per rank at every N)")
    ap.add_argument("--c1-trees", type=int, default=1000, help="after the headline run: BASELINE.json configs[1] as stated (c1 shape, this many trees) -> config.c1_full_run (0 = skip; N = 1 only)")
    ap.add_argument("--c2-trees", type=int, default=1000, help="after the headline run (c2, N = 1): the same shape as a whole run of this many trees -> config.c2_full_run (0 = skip)")
    ap.add_argument("--shard1-rounds", type=int, default=40, help="after the headline run (c2, N = 1): the sharded code path on a one-rank RCCL communicator beside the plain path -> config.sharded_path_one_rank (0 = skip)")
    ap.add_argument("--ns-rounds", type=int, default=20, help="after the headline run (c2, N = 1): the north-star list-length variant of the same shape, c2ns "
                    "(~10 docs/query, 377 k queries), timed over this many rounds -> config.c2ns (0 = skip)")
    ap.add_argument("--trees", type=int, default=10000, help="infer: trees in the scored model")
    ap.add_argument("--docs", type=int, default=100000000, help="infer: rows per GPU and step (configs[4]: 100 M = 54.8 GB of rows in HBM)")
    ap.add_argument("--infer-train-rounds", type=int, default=100)
    ap.add_argument("--infer-ab", default="", help="infer: time the same rows with an environment knob at several values, e.g. RLHIP_EVAL_COMPACT=1,0")
    args = ap.parse_args()
    if args.plain:
        args.cpu_rounds, args.sustain, args.node_rounds, args.no_pmc, args.no_timing, args.c1_trees, args.ns_rounds, args.c2_trees, args.shard1_rounds = 0, 0, 0, True, True, 0, 0, 0, 0
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
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return respawn(args.gpus)
    claim_stdout()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: run `python bench.py --gpus N` (it starts its own ranks) or launch N ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        # control plane only (rendezvous, unique-id broadcast, timing max): gloo.  The data path (histogram
        # all-reduce, gathers) is the library's own RCCL communicator over xGMI.
        dist = init_control_plane()

    n_docs, n_feat, kind, n_trees, n_leaves = synth.SHAPES[args.shape]
    weak = args.scaling == "weak"
    if weak:
        n_docs *= world         # the same documents per rank at every N
    t0 = time.time()
    X, lab, qoff, q_total = synth.make_shard(n_docs, n_feat, kind, rank, world)
    t_gen = time.time() - t0

    flags = 0 if args.no_timing else N.RL_FLAG_TIMING
    if args.first_tie:
        flags |= N.RL_FLAG_FIRST_TIE
    if args.java_order:
        flags |= N.RL_FLAG_JAVA_ORDER
        args.sustain = min(args.sustain, 20)
    total_rounds = args.warmup + args.steps
    g = N.Trainer(n_trees=max(total_rounds + args.sustain + args.node_rounds, 1), n_leaves=n_leaves, device=local_rank, flags=flags,
                  metric=args.metric.upper(), metric_k=0 if args.metric.upper() == "MAP" else 10)
    t0 = time.time()
    g.set_train(X, lab, qoff)
    if world > 1:
        if os.environ.get("RLHIP_BENCH_TRANSPORT") == "gloo":
            # test aid (with RLHIP_BENCH_SAME_GPU): the host-callback transport lets several ranks share the one GPU of a test box, where
            # RCCL refuses duplicate devices -- exercises this script's N > 1 path, not the interconnect
            from ranklib_amd import dist as D
            tr = D.TorchHostTransport()
            g.dist_init_callback(rank, world, tr.allreduce, tr.allgather, tr.alltoallv)
        else:
            box = [g.dist_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            g.dist_init(box[0], rank, world)
    elif args.sharded_one_rank:
        # the SHARDED code path (partition from local counts, limb reduction, one all-reduce per growth step, leaf-owner exchange) on a one-rank
        # RCCL communicator: prices everything of the N > 1 path but the wire, on the one GPU a builder's box has
        g.dist_init(g.dist_unique_id(), 0, 1)
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
    gd0 = g.array("GROW_DOCS").astype(np.float64)
    ds0 = g.dist_stats().astype(np.float64)
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
    ms_lam, n_lam, _ = g.timing("LAMBDA")
    gs = g.array("GROW_STATS")
    gd = (g.array("GROW_DOCS").astype(np.float64) - gd0) / max(args.steps, 1)      # documents per round: built, partitioned, Java-left, Java-split
    ds = (g.dist_stats().astype(np.float64) - ds0) / max(args.steps, 1)             # exchange of this rank per round: all-reduce calls / bytes, all-gather calls / bytes

    # ---- after the headline region: sustained rate over a long run, then a few rounds with events around every node-histogram launch
    sustained = None
    if args.sustain > 0:
        g.set_timing_flags(0)
        barrier()
        t1 = time.perf_counter()
        g.boost_rounds_async(args.sustain)
        g.sync()
        barrier()
        sustained = args.sustain / (time.perf_counter() - t1)
    node = None
    if args.node_rounds > 0 and not args.no_timing:
        g.set_timing_flags(N.RL_FLAG_TIMING | N.RL_FLAG_TIMING_NODES)
        g.reset_timing()
        gdn0 = g.array("GROW_DOCS").astype(np.float64)
        g.boost_rounds_async(args.node_rounds)
        g.sync()
        ms_node, n_node, _ = g.timing("HIST_NODE")
        built = (g.array("GROW_DOCS").astype(np.float64) - gdn0)[0]
        node = {"ms_per_round": ms_node / args.node_rounds, "launches_per_round": n_node / args.node_rounds,
                "docs_per_round": built / args.node_rounds}

    if rank != 0:
        return
    rounds_per_s = args.steps / elapsed
    out = {
        "metric": "boosting rounds/sec (LambdaMART -ranker 6, %s)" % ("MAP" if args.metric.upper() == "MAP" else args.metric.upper() + "@10"),
        "value": rounds_per_s,
        "unit": "rounds/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak" if weak else "strong",
        "vs_baseline": None,
        "dtype": "int64 fixed-point histograms + f64 lambdas/scores (f32 leaf chains as in the Java)",
        "data": "synthetic",
        "config": {
            "workload": "%s%s: synthetic %s, %d docs x %d features, %d queries (%s docs/query) in total, queries sharded "
                        "contiguously over %d GPU(s); LambdaMART -ranker 6, %d leaves, lr 0.1, -tc 256, -mls 1, NDCG@10" %
                        (args.shape, (" x %d (weak scaling: %d documents per GPU at every N)" % (world, n_docs // world)) if weak else "",
                         {"c2": "MSLR-WEB30K-shape (BASELINE.json configs[2], the shape the metric is quoted on)",
                          "c1": "MSLR-WEB10K-shape (BASELINE.json configs[1])"}.get(args.shape, args.shape),
                         n_docs, n_feat, q_total, "~120 log-normal" if kind == "mslr" else "5..15", world, n_leaves),
            "document_rounds_per_s": n_docs * rounds_per_s,      # the figure that aggregates over ranks under weak scaling
            "docs_total": n_docs, "docs_rank0": int(X.shape[0]), "features": n_feat, "queries_total": q_total, "leaves": n_leaves,
            "ndcg10_train_after_%d_rounds" % total_rounds: ndcg_t,
            "init_seconds": round(t_init, 3), "datagen_seconds": round(t_gen, 3),
            "growth_steps_per_tree": round(float(gs[0]) / max(int(gs[3]), 1), 2),
            "nodes_prepared_per_tree": round(float(gs[1]) / max(int(gs[3]), 1), 2),
            "splits_per_tree": round(float(gs[2]) / max(int(gs[3]), 1), 2),
        },
    }
    N_loc, F_, L_ = float(X.shape[0]), float(n_feat), float(n_leaves)
    T_ = float(g.bin_stride())
    rho_java, nu_java = gd[2] / max(float(n_docs), 1.0), gd[3] / max(float(n_docs), 1.0)      # global counts (every rank sees the same tree)
    rho_built, nu_part = gd[0] / max(float(n_docs), 1.0), gd[1] / max(float(n_docs), 1.0)

    def b_round(rho, nu, n):      # SURVEY.md 8d, b = 2 bytes per bin id
        return (n * F_ * 2 * (1 + rho) + n * 8 * (1 + rho) + n * 4 * rho + nu * n * (2 + 4 + 4) + n * 28 + n * 20 + n * 16 + n * 12 +
                (2 * L_ - 1) * F_ * T_ * 12)
    copy_gbs = read_gbs = gather_gbs = None
    if not args.plain:
        try:        # the box's own rates with this library's kernels (2 GiB copy, 4 GiB read / 32-byte row gathers of a quarter of the rows)
            ms_c, b_c = N.membench(0, 2 << 30, 1, 5, local_rank); copy_gbs = b_c / ms_c / 1e6
            ms_r, b_r = N.membench(1, 4 << 30, 1, 5, local_rank); read_gbs = b_r / ms_r / 1e6
            ms_g, b_g = N.membench(3, 4 << 30, 4, 5, local_rank); gather_gbs = b_g / ms_g / 1e6
        except Exception as ex:       # noqa: BLE001
            sys.stderr.write("membench failed: %r\n" % (ex,))
    lds_cal = None
    if not args.plain:
        try:        # LDS atomic throughput in the histogram kernels' own LDS layout: what "bound by LDS atomics" is a fraction of (rl_debug_membench modes 4..7)
            lds_cal = {}
            for name, mode in (("consecutive_bins", 4), ("random_bins", 5), ("same_address", 6), ("random_bins_plus_count", 7), ("three_32bit_atomics_instead", 8)):
                ms_l, n_at = N.membench(mode, 4096, 1, 3, local_rank)
                lds_cal[name] = n_at / (ms_l * 1e-3) / (256 * 2.4e9)
            lds_cal["unit"] = "atomic groups (one 64-bit atomic; + count; or three 32-bit ones) per CU and clock (256 CUs x 2.4 GHz), three 256-thread blocks per CU, 16 x 264 accumulators per block"
        except Exception as ex:       # noqa: BLE001
            sys.stderr.write("LDS atomic calibration failed: %r\n" % (ex,))
            lds_cal = None
    pmc = None if (args.no_pmc or world != 1) else live_pmc(args.shape)
    lds_atomics = None
    try:      # what the root pass is actually bound by (HISTORY.md 4.1): one ds_add_u64 per (document, feature) outside the feature's most populated bin
        cnt_cum, nb_ = g.array("ROOT_COUNT").astype(np.int64), g.array("NBINS")
        lds_atomics = 0.0
        for f_ in range(n_feat):
            per_bin = np.diff(np.concatenate([[0], cnt_cum[f_, :nb_[f_]]]))
            lds_atomics += float(per_bin.sum() - per_bin.max()) * (N_loc / max(float(n_docs), 1.0) if world > 1 else 1.0)
    except Exception as ex:       # noqa: BLE001
        sys.stderr.write("root-pass atomics count failed: %r\n" % (ex,))
    if n_root > 0:
        per_launch_ms = ms_root / n_root
        alg_bytes = bytes_root / n_root
        achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9
        B8d, Bbuilt = b_round(rho_java, nu_java, float(n_docs)), b_round(rho_built, nu_part, float(n_docs))
        root_traffic = (FETCH_FACTOR_WIDE * pmc["root_FETCH_SIZE"] + pmc["root_WRITE_SIZE"]) if pmc else None
        root_atoms_clk = (lds_atomics / (per_launch_ms * 1e-3) / (256 * 2.4e9)) if lds_atomics else None
        nonmode_frac = (lds_atomics / (N_loc * F_)) if lds_atomics else None          # share of the (document, feature) pairs outside the feature's most populated bin
        root_entry = {
            "kernel": "rl::k_hist<true,16> (root histogram, FeatureHistogram.update)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": root_traffic,
            # the bytes the kernel really moves (counter traffic) over its time: the fraction of peak that describes this build; `frac` above is
            # SURVEY.md 8d's definition (b = 2 bytes per bin id), more than the packed rows stream
            "frac_traffic": (root_traffic / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if root_traffic else None,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": per_launch_ms, "launches": int(n_root),
            "layout_bytes_per_launch": N_loc * (((n_feat + 15) // 16) * (18.0 if T_ <= 257 else 32.0) + 8.0),
            "lds_atomics_per_launch": lds_atomics, "lds_atomics_per_cu_clock": root_atoms_clk,
            "lds_atomics_frac_of_measured_peak": (root_atoms_clk / lds_cal["random_bins"]) if (root_atoms_clk and lds_cal) else None,
            "note": "algorithmic bytes = N_local*(F*2 B bin ids + 8 B fixed-point lambda) (SURVEY.md 8d, b = 2); layout bytes = what the packed rows hold; HIP events on the library stream",
            "frac_of_measured_read": (achieved / read_gbs) if read_gbs else None,
        }
        round_entry = {
            "rho_java": rho_java, "nu_java": nu_java, "rho_built": rho_built, "nu_partitioned": nu_part,
            "B_round_8d_bytes": B8d, "B_round_built_bytes": Bbuilt,
            "round_frac_8d": B8d * rounds_per_s / (HBM_PEAK_GBS * 1e9), "round_frac_built": Bbuilt * rounds_per_s / (HBM_PEAK_GBS * 1e9),
            "round_frac_built_of_measured_copy": (Bbuilt * rounds_per_s / (copy_gbs * 1e9)) if copy_gbs else None,
            "note": "B_round = N F b (1+rho) + 8 N (1+rho) + 4 N rho + nu N (b+4+4) + 76 N + (2L-1) F T 12 (SURVEY.md 8d, b = 2); round_frac = B_round x rounds/s / 8e12",
        }
        if node and node["ms_per_round"] > 0 and node["launches_per_round"] > 0:
            # the time-dominant kernel: the child-node histogram passes (the largest share of a round), per launch as the contract asks
            L_round = node["launches_per_round"]
            nb = node["docs_per_round"] * (F_ * 2 + 8 + 4)
            launch_ms = node["ms_per_round"] / L_round
            ach = (nb / L_round) / (launch_ms * 1e-3) / 1e9
            node_traffic_round = (FETCH_FACTOR_GATHER32 * pmc["node_FETCH_SIZE"] + pmc.get("node_WRITE_SIZE", 0.0)) if pmc and "node_FETCH_SIZE" in pmc else None
            node_atoms = (node["docs_per_round"] * F_ * nonmode_frac * 2.0) if nonmode_frac else None      # a 64-bit sum and a 32-bit count per pair outside the mode bin
            node_atoms_clk = (node_atoms / (node["ms_per_round"] * 1e-3) / (256 * 2.4e9)) if node_atoms else None
            out["roofline"] = {
                "kernel": "rl::k_hist<false,16> (child-node histograms, FeatureHistogram.construct): the kernel with the largest share of a round",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": (node_traffic_round / L_round) if node_traffic_round else None,
                "frac_traffic": (node_traffic_round / (node["ms_per_round"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if node_traffic_round else None,
                "traffic_over_algorithmic": (node_traffic_round / nb) if node_traffic_round else None,
                "algorithmic_bytes_per_launch": nb / L_round, "avg_launch_ms": launch_ms, "launches_per_round": L_round,
                "ms_per_round": node["ms_per_round"], "docs_accumulated_per_round": node["docs_per_round"], "algorithmic_bytes_per_round": nb,
                "frac_of_measured_gather32": (ach / gather_gbs) if gather_gbs else None,      # of THIS LIBRARY's own 32-byte row-gather kernel (rl_debug_membench mode 3), not of a hardware figure
                "lds_atomics_per_cu_clock": node_atoms_clk,
                # the calibration counts atomic GROUPS (here: a 64-bit sum + a 32-bit count = one pair) per CU and clock, so the fraction is pairs over
                # pairs.  (Rounds 3-4 divided atomics -- two per pair -- by the pair rate and reported twice this fraction: 0.74 where 0.37 was meant.)
                "lds_atomic_pairs_per_cu_clock": (node_atoms_clk / 2.0) if node_atoms_clk else None,
                "lds_atomics_frac_of_measured_peak": (node_atoms_clk / 2.0 / lds_cal["random_bins_plus_count"]) if (node_atoms_clk and lds_cal) else None,
                "traffic_source": ("live: bench.py re-ran itself under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only); bytes = "
                                   "%.1f x FETCH_SIZE + WRITE_SIZE (factors: profiles/r02_fetch_calibration.txt)" % FETCH_FACTOR_GATHER32) if pmc else None,
                "note": "algorithmic bytes (SURVEY.md 8d) = documents of the accumulated (smaller) children x (F*2 B bin ids + 8 B lambda + 4 B sample id); HIP events around "
                        "every growth step's launch over %d extra rounds (the empty launches of finished trees count as launches); the launches are short and mostly "
                        "latency-bound (HISTORY.md 4.1, 4.2): most of a step's time is neither bytes nor atomics" % args.node_rounds,
                "root_pass": root_entry,
            }
        else:       # (--no-timing / --node-rounds 0: only the root pass was timed)
            out["roofline"] = dict(root_entry)
        out["roofline"]["lds_atomic_calibration"] = lds_cal
        out["roofline"]["measured_copy_GBps"] = copy_gbs
        out["roofline"]["guide_copy_GBps"] = 6290.0       # float4 copy measured in /opt/skills/guides/MI355X_MICROARCH.md: what the fractions "of measured copy" should be read against if this library's copy kernel falls short of it
        out["roofline"]["measured_read_GBps"] = read_gbs
        out["roofline"]["measured_gather32_GBps"] = gather_gbs
        out["roofline"]["round"] = round_entry
        out["kernel_ms_per_round"] = {"hist_root": ms_root / args.steps, "lambda": ms_lam / args.steps,
                                      "hist_nodes": node["ms_per_round"] if node else None}
    if world > 1:
        out["config"]["exchange_per_round_rank0"] = {"allreduce_calls": ds[0], "allreduce_bytes": ds[1], "allgather_calls": ds[2], "allgather_bytes_received": ds[3],
                                                     "alltoall_calls": ds[4], "alltoall_bytes_received": ds[5], "tie_break_calls": ds[6], "tie_break_bytes_received": ds[7],
                                                     "allgather_of_every_lambda_would_be_bytes": 16.0 * n_docs,
                                                     "transport": "host callbacks over gloo (test aid)" if os.environ.get("RLHIP_BENCH_TRANSPORT") == "gloo" else "RCCL",
                                                     "note": "payload handed to the transport by this rank per round: histogram limbs per growth step (all-reduce); lambda / weight of the "
                                                             "leaves this rank owns, from the ranks that hold their documents (all-to-all); per-query metric values and leaf tables (all-gather)"}
        # DESIGN.md 6's latency model of a sharded round next to what this run measured: the day several GPUs run this line the model is tested.
        steps_tree = float(gs[0]) / max(int(gs[3]), 1)
        ar_bytes = ds[1] / max(ds[0], 1.0)
        out["config"]["scaling_model"] = dict(scaling_model(float(X.shape[0]), world, steps_tree, ar_bytes), measured_ms_per_round=1000.0 * elapsed / args.steps)
        if not weak:
            out["config"]["scaling_note"] = ("strong scaling of %d documents is latency-bound: a 31-leaf tree is a chain of ~11 dependent growth steps, each with one "
                                             "all-reduce when sharded, whatever the shard size (DESIGN.md 6); --scaling weak keeps the documents per GPU fixed" % n_docs)
    if world == 1 and args.shape == "c2":
        # what DESIGN.md 6's model says about 2 / 4 / 8 GPUs -- MODELLED, printed so that the first multi-GPU run has something to be held against
        out["config"]["scaling_model"] = {"strong_%d" % n: scaling_model(n_docs / n, n) for n in (2, 4, 8)}
        out["config"]["scaling_model"].update({"weak_%d" % n: scaling_model(float(n_docs), n) for n in (2, 4, 8)})
        out["config"]["scaling_model"]["note"] = ("MODELLED, never measured (no multi-GPU box): strong = this data set over n GPUs, weak = this many documents PER GPU; "
                                                  "rounds/s of the whole job; aggregate document-rounds/s under weak scaling = n x docs_per_rank x rounds/s")
    if sustained is not None:
        out["config"]["sustained_rounds_per_s"] = sustained
        out["config"]["sustained_over_rounds"] = args.sustain
    bb = g.array("BUBBLES")
    out["config"]["host_bubbles"] = {
        "what": "device wall-clock time the main stream idled behind a host decision, averaged over the whole run of this trainer (RL_ARR_BUBBLES): "
                "the bookkeeping that ends a tree -> first instruction of the leaf table (one empty growth step was already enqueued); the leaf chain's last "
                "stitch -> the leaf outputs (the host looks at the stitch's result before it enqueues anything else)",
        "tree_end_to_leaf_table_us": 0.01 * float(bb[0]) / max(int(bb[1]), 1), "stitch_to_leaf_output_us": 0.01 * float(bb[2]) / max(int(bb[3]), 1), "trees": int(bb[1])}
    ts = g.array("TIE_STATS")
    out["config"]["tie_break"] = {
        "mode": "lazy Java-order (exact ties re-decided in the reference's summation order, HISTORY.md 4.13; sharded runs gather the chain nodes and do the same)"
                if (not args.java_order and not args.first_tie) else
                ("strict: every candidate from the Java-order histogram" if args.java_order else "first candidate in scan order (--first-tie)"),
        "resolutions": int(ts[0]), "nodes": int(ts[1]), "chain_documents": int(ts[3]), "host_ms": float(ts[4]) / 1e3,
        "note": "whole run of this trainer (warm-up, timed, sustained and per-step-timing rounds)"}
    if world == 1 and args.c1_trees > 0 and not args.java_order:
        # BASELINE.json configs[1] as stated: MSLR-WEB10K shape, 1000 trees, 31 leaves, one GPU -- the whole run, not a window of it
        n1, f1, k1, _, l1 = synth.SHAPES["c1"]
        X1, lab1, qoff1, q1 = synth.make_shard(n1, f1, k1, 0, 1)
        g1 = N.Trainer(n_trees=args.c1_trees, n_leaves=l1, device=local_rank)
        g1.set_train(X1, lab1, qoff1)
        g1.init()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        g1.boost_rounds_async(args.c1_trees)
        g1.sync()
        dt1 = time.perf_counter() - t1
        ts1 = g1.array("TIE_STATS")
        out["config"]["c1_full_run"] = {"workload": "configs[1]: %d docs x %d features, %d queries, %d trees x %d leaves, one GPU" % (n1, f1, q1, args.c1_trees, l1),
                                        "rounds_per_s": args.c1_trees / dt1, "seconds": dt1, "ndcg10_train": float(g1.round_metrics(args.c1_trees - 1)[0]),
                                        "tie_resolutions": int(ts1[0]), "tie_host_ms": float(ts1[4]) / 1e3}
        del g1
        # ... and once more with a held-out set (a fifth of the shape's size, rl_set_validation, early stopping off): NDCG@10 on train AND held-out of the
        # rolled-back model (LambdaMART.java:253-265), next to what the CPU oracle ended with on the same inputs (tools/long_parity.py follows the two
        # side by side for all 1000 rounds -- ~25 minutes of box time -- and its last line is committed)
        Xh, labh, qh = synth.make_heldout("c1")
        g1 = N.Trainer(n_trees=args.c1_trees, n_leaves=l1, device=local_rank, early_stop_rounds=1 << 30)
        g1.set_train(X1, lab1, qoff1)
        g1.set_validation(Xh, labh, qh)
        g1.init()
        for _ in range(args.c1_trees):          # (a validation set: round by round -- the early-stop decision is the host's, LambdaMART.java:240-250)
            g1.boost_round(want_tree=False)
        tr_g, ho_g = g1.finish()
        lp = {}
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_long_parity_c1.json")) as fh:
                lp = json.loads(fh.read())
        except Exception:       # noqa: BLE001
            lp = {}
        same_run = bool(lp) and lp.get("rounds") == args.c1_trees
        out["config"]["c1_heldout_run"] = {
            "workload": "configs[1] with %d held-out documents in %d lists passed through rl_set_validation, %d trees, early stopping off" % (len(labh), len(qh) - 1, args.c1_trees),
            "ndcg10_train_gpu": tr_g, "ndcg10_heldout_gpu": ho_g, "trees_kept_gpu": g1.num_trees(),
            "ndcg10_train_oracle": lp.get("ndcg10_train_oracle") if same_run else None, "ndcg10_heldout_oracle": lp.get("ndcg10_heldout_oracle") if same_run else None,
            "first_divergent_round": lp.get("first_divergent_round") if same_run else None, "rounds_compared_identical": lp.get("rounds_compared_identical") if same_run else None,
            "splits_compared": lp.get("splits_compared") if same_run else None, "splits_storing_another_candidate": lp.get("splits_storing_another_candidate") if same_run else None,
            "oracle_source": "CACHED figures: profiles/r05_long_parity_c1.json, logged by tools/long_parity.py c1 1000 at commit f0857ba (GPU and oracle side by side, every round compared); "
                             "they describe this run only while ranklib_amd/synth.py generates the same data -- a non-zero |diff| below means the GPU side or the data changed, not that the oracle was re-run" if same_run else None,
            "heldout_abs_diff_vs_oracle": (abs(ho_g - lp["ndcg10_heldout_oracle"]) if same_run else None),
            "train_abs_diff_vs_oracle": (abs(tr_g - lp["ndcg10_train_oracle"]) if same_run else None)}
        del g1, Xh

    if world == 1 and args.c2_trees > 0 and args.shape == "c2" and not args.java_order and not args.sharded_one_rank:
        # BASELINE.json configs[2]'s shape as a WHOLE run on one GPU (the headline is a window of early trees; late trees take twice the growth steps)
        g2 = N.Trainer(n_trees=args.c2_trees, n_leaves=n_leaves, device=local_rank)
        g2.set_train(X, lab, qoff)
        g2.init()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        g2.boost_rounds_async(args.c2_trees)
        g2.sync()
        dt2 = time.perf_counter() - t1
        out["config"]["c2_full_run"] = {"workload": "%d docs x %d features, %d queries, %d trees x %d leaves, one GPU, the whole run" % (n_docs, n_feat, q_total, args.c2_trees, n_leaves),
                                        "rounds_per_s": args.c2_trees / dt2, "seconds": dt2, "ndcg10_train": float(g2.round_metrics(args.c2_trees - 1)[0])}
        del g2
    if world == 1 and args.shard1_rounds > 0 and args.shape == "c2" and not args.java_order and not args.sharded_one_rank:
        # the SHARDED code path (what every rank of an N > 1 job runs: partition from local counts, limb reduction, one all-reduce per growth step,
        # leaf-owner exchange, gathered metric) on a one-rank RCCL communicator, beside the plain path on the same box: everything of the multi-GPU
        # path but the wire.  The trees must be the plain path's (checked on the train metric of the last round).
        def timed(sharded):
            gs = N.Trainer(n_trees=5 + args.shard1_rounds, n_leaves=n_leaves, device=local_rank)
            gs.set_train(X, lab, qoff)
            if sharded:
                gs.dist_init(gs.dist_unique_id(), 0, 1)
            gs.init()
            gs.boost_rounds_async(5); gs.sync()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            gs.boost_rounds_async(args.shard1_rounds); gs.sync()
            dts = time.perf_counter() - t1
            gst = gs.array("GROW_STATS")
            return args.shard1_rounds / dts, float(gs.round_metrics(5 + args.shard1_rounds - 1)[0]), float(gst[0]) / max(int(gst[3]), 1)
        try:
            r_plain, m_plain, _ = timed(False)
            r_shard, m_shard, st_shard = timed(True)
            out["config"]["sharded_path_one_rank"] = {
                "what": "the sharded code path on a one-rank RCCL communicator at this shape, %d rounds after 5, same box, same process" % args.shard1_rounds,
                "sharded_path_one_rank_rounds_per_s": r_shard, "plain_path_rounds_per_s": r_plain, "ratio": r_shard / r_plain,
                "growth_steps_per_tree": round(st_shard, 2), "same_train_metric_as_plain_path": bool(m_plain == m_shard)}
            out["config"]["sharded_path_one_rank_rounds_per_s"] = r_shard
        except Exception as ex:      # noqa: BLE001  (librccl missing on the box: the field says so instead of a number)
            out["config"]["sharded_path_one_rank"] = {"error": repr(ex)[:300]}

    if world == 1 and args.ns_rounds > 0 and args.shape == "c2" and not args.java_order:
        # the north star's own list length ("~10 docs/query") at the same size: SURVEY.md 8d asks for both variants
        nn, fn, kn, _, ln = synth.SHAPES["c2ns"]
        Xn, labn, qoffn, qn = synth.make_shard(nn, fn, kn, 0, 1)
        gn = N.Trainer(n_trees=5 + args.ns_rounds + 100, n_leaves=ln, device=local_rank)
        gn.set_train(Xn, labn, qoffn)
        gn.init()
        gn.boost_rounds_async(5); gn.sync()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gn.boost_rounds_async(args.ns_rounds); gn.sync()
        dtn = time.perf_counter() - t1
        t1 = time.perf_counter()
        gn.boost_rounds_async(100); gn.sync()
        dts = time.perf_counter() - t1
        out["config"]["c2ns"] = {"workload": "c2ns: the same %d docs x %d features with the north star's list length (5..15 docs/query, %d queries), %d leaves, one GPU" % (nn, fn, qn, ln),
                                 "rounds_per_s": args.ns_rounds / dtn, "steps": args.ns_rounds, "warmup": 5, "sustained_rounds_per_s_next_100": 100 / dts,
                                 "ndcg10_train": float(gn.round_metrics(5 + args.ns_rounds - 1)[0])}
        del gn, Xn

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
    emit(json.dumps(out))


if __name__ == "__main__":
    main()
