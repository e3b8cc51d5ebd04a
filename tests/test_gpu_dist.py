"""-m gpu: multi-rank (sharded) training == single-GPU training, bit for bit.

Only one GPU is available to the tests, so k ranks run as k processes on the SAME device and exchange through the
host-callback transport over gloo (rl_dist_init_callback).  Every rank executes exactly the code path an RCCL rank
executes (k_hist_reduce -> all-reduce of int64 limbs -> k_hist_finish<.,true>, global thresholds, local/global
node counts, gathered float chains); only the transport differs.  The RCCL transport itself is exercised with a
1-rank communicator.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from ranklib_amd import _native as N
from ranklib_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def single(n_docs, n_feat, kind, seed, leaves, rounds, dist_mode=None, ranker="LAMBDAMART", metric="NDCG", k=10, opts=()):
    X, lab, qoff = synth.make_dataset(n_docs, n_feat, kind, seed_offset=seed)
    if "dupcols" in opts:    # as tests/dist_worker.py: the second half of the columns repeats the first (other thresholds, the same cuts)
        X = X.copy(); h = n_feat // 2; X[:, h:2 * h] = 2.0 * X[:, :h] + 1.0
    # default flags on both sides: sharded runs re-decide exact ties in the Java's summation order as well (the members' values are gathered,
    # every rank evaluates the same global chains) -- k shards must equal one shard in the stored (feature, threshold) pairs too
    g = N.Trainer(n_trees=rounds, n_leaves=-1 if "leafm1" in opts else leaves, ranker=ranker, metric=metric, metric_k=k,
                  min_leaf_support=40 if "leafm1" in opts else 1, n_threshold=-1 if "tcm1" in opts else 256)
    g.set_train(X, lab, qoff)
    if "qrel" in opts:       # as tests/dist_worker.py
        qi = np.arange(len(qoff) - 1)
        g.set_external_judgments(False, ideal_dcg=np.where(qi % 3 == 0, 10.0 + (qi % 7), np.nan), rel_doc_count=(qi % 4).astype(np.int32))
    if dist_mode == "rccl1":
        g.dist_init(g.dist_unique_id(), 0, 1)
    elif dist_mode == "cb1":
        g.dist_init_callback(0, 1, lambda arr, op: None, lambda src: src.copy())
    g.init()
    trees, mets = [], []
    for _ in range(rounds):
        t, tm, _, _ = g.boost_round()
        trees.append(t.trimmed()); mets.append(float(tm))
    final, _ = g.finish()
    return trees, mets, g.array("SCORE"), final


def same(a, b):
    ta, ma, sa, fa = a
    tb, mb, sb, fb = b
    assert ma == mb and fa == fb
    assert np.array_equal(sa.view(np.int64), sb.view(np.int64))
    for x, y in zip(ta, tb):
        for k in ("feature", "left", "right", "count"):
            assert np.array_equal(x[k], y[k]), k
        assert np.array_equal(x["deviance"].view(np.int64), y["deviance"].view(np.int64))      # exact sums of lambda and lambda^2: rank-count invariant
        assert np.array_equal(x["threshold"].view(np.uint32), y["threshold"].view(np.uint32))
        assert np.array_equal(x["output"].view(np.uint32), y["output"].view(np.uint32))


CFG = (9000, 24, "mslr", 3, 12, 5)


def test_one_rank_distributed_paths_equal_plain_path():
    ref = single(*CFG)
    same(ref, single(*CFG, dist_mode="cb1"))       # host-callback transport, 1 rank
    same(ref, single(*CFG, dist_mode="rccl1"))     # RCCL transport (dlopen'ed librccl), 1-rank communicator


def test_one_rank_sharded_path_beyond_one_block_row_of_chunks():
    """Round 6: at 9 000 documents a growth step has fewer chunks than a balanced step starts at (28), and a one-rank communicator that balanced
    its chunks while k_hist_reduce cut the nodes by the per-node rule passed every test -- and grew other trees from 60 000 documents on (bench.py
    --sharded-one-rank: NDCG@10 0.56 against 0.66 at c2; tools/dist_scale_check.py).  Here: 120 000 x 24, 31 leaves, steps of ~100 balanced chunks, both
    transports; the sharded path must run the plain path's trees."""
    big = (120000, 24, "mslr", 3, 31, 3)
    ref = single(*big)
    same(ref, single(*big, dist_mode="rccl1"))
    same(ref, single(*big, dist_mode="cb1"))


CFG2K = (16384, 24, "ns", 3, 12, 4)       # shards of 8199 / 8185 documents: different ceil(log2(N + 1)), one lambda^2 scale for all ranks


def test_shard_sizes_straddle_a_power_of_two():
    from ranklib_amd import dist as D
    X, lab, qoff = synth.make_dataset(*CFG2K[:3], seed_offset=CFG2K[3])
    sizes = [D.shard(X, lab, qoff, r, 2)[0].shape[0] for r in range(2)]
    assert min(sizes) < 8191 < max(sizes), sizes


CFG_TIES = (2500, 5, "mslr", 4, 31, 4)    # small nodes, five features: exact ties in every round -- the sharded lazy tie-break has to run (and to agree with one GPU)
CFG31 = (9000, 24, "mslr", 4, 31, 4)      # 30 growth steps allowed, trees finish after ~10: the ranks must stop enqueuing at the same step
CFG_BIG = (150000, 24, "mslr", 5, 31, 3)  # shards of 50-75 k documents: steps of many balanced chunks per rank, local left sizes from the ranks' own cumulative counts (round 6)


@pytest.mark.parametrize("world,ranker,metric,k,cfg", [(2, "LAMBDAMART", "NDCG", 10, CFG), (3, "LAMBDAMART", "NDCG", 10, CFG),
                                                        (2, "MART", "NDCG", 10, CFG), (2, "LAMBDAMART", "MAP", 0, CFG),
                                                        (3, "LAMBDAMART", "ERR", 10, CFG), (2, "LAMBDAMART", "NDCG", 10, CFG31),
                                                        (3, "MART", "NDCG", 10, CFG31), (2, "LAMBDAMART", "NDCG", 10, CFG2K),
                                                        (2, "LAMBDAMART", "NDCG", 10, CFG_TIES), (3, "LAMBDAMART", "NDCG", 10, CFG_TIES),
                                                        (2, "LAMBDAMART", "NDCG", 10, CFG_BIG), (3, "LAMBDAMART", "NDCG", 10, CFG_BIG)])
def test_k_shards_equal_one_shard(world, ranker, metric, k, cfg, tmp_path):
    CFG = cfg
    ref = single(*CFG, ranker=ranker, metric=metric, k=k)
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29511 + world), os.path.join(ROOT, "tests", "dist_worker.py"), out] + [str(v) for v in CFG] + [ranker, metric, str(k)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    z = np.load(out)
    rounds = CFG[5]
    trees = [{k: z["t%d_%s" % (i, k)] for k in ("feature", "threshold", "left", "right", "output", "deviance", "count")} for i in range(rounds)]
    same(ref, (trees, [float(v) for v in z["mets"]], z["scores"], float(z["final"])))
    # round 6, distributed float chains (rl_dist.inc "piece mode"): NO document's lambda / weight leaves its rank -- every rank evaluates its own pieces
    # of every leaf and only per-piece tables (260 bytes per leaf, value array and rank), totals and drifts are all-gathered: the bytes of a round do
    # not grow with the documents (the leaf-owner exchange that remains for > 256 leaves is covered by test_sharded_options[ownerx])
    st = z["dist_stats"].astype(np.float64)
    if cfg is CFG_TIES:      # the tie-break ran sharded: resolutions, and exchanges of the chain nodes' values counted apart from the per-round pattern
        assert z["tie_stats"][0] > 0 and st[6] > 0 and st[7] > 0, (z["tie_stats"], st)
    assert st[4] == 0 and st[5] == 0, st
    if CFG[0] >= 100000:     # (st[3] also holds rl_init's one-off exchange of the distinct-value sets)
        assert st[3] / rounds <= 0.25 * 16.0 * CFG[0] * (world - 1) / world, st
    # (st[3], the all-gather bytes, also holds rl_init's one-off exchange of the distinct-value sets; the per-round figure is checked through
    # bench.py's counters in test_bench_entry_starts_its_own_ranks)


@pytest.mark.parametrize("world,metric,k,opt", [(2, "NDCG", 10, "noa2a"), (3, "NDCG", 10, "piecemiss"), (2, "NDCG", 10, "countpass,ownerx"), (3, "NDCG", 10, "ownerx"), (2, "NDCG", 10, "ownerx,noa2a"), (3, "NDCG", 10, "leafm1"), (2, "NDCG", 10, "qrel"), (3, "MAP", 0, "qrel"),
                                                (2, "NDCG", 10, "dupcols"), (3, "NDCG", 10, "dupcols,regrow"), (2, "NDCG", 10, "tcm1"), (3, "NDCG", 10, "tcm1")])
def test_sharded_options(world, metric, k, opt, tmp_path):
    """ownerx: the leaf-owner exchange (RLHIP_DIST_OWNER_CHAINS=1; what runs beyond 256 leaves) instead of the distributed float chains: lambda / weight of a
    leaf's documents travel to ONE owner rank -- at most 16 bytes per document that lives elsewhere, one all-to-all a round;
    noa2a: a host transport WITHOUT an all-to-all (the exchange is emulated with all-gathers); leafm1: -leaf -1 (the leaf budget comes from the
    GLOBAL document count); qrel: external relevance judgments, every rank passing the entries of its own lists; dupcols: duplicated columns -- ties
    over several features that share one cut are deferred to the per-tree batch, every rank checks the cuts on its own documents and the verdict
    is all-reduced (regrow: forced to fail, all ranks grow the tree again); tcm1: -tc -1, threshold tables of more than 4095 entries (the ranks merge
    their distinct values at rl_init and split the tables into the same virtual features: LambdaMART.java:135-140 has no limit) -- each equals the one-GPU run"""
    cfg = (9000, 16, "mslr", 5, 10, 4)
    ref = single(*cfg, metric=metric, k=k, opts=tuple(opt.split(",")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if "regrow" in opt:
        env["RLHIP_TIE_FORCE_REGROW"] = "1"
    if "ownerx" in opt:
        env["RLHIP_DIST_OWNER_CHAINS"] = "1"
    if "countpass" in opt:       # round 5's sharded partition (count pass + two-pass scatter, chunks by the per-node rule) instead of the single pass from local cumulative counts
        env["RLHIP_DIST_COUNT_PASS"] = "1"
    if "piecemiss" in opt:       # every piece of a leaf's chain behind the first is treated as a detected window miss: its rank re-evaluates it from the exact start
        env["RLHIP_PIECE_FORCE_MISS"] = "1"
    out = str(tmp_path / "o.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29551 + world), os.path.join(ROOT, "tests", "dist_worker.py"), out] + [str(v) for v in cfg] + ["LAMBDAMART", metric, str(k), opt]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    z = np.load(out)
    trees = [{kk: z["t%d_%s" % (i, kk)] for kk in ("feature", "threshold", "left", "right", "output", "deviance", "count")} for i in range(cfg[5])]
    same(ref, (trees, [float(v) for v in z["mets"]], z["scores"], float(z["final"])))
    if "piecemiss" in opt:
        assert z["piece_stats"][0] >= cfg[5] * (world - 1) and z["piece_stats"][1] > 0, z["piece_stats"]        # the repair loop ran: world - 1 rounds a tree at least
    if "ownerx" in opt:
        st = z["dist_stats"].astype(np.float64)
        assert st[4] == cfg[5] + int(z["tie_stats"][9]) and st[5] > 0, st          # one leaf-owner exchange per round (one more for a tree grown a second time)
        assert st[5] / st[4] <= 16.0 * cfg[0] * (world - 1) / world * 0.95, st
    if opt == "leafm1":
        assert max(len(t["feature"]) for t in trees) > 2 * cfg[4] - 1, "the trees never outgrew the explicit leaf budget: -leaf -1 was not exercised"


def test_bench_entry_starts_its_own_ranks():
    """`python bench.py --gpus 2` AS TYPED (no launcher around it): the script re-executes itself under torch.distributed.run with two ranks.  Both
    share the one GPU of the test box through the host-callback transport (RLHIP_BENCH_TRANSPORT=gloo RLHIP_BENCH_SAME_GPU=1; RCCL refuses
    two ranks on one device) -- the N > 1 path of the driver's scaling run, byte counters included."""
    import json
    env = dict(os.environ, RLHIP_BENCH_TRANSPORT="gloo", RLHIP_BENCH_SAME_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    for extra in ([], ["--scaling", "weak"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--shape", "c0", "--plain"] + extra,
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1 and lines[0].startswith("{"), "stdout must hold the one JSON line and nothing else: " + r.stdout[-2000:]
        o = json.loads(lines[0])
        assert o["n_gpus"] == 2 and o["value"] > 0 and o["scaling"] == ("weak" if extra else "strong")
        ex = o["config"]["exchange_per_round_rank0"]
        assert ex["alltoall_calls"] == 0 and ex["alltoall_bytes_received"] == 0       # round 6: distributed float chains -- no document's lambda leaves its rank
        assert ex["allgather_bytes_received"] < 0.5 * ex["allgather_of_every_lambda_would_be_bytes"]      # leaf tables, piece totals / drifts / tables, per-query metric values
        assert o["config"]["docs_total"] == (20000 if extra else 10000)


def test_bench_stdout_is_one_json_line_with_an_rccl_communicator_in_the_process():
    """RCCL prints a version banner through the C library's buffered stdout when a communicator is created -- it used to land BEHIND the JSON line when
    the process ended (the default run creates a one-rank communicator for config.sharded_path_one_rank).  bench.py points file descriptor 1 at stderr
    for its whole life and writes the line to the real stdout itself: stdout holds the one line, nothing else."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--shape", "c0", "--plain", "--sharded-one-rank"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), "stdout must hold the one JSON line and nothing else: " + r.stdout[-2000:]
    o = json.loads(lines[0])
    assert o["n_gpus"] == 1 and o["value"] > 0


def _run_workers(world, cfg, out, extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), out] + [str(v) for v in cfg] + ["LAMBDAMART", "NDCG", "10", extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return np.load(out, allow_pickle=True)


@pytest.mark.parametrize("world", [2, 3])
def test_validation_set_under_sharding(world, tmp_path):
    """training AND validation lists sharded over the ranks (learning/tree/LambdaMART.java:228-250 has no such notion: one JVM): per-round
    validation metric, early stop, rollback and the kept trees equal the one-GPU run's"""
    cfg = (9000, 16, "ns", 9, 8, 30)
    X, lab, qoff = synth.make_dataset(cfg[0], cfg[1], cfg[2], seed_offset=cfg[3])
    Xv, lv, qv = synth.make_dataset(cfg[0] // 3, cfg[1], cfg[2], seed_offset=cfg[3] + 77)
    lv = lv[::-1].copy()                      # as tests/dist_worker.py does
    g = N.Trainer(n_trees=cfg[5], n_leaves=cfg[4], early_stop_rounds=1)
    g.set_train(X, lab, qoff); g.set_validation(Xv, lv, qv); g.init()
    mets, vmets = [], []
    for _ in range(cfg[5]):
        t, tm, vm, stop = g.boost_round()
        mets.append(float(tm)); vmets.append(float(vm))
        if stop:
            break
    final, vfinal = g.finish()
    z = _run_workers(world, cfg, str(tmp_path / "v.npz"), "valid", 29531 + world)
    assert [float(v) for v in z["mets"]] == mets and [float(v) for v in z["vmets"]] == vmets
    assert float(z["final"]) == final and float(z["vfinal"]) == vfinal and int(z["kept"]) == g.num_trees()
    assert len(mets) < cfg[5], "the early stop never fired: the case does not test the rollback"
    st = z["dist_stats"]
    assert st[0] > 0 and st[2] > 0


def test_rccl_transport_with_two_ranks():
    """ncclCommInitRank with N = 2 through the dlsym'ed entry points (128-byte ncclUniqueId by value, rl_dist.inc).  The test box has ONE
    GPU: if RCCL accepts two ranks on it the sharded run must equal the one-GPU run; if it refuses ("Duplicate GPU"), the refusal must
    arrive as a clean RL_ERR_COMM on every rank -- no crash, no hang -- which is all a one-GPU box can prove about the call."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        z = _run_workers(2, CFG, os.path.join(d, "r.npz"), "rccl", 29541)
        if "refused" in z:
            msg = str(z["refused"])
            assert "ncclCommInitRank" in msg, msg
            print("RCCL refused two ranks on one device:", msg)
        else:
            ref = single(*CFG)
            trees = [{k: z["t%d_%s" % (i, k)] for k in ("feature", "threshold", "left", "right", "output", "deviance", "count")} for i in range(CFG[5])]
            same(ref, (trees, [float(v) for v in z["mets"]], z["scores"], float(z["final"])))


def test_rccl_transport_one_rank_per_device():
    """The real multi-GPU data path, whenever the box has it: world = visible devices (at most 8), one rank per GPU, the library's own RCCL
    communicator (histogram all-reduce per growth step, leaf-owner all-to-all, gathers).  The sharded run must equal the one-GPU run bit for
    bit -- trees, per-round metrics, scores -- with the default flags (lazy tie-break included) and with a validation set.  Skipped on a one-GPU
    box, where test_rccl_transport_with_two_ranks is all that can be proven; on an 8-GPU node this, not the scaling bench, is the first
    execution of an N > 1 collective."""
    ndev = N.device_count()
    if ndev < 2:
        pytest.skip("one GPU visible: RCCL needs a device per rank")
    world = min(ndev, 8)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        z = _run_workers(world, CFG, os.path.join(d, "r.npz"), "rccl,perdev", 29561)
        assert "refused" not in z, str(z.get("refused"))
        ref = single(*CFG)
        trees = [{k: z["t%d_%s" % (i, k)] for k in ("feature", "threshold", "left", "right", "output", "deviance", "count")} for i in range(CFG[5])]
        same(ref, (trees, [float(v) for v in z["mets"]], z["scores"], float(z["final"])))
        st = z["dist_stats"]
        assert st[0] > 0 and st[1] > 0, "no all-reduce went through the RCCL communicator: %s" % st
        # a tie-heavy data set (duplicated columns: ties over several features that share one cut) and -leaf -1 as well
        z2 = _run_workers(world, CFG, os.path.join(d, "r2.npz"), "rccl,perdev,dupcols", 29563)
        ref2 = single(*CFG, opts=("dupcols",))
        trees2 = [{k: z2["t%d_%s" % (i, k)] for k in ("feature", "threshold", "left", "right", "output", "deviance", "count")} for i in range(CFG[5])]
        same(ref2, (trees2, [float(v) for v in z2["mets"]], z2["scores"], float(z2["final"])))
