"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): bit-exact for integer / index work (thresholds, bins, counts, split
feature + threshold, tree shape); leaf values and scores are compared BIT-EXACT as well (the path emulates
the Java float chains), which is stricter than the 1e-5 the north star asks for; NDCG within 1e-4 is then
implied, and asserted with == on the float-accumulated per-round value.
"""
from fractions import Fraction

import os

import numpy as np
import pytest

import oracle_ffi as O
from ranklib_amd import _native as N
from ranklib_amd import synth
from tree_equiv import assert_equivalent

pytestmark = pytest.mark.gpu


def make(n_docs, n_feat, kind="ns", seed=0):
    return synth.make_dataset(n_docs, n_feat, kind, seed_offset=seed)


def pair(X, lab, qoff, **kw):
    args = dict(n_trees=kw.get("n_trees", 5), n_leaves=kw.get("n_leaves", 10))
    o = O.Oracle(X, lab, qoff, lr=kw.get("lr", 0.1), n_threshold=kw.get("n_threshold", 256), mls=kw.get("mls", 1),
                 k=kw.get("k", 10), early_stop=kw.get("early_stop", 100), qkey=kw.get("qkey"), **args)
    g = N.Trainer(learning_rate=kw.get("lr", 0.1), n_threshold=kw.get("n_threshold", 256),
                  min_leaf_support=kw.get("mls", 1), metric_k=kw.get("k", 10),
                  early_stop_rounds=kw.get("early_stop", 100), **args)
    g.set_train(X, lab, qoff, qkey=kw.get("qkey"))
    return o, g


STATS = {}


def assert_same_tree(to, tg, X, ctx=""):
    """bit-exact, stored (feature, threshold) pairs included"""
    ties = assert_equivalent(to, tg, X, ctx, STATS)
    # the default path on one GPU without feature sampling: exact ties are re-decided in the Java's summation order (rl_tie.inc), so the stored
    # (feature, threshold) pairs are the oracle's -- no split may be resolved differently (tests with feature sampling call assert_equivalent)
    assert ties == 0, "%d split(s) store another (feature, threshold) than the oracle's in %s (%s)" % (ties, os.environ.get("PYTEST_CURRENT_TEST", "?"), ctx)
    return ties


def teardown_module(module):
    if STATS:
        print("\n[parity] splits compared: %d, stored (feature, threshold) pairs that differ from the oracle's: %d"
              % (STATS.get("splits", 0), STATS.get("plateau", 0)))


def test_init_thresholds_bins_counts_bit_exact():
    X, lab, qoff = make(5000, 24, "ns", 1)
    o, g = pair(X, lab, qoff)
    o.init(); g.init()
    nb = g.array("NBINS")
    thr = g.array("THRESHOLDS")
    bins = g.array("BINS")
    cnt = g.array("ROOT_COUNT")
    for f in range(24):
        T = o.n_bins(f)
        assert nb[f] == T
        assert np.array_equal(thr[f, :T].view(np.uint32), o.thresholds(f).view(np.uint32))
        assert np.array_equal(bins[f].astype(np.int32), o.bins(f))
        assert np.array_equal(cnt[f, :T], o.root_count(f))
    assert nb.max() == 257          # continuous features: the 257th bin (H4)


def test_lambdas_weights_bit_exact_round0_and_later():
    X, lab, qoff = make(6000, 16, "mslr", 2)
    o, g = pair(X, lab, qoff, n_trees=3, n_leaves=8)
    o.init(); g.init()
    for r in range(3):
        o.round(); g.boost_round()
        assert np.array_equal(g.array("LAMBDA").view(np.int64), o.lambdas().view(np.int64)), r
        assert np.array_equal(g.array("WEIGHT").view(np.int64), o.weights().view(np.int64)), r
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r


def test_root_histogram_is_exact_fixed_point_sum():
    X, lab, qoff = make(20000, 12, "ns", 3)
    o, g = pair(X, lab, qoff, n_trees=1, n_leaves=2)
    o.init(); g.init()
    g.boost_round(); o.round()
    lam = g.array("LAMBDA")
    E = g.quant_exponent()
    q = g.array("QUANT")
    assert np.array_equal(q, np.rint(np.ldexp(lam, E)).astype(np.int64))
    assert 2 ** 47 <= np.abs(q).max() < 2 ** 49      # 62 - log2(chunk) bits used
    bins = g.array("BINS")
    fixed = g.array("ROOT_SUM_FIXED")
    sums = g.array("ROOT_SUM")
    nb = g.array("NBINS")
    for f in range(12):
        per_bin = [0] * nb[f]
        for k in range(len(q)):
            per_bin[bins[f, k]] += int(q[k])
        run = 0
        for t in range(nb[f]):
            run += per_bin[t]
            hi, lo = int(fixed[f, t, 0]), int(fixed[f, t, 1]) & 0xFFFFFFFFFFFFFFFF
            assert hi * 2 ** 64 + lo == run, (f, t)      # two's complement: signed hi, unsigned lo
            assert sums[f, t] == float(Fraction(run, 2 ** E))
            # and it agrees with the Java-order f64 sum up to rounding noise
            assert abs(sums[f, t] - o.root_sum(f)[t]) <= 1e-9 * max(1.0, np.abs(lam).sum())


@pytest.mark.parametrize("n_docs,n_feat,kind,leaves,mls,rounds,seed", [
    (3000, 10, "ns", 10, 1, 8, 0),
    (8000, 136, "ns", 10, 1, 6, 1),       # c0 shape
    (12000, 20, "mslr", 31, 1, 5, 2),
    (4000, 8, "ns", 6, 25, 6, 3),
    (2500, 5, "mslr", 31, 1, 4, 4),
])
def test_trees_scores_metrics_bit_exact(n_docs, n_feat, kind, leaves, mls, rounds, seed):
    X, lab, qoff = make(n_docs, n_feat, kind, seed)
    o, g = pair(X, lab, qoff, n_trees=rounds, n_leaves=leaves, mls=mls)
    o.init(); g.init()
    for r in range(rounds):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert_same_tree(to, tg, X, "round %d trace %s" % (r, o.split_trace()))
        assert np.float32(tmo).view(np.uint32) == np.float32(tmg).view(np.uint32), r
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r
    so, _ = o.finish()
    sg, _ = g.finish()
    assert so == sg
    assert np.array_equal(g.predict(X[:777]).view(np.uint32), o.predict(X[:777]).view(np.uint32))
    if n_docs == 2500:      # this data set's small nodes hold exact ties in most rounds: the lazy tie-break must have run
        st = g.array("TIE_STATS")
        assert st[0] > 0 and st[1] >= st[0] and st[2] >= st[1], st


def _eval_tree_rows(tr, X):
    out = np.zeros(X.shape[0], np.float64)
    for i in range(X.shape[0]):
        n = 0
        while tr["feature"][n] != -1:
            n = tr["left"][n] if X[i, tr["feature"][n] - 1] <= tr["threshold"][n] else tr["right"][n]
        out[i] = tr["output"][n]
    return out


def test_validation_early_stop_rollback():
    X, lab, qoff = make(4000, 12, "ns", 5)
    Xv, labv, qoffv = make(2500, 12, "ns", 6)
    o, g = pair(X, lab, qoff, n_trees=40, n_leaves=5, early_stop=3)
    o.set_validation(Xv, labv, qoffv); g.set_validation(Xv, labv, qoffv)
    o.init(); g.init()
    vs = np.zeros(len(labv))
    best, best_round, plateau, rounds = 0.0, None, 0, 0
    for r in range(40):
        to, tmo, vmo, so = o.round()
        tg, tmg, vmg, sg = g.boost_round()
        plateau += assert_same_tree(to, tg, X, "round %d" % r)
        assert tmo == tmg, r
        rounds += 1
        # validation docs are unseen data: a plateau-tie threshold may route them differently from the oracle's,
        # so the validation path is checked against a replay of the GPU's OWN trees (LambdaMART.java:230-243)
        vs += np.float64(np.float32(0.1)) * _eval_tree_rows(tg.trimmed(), Xv)
        assert np.array_equal(g.array("VALID_SCORE").view(np.int64), vs.view(np.int64)), r
        acc = np.float32(0)
        for q in range(len(qoffv) - 1):
            a, b = qoffv[q], qoffv[q + 1]
            acc = np.float32(float(acc) + O.query_ndcg(vs[a:b], labv[a:b], 10))
        assert vmg == np.float32(acc / np.float32(len(qoffv) - 1)), r
        if float(vmg) > best:
            best, best_round = float(vmg), r
        assert sg == (r - best_round > 3)
        if plateau == 0:
            assert vmo == vmg and so == sg, r
        if sg:
            break
    ts_g, vs_g = g.finish()
    assert g.num_trees() == best_round + 1 and g.best_validation()[0] == best_round
    assert rounds > best_round + 3 or rounds == 40
    if plateau == 0:
        ts_o, vs_o = o.finish()
        assert ts_o == ts_g and vs_o == vs_g and g.num_trees() == o.trees_kept()


def test_shared_qid_cache_quirk():
    X, lab, qoff = make(1500, 6, "ns", 7)
    Q = len(qoff) - 1
    qkey = np.arange(Q) // 2          # pairs of consecutive lists share a qid
    o, g = pair(X, lab, qoff, n_trees=4, n_leaves=6, qkey=qkey)
    o.init(); g.init()
    for r in range(4):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert_same_tree(to, tg, X, "round %d" % r)
        assert tmo == tmg


def test_long_lists_use_block_kernel_and_tc_variants():
    rng = np.random.default_rng(9)
    sizes = [900, 5, 400, 30, 1300, 7]
    qoff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    n = int(qoff[-1])
    X = rng.integers(0, 40, (n, 6)).astype(np.float32) / 4
    X[:, 1] = rng.random(n).astype(np.float32)
    lab = np.clip((X[:, 0] + X[:, 1] * 4 + rng.random(n) * 3).astype(np.int32) // 4, 0, 4).astype(np.float32)
    for tc in (256, 16, -1):
        Xc = X.copy()
        if tc == -1:
            Xc[:, 1] = np.round(Xc[:, 1] * 1000) / 1000      # keep distinct values under the bin limit
        o, g = pair(Xc, lab, qoff, n_trees=3, n_leaves=7, n_threshold=tc)
        o.init(); g.init()
        for r in range(3):
            to, tmo, _, _ = o.round()
            tg, tmg, _, _ = g.boost_round()
            assert_same_tree(to, tg, Xc, "tc %d round %d" % (tc, r))
            assert tmo == tmg


def test_model_text_roundtrip_and_scoring():
    X, lab, qoff = make(3000, 10, "ns", 11)
    o, g = pair(X, lab, qoff, n_trees=5, n_leaves=6)
    o.init(); g.init()
    for _ in range(5):
        o.round(); g.boost_round()
    g.finish()
    text = g.model_text()
    assert text.startswith("## LambdaMART\n## No. of trees = 5\n## No. of leaves = 6\n"
                           "## No. of threshold candidates = 256\n## Learning rate = 0.1\n## Stop early = 100\n\n"
                           "<ensemble>\n\t<tree id=\"1\" weight=\"0.1\">\n\t\t<split>\n")
    m = N.Model(text)
    assert m.num_trees() == 5
    rows = np.zeros((500, 11), np.float32)
    rows[:, 1:] = X[:500]
    assert np.array_equal(m.predict_rows(rows).view(np.uint32), o.predict(X[:500]).view(np.uint32))


def test_run_twice_is_deterministic_and_async_equals_sync():
    X, lab, qoff = make(9000, 30, "ns", 12)
    outs = []
    for mode in ("sync", "async"):
        g = N.Trainer(n_trees=6, n_leaves=10)
        g.set_train(X, lab, qoff)
        g.init()
        if mode == "sync":
            for _ in range(6):
                g.boost_round(want_tree=False)
        else:
            g.boost_rounds_async(6)
            g.sync()
        outs.append((g.array("SCORE").copy(), [g.get_tree(i).trimmed() for i in range(6)],
                     [g.round_metrics(i)[0] for i in range(6)]))
    assert np.array_equal(outs[0][0].view(np.int64), outs[1][0].view(np.int64))
    assert outs[0][2] == outs[1][2]
    for a, b in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(a["feature"], b["feature"]) and np.array_equal(a["output"], b["output"])


def test_exact_parallel_float_chain_equals_serial_chain_and_oracle():
    """rl_chain.inc: speculative-window evaluation of the Java float running sums must equal the literal serial
    kernel (RL_FLAG_SERIAL_CHAIN) and the oracle bit for bit, including the metric chain over > 4096 queries."""
    X, lab, qoff = make(60000, 12, "ns", 21)          # ~6000 queries -> parallel metric chain
    res = []
    for flags in (0, N.RL_FLAG_SERIAL_CHAIN):
        g = N.Trainer(n_trees=6, n_leaves=12, flags=flags)
        g.set_train(X, lab, qoff)
        g.init()
        trees, mets = [], []
        for _ in range(6):
            t, tm, _, _ = g.boost_round()
            trees.append(t.trimmed()); mets.append(tm)
        res.append((trees, mets, g.array("SCORE"), g.array("CHAIN_STATS")))
    (ta, ma, sa, st_a), (tb, mb, sb, st_b) = res
    assert st_a[0] > 0 and st_a[3] == 6 and st_b[0] == 0
    assert ma == mb and np.array_equal(sa.view(np.int64), sb.view(np.int64))
    for a, b in zip(ta, tb):
        assert np.array_equal(a["output"].view(np.uint32), b["output"].view(np.uint32))
        assert np.array_equal(a["feature"], b["feature"])
    o = O.Oracle(X, lab, qoff, n_trees=6, n_leaves=12)
    o.init()
    for r in range(6):
        _, tmo, _, _ = o.round()
        assert tmo == ma[r]
    assert np.array_equal(o.scores().view(np.int64), sa.view(np.int64))
    print("chain stats (leaf: evaluated, misses repaired, serial finishes; metric: same):", st_a)


def _chain_cases():
    rng = np.random.default_rng(77)
    n = 300_000
    lam = rng.standard_normal(n) * np.exp(rng.standard_normal(n) * 3)
    yield "lambda-like", lam, None
    yield "positive weights", np.abs(lam) * 1e-3, None
    tiny = lam.copy(); tiny[::7] = 1e-45; tiny[::11] = -4.9e-324; tiny[::13] = -0.0
    yield "denormal and zero elements", tiny, None
    yield "all tiny: float subnormal sums", np.full(5000, 1.4e-45) * rng.integers(-3, 4, 5000), None
    yield "all zero / negative zero", np.where(rng.random(4000) < 0.5, 0.0, -0.0), None
    big = lam.copy(); big[1000] = 3e38; big[50_000] = -3e38; big[200_000] = 1e39
    yield "overflow to infinity", big, None
    nanv = lam[:20_000].copy(); nanv[7777] = np.nan
    yield "NaN", nanv, None
    yield "cancellation", np.tile(np.array([1.0, -1.0 + 2.0 ** -52, 1e-30, -1e-30, 16777216.0, 1.0, -16777216.0]), 9000), None
    seg = np.sort(np.concatenate([[0, 0, 1, 2, 513, 1024, n, n], rng.integers(0, n, 40)]))
    yield "ragged segments (empty, 1 element, tile edges)", lam, seg
    yield "single element", np.array([0.1]), None
    # chains longer than a repair window (8192 chunks of <= 1024 elements, rl_chain.inc): the first stitch takes one window, the passes the rest,
    # every one with drifts re-measured from the exact state at its first chunk -- what a leaf of ten million documents needs
    rng2 = np.random.default_rng(78)
    nl = 12_000_000
    long_lam = rng2.standard_normal(nl) * np.exp(rng2.standard_normal(nl) * 2)
    yield "long chain: several repair windows", long_lam, None
    yield "two long segments and a short one", long_lam, np.array([0, 5_000_001, 5_000_100, nl])


@pytest.mark.parametrize("case", list(_chain_cases()), ids=lambda c: c[0].split(" (")[0].replace(" ", "_"))
def test_float_chain_kernels_on_adversarial_data_equal_the_serial_java_sum(case):
    """rl_debug_float_chain (rl_chain.inc: window tables, repair passes, serial finish) == the oracle's literal
    `float s += x` loop (LambdaMART.java:401-408), bit for bit"""
    name, x, seg = case
    seg = np.array([0, len(x)]) if seg is None else seg
    got, stats = N.debug_float_chain(x, seg)
    want = np.array([O.float_chain(x[seg[i]:seg[i + 1]]) for i in range(len(seg) - 1)], np.float32)
    both_nan = np.isnan(got) & np.isnan(want)           # NaN payloads are not part of the contract (Java prints NaN)
    assert np.array_equal(got.view(np.uint32)[~both_nan], want.view(np.uint32)[~both_nan]), (name, got[:8], want[:8], stats)
    print(name, "stats", stats.tolist())


@pytest.mark.parametrize("k", [1, 3, 16, 20, 1000])
def test_ndcg_cutoffs_fused_and_general_lambda_paths(k):
    """NDCG@k for k <= 16 runs the LDS-fused lambda kernel, larger k the global-matrix kernels; k >= list length too"""
    X, lab, qoff = make(5000, 8, "mslr", 31)
    o, g = pair(X, lab, qoff, n_trees=3, n_leaves=8, k=k)
    o.init(); g.init()
    for r in range(3):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert np.array_equal(g.array("LAMBDA").view(np.int64), o.lambdas().view(np.int64)), (k, r)
        assert np.array_equal(g.array("WEIGHT").view(np.int64), o.weights().view(np.int64)), (k, r)
        assert_same_tree(to, tg, X, "k %d round %d" % (k, r))
        assert tmo == tmg


# ---- MART and the other train metrics ----------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("ranker,metric,k,kind,seed", [
    ("MART", "NDCG", 10, "ns", 0), ("MART", "ERR", 10, "mslr", 1),
    ("LAMBDAMART", "DCG", 10, "ns", 2), ("LAMBDAMART", "DCG", 3, "mslr", 3),
    ("LAMBDAMART", "MAP", 0, "ns", 4), ("LAMBDAMART", "MAP", 4, "mslr", 5),
    ("LAMBDAMART", "ERR", 10, "ns", 6), ("LAMBDAMART", "ERR", 3, "mslr", 7), ("LAMBDAMART", "NDCG", 20, "mslr", 8)])
def test_other_rankers_and_metrics_match_the_oracle(ranker, metric, k, kind, seed):
    """lambdas / weights / scores / per-round metric bit for bit, trees equivalent (tree_equiv), final metric equal"""
    n_docs = 3000 if kind == "ns" else 6000
    X, lab, qoff = synth.make_dataset(n_docs, 12, kind, seed_offset=50 + seed)
    rounds, leaves = 5, 8
    o = O.Oracle(X, lab, qoff, n_trees=rounds, n_leaves=leaves, k=k, ranker=ranker, metric=metric, n_threads=4)
    g = N.Trainer(n_trees=rounds, n_leaves=leaves, metric_k=k, metric=metric, ranker=ranker)
    g.set_train(X, lab, qoff)
    o.init(); g.init()
    for m in range(rounds):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert np.array_equal(g.array("LAMBDA"), o.lambdas()), "lambda mismatch in round %d" % m
        if ranker != "MART":
            assert np.array_equal(g.array("WEIGHT"), o.weights())
        assert_equivalent(to, tg, X, ctx="round %d" % m)
        assert np.array_equal(g.array("SCORE"), o.scores()), "scores mismatch in round %d" % m
        assert np.float32(tmg) == np.float32(tmo)
    so, _ = o.finish()
    sg, _ = g.finish()
    assert sg == so
    text = g.model_text()
    assert text.startswith("## %s\n" % ("MART" if ranker == "MART" else "LambdaMART"))


def test_lambda_kernel_variants_by_list_length_give_identical_lambdas(monkeypatch):
    """k_lambda_tiny (<= 16 documents, 16-lane groups) and the 64 / 128 / 192 / 256-wide block variants against the oracle, and
    against each other on the same mixed-length data (RLHIP_TINY_MIN moves the tiny lists between the two kernels)."""
    rng = np.random.default_rng(4)
    sizes = np.concatenate([rng.integers(1, 17, 300), rng.integers(17, 65, 60), rng.integers(65, 129, 30), rng.integers(129, 193, 15),
                            rng.integers(193, 300, 8), [16, 17, 64, 65, 128, 129, 192, 193, 256, 257]])
    rng.shuffle(sizes)
    qoff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    n = int(qoff[-1])
    X = rng.random((n, 6)).astype(np.float32)
    lab = np.floor(3 * X[:, 0] * X[:, 1] + 2 * rng.random(n)).astype(np.float32)
    res = []
    for tiny_min in ("1", "1000000"):
        monkeypatch.setenv("RLHIP_TINY_MIN", tiny_min)
        g = N.Trainer(n_trees=3, n_leaves=6)
        g.set_train(X, lab, qoff)
        g.init()
        rec = []
        for _ in range(3):
            g.boost_round()
            rec.append((g.array("LAMBDA").copy(), g.array("WEIGHT").copy()))
        res.append(rec)
    o = O.Oracle(X, lab, qoff, n_trees=3, n_leaves=6)
    o.init()
    for r in range(3):
        o.round()
        for rec in res:
            assert np.array_equal(rec[r][0], o.lambdas()) and np.array_equal(rec[r][1], o.weights()), "round %d" % r


@pytest.mark.gpu
@pytest.mark.parametrize("metric,k", [("ERR", 10), ("ERR", 3), ("ERR", 16), ("ERR", 20), ("MAP", 0), ("MAP", 4), ("MAP", 15), ("MAP", 30), ("DCG", 5)])
def test_fused_lambda_kernel_of_every_metric_on_mixed_list_lengths(metric, k, monkeypatch):
    """the LDS-resident lambda kernel computes ERR's swap changes from a [size][size + 2] table per query and MAP's from one sequential
    chain per row (k_lambda_fused MODE 1 / 2); cutoffs past kLambdaFusedMaxK (16; MAP visits k + 1 rows) and RLHIP_LAMBDA_UNFUSED take the
    pair-term matrix.  Lists of 1 .. 700 documents (one to three tiles of the widest variant), lambdas / weights bit for bit against the
    oracle over three rounds, and the two paths against each other."""
    rng = np.random.default_rng(40 + k)
    sizes = np.concatenate([rng.integers(1, 17, 200), rng.integers(17, 65, 60), rng.integers(65, 129, 30), rng.integers(129, 193, 12),
                            rng.integers(193, 300, 8), [k, k + 1, k + 2, 16, 17, 64, 65, 256, 257, 512, 513, 700]])
    sizes = sizes[sizes > 0]
    rng.shuffle(sizes)
    qoff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    n = int(qoff[-1])
    X = rng.random((n, 6)).astype(np.float32)
    lab = np.floor(3 * X[:, 0] * X[:, 1] + 2 * rng.random(n)).astype(np.float32)
    lab[qoff[5]:qoff[6]] = 0          # a list without a relevant document (MAP: rdCount == 0)
    lab[qoff[7]:qoff[8]] = 2          # a list of equal labels
    res = []
    for unfused in (False, True):
        if unfused:
            monkeypatch.setenv("RLHIP_LAMBDA_UNFUSED", "1")
        g = N.Trainer(n_trees=3, n_leaves=6, metric=metric, metric_k=k)
        g.set_train(X, lab, qoff)
        g.init()
        rec = []
        for _ in range(3):
            _, tm, _, _ = g.boost_round()
            rec.append((g.array("LAMBDA").copy(), g.array("WEIGHT").copy(), tm))
        res.append(rec)
    o = O.Oracle(X, lab, qoff, n_trees=3, n_leaves=6, metric=metric, k=k)
    o.init()
    for r in range(3):
        _, tmo, _, _ = o.round()
        for which, rec in enumerate(res):
            assert np.array_equal(rec[r][0], o.lambdas()) and np.array_equal(rec[r][1], o.weights()), "round %d path %d" % (r, which)
            assert np.float32(rec[r][2]) == np.float32(tmo)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["NDCG", "ERR", "DCG"])
def test_labels_above_30_wrap_as_java_ints_do(metric):
    """gain = (1 << label) - 1 is Java int arithmetic (metric/DCGScorer.java:28-31,137-139; ERRScorer.java:71-73): the shift count is taken
    mod 32 and the subtraction wraps -- label 31 gives 2^31 - 1, 32 gives 0, 33 gives 1.  RankLib trains on such labels; so does this."""
    rng = np.random.default_rng(5)
    n = 4000
    X = rng.random((n, 8)).astype(np.float32)
    wild = np.array([0, 1, 2, 30, 31, 32, 33, 40, 63, 64, 1000], np.float32)
    # ERR: R = gain / 16 above 1 squares |p| at every position until it overflows; the labels of the training run keep R below 1
    lab = rng.choice(np.array([0, 1, 2, 3, 4, 32, 33, 34, 35, 36, 64, 68], np.float32) if metric == "ERR" else wild, n)
    sizes = rng.integers(1, 40, 400)
    qoff = np.concatenate([[0], np.cumsum(sizes)])
    qoff = qoff[qoff < n].astype(np.int32)
    qoff = np.concatenate([qoff, [n]]).astype(np.int32)
    o = O.Oracle(X, lab, qoff, n_trees=4, n_leaves=8, metric=metric, k=10)
    g = N.Trainer(n_trees=4, n_leaves=8, metric=metric, metric_k=10)
    g.set_train(X, lab, qoff)
    o.init(); g.init()
    for r in range(4):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert np.array_equal(g.array("LAMBDA").view(np.int64), o.lambdas().view(np.int64)), r
        assert np.array_equal(g.array("WEIGHT").view(np.int64), o.weights().view(np.int64)), r
        assert_equivalent(to, tg, X, ctx="round %d" % r)
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r
        assert np.float32(tmg).view(np.uint32) == np.float32(tmo).view(np.uint32)
    assert o.finish()[0] == g.finish()[0]
    if metric == "ERR":
        # the overflow regime: np[] reaches +-Infinity inside the top ten and the swap changes turn Infinity / NaN exactly where the Java's do
        # (its loop past the cutoff adds p * 0, NaN once p is infinite): the first round's lambdas, bit for bit including the NaNs
        lab2 = rng.choice(wild, n)
        o2 = O.Oracle(X, lab2, qoff, n_trees=1, n_leaves=4, metric="ERR", k=10)
        g2 = N.Trainer(n_trees=1, n_leaves=4, metric="ERR", metric_k=10)
        g2.set_train(X, lab2, qoff)
        o2.init(); g2.init()
        o2.round(); g2.boost_round()
        assert not np.isfinite(o2.lambdas()).all()
        assert np.array_equal(g2.array("LAMBDA").view(np.int64), o2.lambdas().view(np.int64))
        assert np.array_equal(g2.array("WEIGHT").view(np.int64), o2.weights().view(np.int64))


# ---- feature sampling of Random Forests (FeatureHistogram.samplingRate), seeded ------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("ranker,frate,n_feat,leaves,seed", [
    ("MART", 0.3, 12, 8, 1), ("MART", 0.3, 136, 20, 2), ("LAMBDAMART", 0.5, 12, 8, 3), ("LAMBDAMART", 0.3, 40, 31, 4),
    ("MART", 0.05, 12, 8, 5), ("MART", 0.999, 12, 8, 6)])
def test_feature_sampling_matches_the_oracle(ranker, frate, n_feat, leaves, seed):
    """every split attempt looks at (int)(rate * F) features drawn by the seeded hash of (seed, tree, node path); the first drawn
    wins a tie (FeatureHistogram.java:272-309).  rate 0.05 * 12 features = 0 features: single-leaf trees."""
    X, lab, qoff = synth.make_dataset(4000, n_feat, "mslr", seed_offset=70 + seed)
    rounds = 4
    o = O.Oracle(X, lab, qoff, n_trees=rounds, n_leaves=leaves, ranker=ranker, n_threads=3, frate=frate, seed=1000 + seed)
    g = N.Trainer(n_trees=rounds, n_leaves=leaves, ranker=ranker, feature_sampling_rate=frate, seed=1000 + seed)
    g.set_train(X, lab, qoff)
    o.init(); g.init()
    used = set()
    for m in range(rounds):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert_equivalent(to, tg, X, ctx="round %d" % m)
        assert np.array_equal(g.array("SCORE"), o.scores()), "scores mismatch in round %d" % m
        assert np.float32(tmg) == np.float32(tmo)
        used |= set(tg.trimmed()["feature"].tolist())
    if frate == 0.05:
        assert used == {-1}
    # the draw really differs from "all features": a different seed grows different trees
    if 0.1 < frate < 0.9:
        g2 = N.Trainer(n_trees=1, n_leaves=leaves, ranker=ranker, feature_sampling_rate=frate, seed=5)
        g2.set_train(X, lab, qoff); g2.init()
        t2, _, _, _ = g2.boost_round()
        g3 = N.Trainer(n_trees=1, n_leaves=leaves, ranker=ranker, feature_sampling_rate=frate, seed=6)
        g3.set_train(X, lab, qoff); g3.init()
        t3, _, _, _ = g3.boost_round()
        assert not np.array_equal(t2.trimmed()["feature"], t3.trimmed()["feature"]) or \
            not np.array_equal(t2.trimmed()["threshold"], t3.trimmed()["threshold"])


# ---- Ensemble.eval at scale (LDS-tiled kernel, packed nodes) ------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n_feat,leaves,rounds", [(136, 31, 37), (20, 10, 16), (12, 150, 3)])
def test_tiled_ensemble_eval_equals_oracle(n_feat, leaves, rounds):
    """trees > one LDS tile, partial document tiles, 137-column rows (140 KiB of LDS), -missingZero columns, and a model
    with more than 256 nodes per tree (falls back to the generic kernel): scores bit for bit"""
    import torch
    X, lab, qoff = make(5000, n_feat, "mslr", 21)
    o, g = pair(X, lab, qoff, n_trees=rounds, n_leaves=leaves)
    o.init(); g.init()
    for _ in range(rounds):
        o.round(); g.boost_round(want_tree=False)
    g.finish()
    m = N.Model(g.model_text())
    assert m.num_trees() == rounds
    n = 1337
    want = o.predict(X[:n])
    rows = np.zeros((n, n_feat + 1), np.float32)
    rows[:, 1:] = X[:n]
    assert np.array_equal(m.predict_rows(rows).view(np.uint32), want.view(np.uint32))
    # rows that stop before the last feature ids: missing columns read as 0 (-missingZero).  Zeroed columns are rows the
    # trees were not grown on, where a tie-resolved split may legitimately route differently from the oracle's (DESIGN.md
    # section 1), so the reference here is a direct evaluation of the SAME trees: Split.eval + Ensemble.eval in numpy.
    cut = n_feat // 2
    trees = [g.get_tree(i).trimmed() for i in range(rounds)]
    lr = np.float64(np.float32(0.1))
    ref = np.zeros(n, np.float32)
    for i in range(n):
        acc = np.float32(0)
        for t in trees:
            nd = 0
            while t["feature"][nd] != -1:
                f = int(t["feature"][nd])
                v = rows[i, f] if f <= cut else np.float32(0)
                nd = int(t["left"][nd]) if v <= t["threshold"][nd] else int(t["right"][nd])
            acc = np.float32(np.float64(acc) + np.float64(t["output"][nd]) * lr)
        ref[i] = acc
    assert np.array_equal(m.predict_rows(rows[:, :cut + 1].copy()).view(np.uint32), ref.view(np.uint32))
    # device-resident rows (the serving call)
    dX = torch.from_numpy(rows).cuda()
    dO = torch.zeros(n, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    m.predict_device(dX.data_ptr(), n, rows.shape[1], dO.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(dO.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_device_exp_equals_fdlibm_restatement_everywhere():
    """both device exps (branch-free fast path, literal e_exp) == the oracle's StrictMath.exp restatement, bit for bit, on the
    range boundaries of e_exp, around 0, +-0.5 ln2, +-1.5 ln2, the overflow / underflow / subnormal edges, and random arguments"""
    import ctypes as C
    edges = [0.0, -0.0, 1.0, -1.0, 2.0 ** -28, 2.0 ** -29, -(2.0 ** -28), 0.5 * np.log(2), 1.5 * np.log(2), 0.34657359027997264,
             1.0397207708399179, 699.9, 700.0, 700.1, -699.9, -700.1, 709.782712893384, 709.7827128933841, -745.1332191019411,
             -745.2, -708.3, -708.5, 710.0, np.inf, -np.inf, np.nan, 1e-300, -1e-300, 88.0, -88.0]
    xs = []
    for e in edges:
        if np.isfinite(e) and e != 0:
            xs += [np.nextafter(e, -np.inf), e, np.nextafter(e, np.inf), -e]
        else:
            xs.append(e)
    rng = np.random.RandomState(5)
    xs = np.concatenate([np.array(xs), rng.uniform(-40, 40, 20000), rng.uniform(-760, 760, 5000), rng.uniform(-2, 2, 20000),
                         rng.uniform(-1e-7, 1e-7, 2000)])
    fast, ref = N.debug_exp(xs)
    L = O.lib()
    want = np.array([L.ro_exp(C.c_double(v)) for v in xs])
    for got in (fast, ref):
        same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), xs[~same][:5]


def test_reciprocal_form_divisions_equal_the_ieee_quotient_bit_for_bit():
    """rl_device.h div_by_rcp / rcp_newton2 / rho_fdlibm: the lambda kernels' divisions without the scaling steps == the compiler's IEEE division
    (and == the oracle's 1 / (1 + StrictMath.exp) on the host), bit for bit, on the operand ranges the kernels use them on
    (learning/tree/LambdaMART.java:383, metric/NDCGScorer.java:154) and on their edges"""
    import ctypes as C
    rng = np.random.RandomState(17)
    # rho: every argument class of e_exp, the |x| >= 700 hand-over, infinities, NaN
    edges = [0.0, -0.0, 2.0 ** -28, 2.0 ** -29, 0.34657359027997264, 1.0397207708399179, 699.9, 700.0, 700.1, 709.782712893384, 709.7827128933841,
             -745.1332191019411, -745.2, -708.3, 710.0, 36.0, 36.8, 37.5, 1e-300, 88.0, 0.6931471805599453, 1.3862943611198906]
    xs = []
    for e in edges:
        xs += [e, -e] if e == 0 else [np.nextafter(e, -np.inf), e, np.nextafter(e, np.inf), -np.nextafter(e, -np.inf), -e, -np.nextafter(e, np.inf)]
    xs += [np.inf, -np.inf, np.nan]
    xs = np.concatenate([np.array(xs), rng.uniform(-40, 40, 400000), rng.uniform(-760, 760, 100000), rng.uniform(-2, 2, 400000),
                         rng.uniform(-1e-7, 1e-7, 20000), rng.standard_normal(400000) * 3, np.arange(-699, 700) * np.log(2.0)])
    fast, ref = N.debug_rho(xs)
    same = (fast.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(fast) & np.isnan(ref))
    assert same.all(), xs[~same][:5]
    L = O.lib()
    sub = np.concatenate([xs[:200], xs[-3000::7]])
    f_sub = np.concatenate([fast[:200], fast[-3000::7]])
    with np.errstate(over="ignore", invalid="ignore"):
        want = np.array([1.0 / (1.0 + L.ro_exp(C.c_double(v))) for v in sub])
    same = (f_sub.view(np.uint64) == want.view(np.uint64)) | (np.isnan(f_sub) & np.isnan(want))
    assert same.all(), sub[~same][:5]
    # quotients: numerators 0 or 2^-60 .. 2^70 of either sign, divisors 2^-200 .. 2^200 (lambda_div_fast), and e_exp's (x c) / +-(2 - c) shape
    n = 1500000
    num = np.ldexp(rng.uniform(1, 2, n), rng.randint(-60, 71, n)) * rng.choice([-1.0, 1.0], n)
    den = np.ldexp(rng.uniform(1, 2, n), rng.randint(-200, 201, n))
    num[:1000] = 0.0
    den[1000:3000] = np.ldexp(1.0, rng.randint(-200, 201, 2000))                           # exact powers of two
    den[3000:5000] = np.nextafter(np.ldexp(1.0, rng.randint(-199, 201, 2000)), 0)          # all-ones significands
    num2 = rng.uniform(-0.13, 0.13, n) * np.ldexp(1.0, -rng.randint(0, 50, n))
    den2 = rng.uniform(1.6, 2.4, n) * rng.choice([-1.0, 1.0], n)
    # the discount / gain shapes themselves: (1/log2(i+2) - 1/log2(j+2)) (2^a - 2^b) over sums of gain * discount
    i, j = rng.randint(0, 10, n), rng.randint(0, 3000, n)
    disc = lambda p: 1.0 / (np.log(p + 2.0) / np.log(2.0))
    num3 = (disc(i) - disc(j)) * (np.ldexp(1.0, rng.randint(0, 6, n)) - np.ldexp(1.0, rng.randint(0, 6, n)))
    den3 = rng.uniform(0.3, 60.0, n)
    for a, b in ((num, den), (num2, den2), (num3, den3)):
        fast, ref = N.debug_rho(a, b)
        nz = a != 0                                      # a zero numerator may give the other zero (rl_device.h): equal as values
        assert (fast[nz].view(np.uint64) == ref[nz].view(np.uint64)).all()
        assert (fast[~nz] == 0).all() and (ref[~nz] == 0).all()
        assert (ref.view(np.uint64) == (a / b).view(np.uint64))[nz].all()


# ---- edge cases: ragged / degenerate inputs, every way a tree can stop growing -----------------------------------------
def _edge_cases():
    rng = np.random.RandomState(11)
    cases = {}
    # one-document queries between ordinary ones, a 2-document query, 70 documents in total (less than one wavefront per kernel tile)
    sizes = [1, 5, 1, 1, 9, 2, 30, 1, 20]
    n = sum(sizes)
    cases["ragged_tiny"] = (rng.rand(n, 3).astype(np.float32), rng.randint(0, 5, n).astype(np.float32),
                            np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32), dict(n_leaves=6, n_trees=4))
    # every label equal: no active pair, all lambdas 0, the root has deviance 0 -> single-leaf trees of output 0
    cases["all_labels_equal"] = (rng.rand(300, 4).astype(np.float32), np.full(300, 2, np.float32),
                                 np.arange(0, 301, 10).astype(np.int32), dict(n_leaves=8, n_trees=3))
    # all labels 0: ideal DCG 0 -> NDCG 0 for every query
    cases["all_zero_labels"] = (rng.rand(200, 3).astype(np.float32), np.zeros(200, np.float32), np.arange(0, 201, 20).astype(np.int32),
                                dict(n_leaves=5, n_trees=2))
    # constant features: thresholds = {value, MAX_VALUE}, no admissible split anywhere
    cases["constant_features"] = (np.full((150, 3), 0.25, np.float32), rng.randint(0, 3, 150).astype(np.float32),
                                  np.arange(0, 151, 15).astype(np.int32), dict(n_leaves=7, n_trees=2))
    # one feature, two leaves; and min leaf support larger than half the data (root cannot split)
    X1 = rng.rand(400, 1).astype(np.float32)
    l1 = (X1[:, 0] * 3).astype(np.int32).astype(np.float32)
    cases["one_feature_two_leaves"] = (X1, l1, np.arange(0, 401, 8).astype(np.int32), dict(n_leaves=2, n_trees=5))
    cases["one_leaf"] = (X1, l1, np.arange(0, 401, 8).astype(np.int32), dict(n_leaves=1, n_trees=3))
    cases["mls_blocks_root"] = (X1, l1, np.arange(0, 401, 8).astype(np.int32), dict(n_leaves=10, n_trees=2, mls=250))
    cases["mls_stops_children"] = (rng.rand(500, 5).astype(np.float32), rng.randint(0, 5, 500).astype(np.float32),
                                   np.arange(0, 501, 10).astype(np.int32), dict(n_leaves=31, n_trees=4, mls=60))
    # more leaves asked for than the data can give (every leaf ends with one distinct row), large labels, duplicate rows
    Xd = np.repeat(rng.rand(40, 2).astype(np.float32), 3, axis=0)
    cases["more_leaves_than_rows"] = (Xd, rng.randint(0, 31, 120).astype(np.float32), np.arange(0, 121, 12).astype(np.int32),
                                      dict(n_leaves=100, n_trees=3))
    # a single query holding everything
    cases["single_query"] = (rng.rand(700, 6).astype(np.float32), rng.randint(0, 5, 700).astype(np.float32), np.array([0, 700], np.int32),
                             dict(n_leaves=12, n_trees=3))
    return cases


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(_edge_cases()))
def test_edge_cases_match_the_oracle(name):
    X, lab, qoff, kw = _edge_cases()[name]
    o, g = pair(X, lab, qoff, **kw)
    o.init(); g.init()
    for r in range(kw["n_trees"]):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert_same_tree(to, tg, X, "%s round %d" % (name, r))
        assert np.array_equal(g.array("LAMBDA").view(np.int64), o.lambdas().view(np.int64))
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64))
        assert np.float32(tmo).view(np.uint32) == np.float32(tmg).view(np.uint32)
    so, _ = o.finish()
    sg, _ = g.finish()
    assert so == sg
    m = N.Model(g.model_text())
    rows = np.zeros((X.shape[0], X.shape[1] + 1), np.float32); rows[:, 1:] = X
    assert np.array_equal(m.predict_rows(rows).view(np.uint32), g.predict(X).view(np.uint32))


@pytest.mark.gpu
def test_bad_inputs_are_rejected_with_ranklib_style_errors():
    X, lab, qoff = make(200, 3, "ns", 9)
    Xb = X.copy(); Xb[17, 1] = np.nan                      # NaN never reaches the trainer in the reference (DataPoint.getFeatureValue resolves it to 0);
    g = N.Trainer(n_trees=1, n_leaves=4)                    # +-Infinity is a value like any other since round 4 (test_infinite_feature_values_...)
    with pytest.raises(N.RankLibError):
        g.set_train(Xb, lab, qoff); g.init()
    g = N.Trainer(n_trees=1, n_leaves=4)
    with pytest.raises(N.RankLibError):
        g.set_train(X, -lab - 1, qoff)                      # negative labels (learning/DataPoint.java:71-73)
    with pytest.raises(N.RankLibError):
        g.set_train(X, lab, np.array([0, 100, 50, 200], np.int32))      # query offsets must ascend
    with pytest.raises(N.RankLibError):
        g.set_train(X[:0], lab[:0], np.array([0], np.int32))            # no data
    g.set_train(X, lab, qoff)
    with pytest.raises(N.RankLibError):
        g.boost_round()                                     # rl_init has not been called
    g.init(); g.boost_round()
    with pytest.raises(N.RankLibError):
        g.boost_round()                                     # more rounds than n_trees


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sparse", "dense", "rows"])
def test_sparse_700_feature_shape_matches_the_oracle(mode, monkeypatch):
    """Yahoo-set1 shape (BASELINE.json configs[3]) in small: 700 columns, 175 of them all zero (one threshold + MAX_VALUE,
    never split on), the rest 85 % zeros.  The tree learner densifies everything (learning/SparseDataPoint.java:82-95), so only the
    column count changes for the oracle; the GPU's root pass runs its sparse-column path (rl_csc.inc: entry lists instead of rows) or,
    with RLHIP_CSC_DENS=0, the dense rows.  Both must give the oracle's trees / scores / metrics: the paths are bit-identical."""
    if mode == "dense":
        monkeypatch.setenv("RLHIP_CSC_DENS", "0")
    if mode != "sparse":
        monkeypatch.setenv("RLHIP_CROWS", "0")          # child passes from the 32-byte rows instead of the compact rows
    X, lab, qoff = make(6000, 700, "yahoo", 31)
    o, g = pair(X, lab, qoff, n_trees=4, n_leaves=12)
    o.init(); g.init()
    nb = g.array("NBINS")
    assert (nb[np.abs(X).sum(0) == 0] == 2).all()
    info = g.array("SPARSE_INFO")
    assert (info[4] >= 30) if mode == "sparse" else (info[4] == 0), info          # compact rows of the child passes: on by themselves for the groups of 85 %-zero columns
    if mode == "sparse":
        assert 0 < info[5] <= int((X != 0).sum()) and info[6] * 20 <= 6000 * info[4], info    # entries = cells outside the mode bins of those groups; few dense fallbacks
    if mode == "dense":
        assert info[0] == 0 and info[1] == 0 and info[2] == 44
    else:
        nzv = int((X[:, nb > 2] != 0).sum())
        assert info[0] >= 40 and info[0] + info[2] == 44 and info[3] <= int((nb > 2).sum()), info   # (nearly) every group is sparse at 15 % density
        assert 0.5 * nzv <= info[1] <= nzv, (info, nzv)         # one entry per non-zero value of the sparse groups (the zero bin is the mode bin)
    for r in range(4):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert_same_tree(to, tg, X, "round %d" % r)
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64))
        assert np.float32(tmo).view(np.uint32) == np.float32(tmg).view(np.uint32)
        # the root histogram: the same exact sums whichever path accumulated them
        tot = int(g.array("QUANT").astype(object).sum())
        fixed = g.array("ROOT_SUM_FIXED")
        for f in np.nonzero(nb > 2)[0][:60]:
            hi, lo = int(fixed[f, nb[f] - 1, 0]), int(fixed[f, nb[f] - 1, 1]) & 0xFFFFFFFFFFFFFFFF
            assert hi * 2 ** 64 + lo == tot, (r, f)


def test_infinite_feature_values_are_binned_as_the_reference_bins_them():
    """+-Infinity in the rows (learning/tree/LambdaMART.java:108-149 accepts them; NaN never reaches the trainer: DataPoint.getFeatureValue).  A table of
    distinct values holds the infinite value as a value; a 256-step table of a column with maximum +Infinity is [fmin, Inf, .., MAX_VALUE]; one with minimum
    -Infinity is [-Inf, NaN, .., MAX_VALUE] and every document above -Infinity lands in bin 1 (`value > NaN` never breaks the Java's loop).  Thresholds,
    bins, counts, trees, scores and Ensemble.eval of the saved model equal the oracle's."""
    rng = np.random.default_rng(3)
    X, lab, qoff = make(4000, 8, "mslr", 17)
    X = X.copy()
    n = X.shape[0]
    X[:, 0] = np.floor(rng.random(n) * 9)                         # few distinct values: exact-value thresholds ...
    X[rng.random(n) < 0.03, 0] = np.inf; X[rng.random(n) < 0.02, 0] = -np.inf      # ... among them +-Infinity
    X[:, 1] = rng.random(n).astype(np.float32); X[rng.random(n) < 0.05, 1] = np.inf           # > 256 distinct values, maximum +Infinity
    X[:, 2] = rng.random(n).astype(np.float32); X[rng.random(n) < 0.05, 2] = -np.inf          # minimum -Infinity: NaN thresholds
    X[:, 3] = rng.random(n).astype(np.float32); X[rng.random(n) < 0.04, 3] = np.inf; X[rng.random(n) < 0.04, 3] = -np.inf
    o, g = pair(X, lab, qoff, n_trees=4, n_leaves=12)
    o.init(); g.init()
    nb, thr, bins, cnt = g.array("NBINS"), g.array("THRESHOLDS"), g.array("BINS"), g.array("ROOT_COUNT")
    for f in range(8):
        T = o.n_bins(f)
        assert nb[f] == T, f
        assert np.array_equal(thr[f, :T], o.thresholds(f), equal_nan=True), (f, thr[f, :6], o.thresholds(f)[:6])
        assert np.array_equal(bins[f].astype(np.int32), o.bins(f)), f
        assert np.array_equal(cnt[f, :T], o.root_count(f)), f
    assert np.isinf(thr[0, nb[0] - 2]) and np.isnan(thr[2, 1]) and np.isinf(thr[1, 1])
    for r in range(4):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert_same_tree(to, tg, X, "round %d" % r)
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64))
        assert tmo == tmg
    g.finish()
    m = N.Model(g.model_text())
    rows = np.zeros((n, 9), np.float32); rows[:, 1:] = X
    assert np.array_equal(m.predict_rows(rows).view(np.uint32), o.predict(X).view(np.uint32))


@pytest.mark.parametrize("tc", [-1, 6000])
def test_threshold_tables_beyond_4095_entries_split_into_virtual_features(tc):
    """-tc -1 on columns with 9 000 distinct values (every distinct value a threshold, learning/tree/LambdaMART.java:135-140) and -tc 6000 (a 6 001-entry
    step table, :141-149): the reference has no limit on the table size.  A feature whose table exceeds 4 095 entries is histogrammed as several virtual
    features over runs of 4 094 thresholds (rl_init); the trees name the real feature id and the real threshold, and trees, lambdas, scores, metrics and
    Ensemble.eval of the saved model equal the oracle's.  Exact ties keep the first candidate in the Java's scan order on such data (no lazy
    re-decision: the Java's f64 prefix runs over ALL bins of the real feature)."""
    rng = np.random.default_rng(9)
    X, lab, qoff = make(9000, 6, "mslr", 29)
    X = X.copy()
    n = X.shape[0]
    X[:, 1] = rng.permutation(n).astype(np.float32) / 7.0        # 9 000 distinct values
    X[:, 4] = (rng.random(n) * 1e4).astype(np.float32)           # ~9 000 distinct values, another scale
    X[:, 2] = np.floor(rng.random(n) * 12)                       # an ordinary low-cardinality column in between
    o, g = pair(X, lab, qoff, n_trees=3, n_leaves=16, n_threshold=tc)
    o.init(); g.init()
    nf, cols = g.hist_features()
    nb, thr = g.array("NBINS"), g.array("THRESHOLDS")
    assert nf > 6 and sorted(set(cols.tolist())) == list(range(6)) and (np.diff(cols) >= 0).all(), (nf, cols)
    for f in range(6):                                           # the runs of a real feature, MAX_VALUE catch-alls dropped, are its table
        runs = [v for v in range(nf) if cols[v] == f]
        parts = [thr[v, :nb[v] - (1 if v != runs[-1] else 0)] for v in runs]
        table = np.concatenate(parts)
        assert len(table) == o.n_bins(f), (f, len(table), o.n_bins(f))
        assert np.array_equal(table.view(np.uint32), o.thresholds(f).view(np.uint32)), f
        assert all(nb[v] <= 4095 for v in runs)
    for r in range(3):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert np.array_equal(g.array("LAMBDA").view(np.int64), o.lambdas().view(np.int64)), r
        assert_equivalent(to, tg, X, "round %d" % r)              # (ties may store another member of the tie: first candidate wins here)
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64))
        assert tmo == tmg
        tt = tg.trimmed()
        assert set(tt["feature"][tt["feature"] != -1].tolist()) <= set(range(1, 7))         # real feature ids
    g.finish()
    m = N.Model(g.model_text())
    rows = np.zeros((n, 7), np.float32); rows[:, 1:] = X
    assert np.array_equal(m.predict_rows(rows).view(np.uint32), g.predict(X).view(np.uint32))


@pytest.mark.parametrize("kind,nfeat", [("mslr", 40), ("mixed", 48)])
def test_compact_rows_forced_on_dense_and_mixed_data(kind, nfeat, monkeypatch):
    """RLHIP_CROWS=1 on data the heuristic would never pick: dense MSLR-shaped columns (every row has more than eight entries outside the mode bins
    and is read through the dense fallback of the compact-row kernel) and a mix of sparse and dense groups (both kinds of row in one launch).
    The partial histograms are the same atomics either way: trees, scores and metrics equal the oracle's."""
    monkeypatch.setenv("RLHIP_CROWS", "1")
    rng = np.random.default_rng(11)
    X, lab, qoff = make(7000, nfeat, "mslr", 43)
    if kind == "mixed":
        X = X.copy()
        for f in range(16, 40):
            X[rng.random(7000) < (0.93 if f < 32 else 0.6), f] = 0.0      # group 1: <= a couple of entries a row; group 2: around the limit of eight
    o, g = pair(X, lab, qoff, n_trees=3, n_leaves=14)
    o.init(); g.init()
    info = g.array("SPARSE_INFO")
    assert info[4] == (nfeat + 15) // 16 and info[6] > 0, info          # on for every group, and some rows take the dense fallback
    for r in range(3):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert_same_tree(to, tg, X, "round %d" % r)
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64))
        assert tmo == tmg


def test_mixed_sparse_and_dense_columns():
    """a group of sparse columns (entry lists), a dense group (rows) and a mixed one (stays dense), dead columns among them"""
    rng = np.random.default_rng(5)
    X, lab, qoff = make(9000, 40, "mslr", 41)
    X = X.copy()
    for f in list(range(0, 16)) + list(range(32, 40, 2)):
        X[rng.random(9000) < 0.9, f] = 0.0           # 90 % zeros: group 0 is sparse, group 2 half and half
    X[:, 7] = 0.0                                    # dead columns
    X[:, 20] = 0.0
    o, g = pair(X, lab, qoff, n_trees=4, n_leaves=16)
    o.init(); g.init()
    info = g.array("SPARSE_INFO")
    assert info[0] >= 1 and info[2] >= 1 and info[0] + info[2] == 3, info             # 40 columns = 3 groups
    for r in range(4):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert_same_tree(to, tg, X, "round %d" % r)
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64))
        assert tmo == tmg


@pytest.mark.gpu
def test_fuzz_of_small_configurations_has_no_unexplained_mismatch():
    """tools/fuzz_parity.py: 300 random configurations (rankers x metrics x cut-offs x leaves x mls x -tc x feature sampling x list
    lengths); wherever the GPU's and the oracle's trees part, the tool must be able to verify a tie situation of DESIGN.md 1 in
    rational arithmetic -- anything else is a mismatch and fails the run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "300", "2024"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "300 cases: 0 mismatches" in r.stdout, r.stdout[-500:]


def test_chunked_row_upload_equals_one_shot():
    """rl_set_train(X = NULL) + rl_set_rows in blocks (the JNI shim's path for matrices beyond a 2 GiB direct buffer) == one upload"""
    X, lab, qoff = make(5000, 9, "ns", 21)
    a = N.Trainer(n_trees=3, n_leaves=7); a.set_train(X, lab, qoff); a.init()
    b = N.Trainer(n_trees=3, n_leaves=7); b.set_train(X, lab, qoff, chunk_rows=1237); b.init()
    for _ in range(3):
        ta, ma, _, _ = a.boost_round()
        tb, mb, _, _ = b.boost_round()
        assert ma == mb
        for k in ("feature", "threshold", "left", "right", "output", "count"):
            assert np.array_equal(ta.trimmed()[k], tb.trimmed()[k])
    lib = N.lib()
    d = N.Trainer(n_trees=1, n_leaves=3)
    Xc, labc, qoffc = np.ascontiguousarray(X, np.float32), np.ascontiguousarray(lab, np.float32), np.ascontiguousarray(qoff, np.int32)
    N.check(lib.rl_set_train(d.h, None, Xc.shape[0], Xc.shape[1], labc.ctypes.data, qoffc.ctypes.data, len(qoffc) - 1, None, None))
    with pytest.raises(N.RankLibError):
        N.check(lib.rl_init(d.h))                      # rows missing
    with pytest.raises(N.RankLibError):
        N.check(lib.rl_set_rows(d.h, 0, 10, 5, Xc.ctypes.data))      # not consecutive


def test_err_with_another_gmax_matches_the_oracle():
    """-gmax 3: ERRScorer.MAX = 8 (eval/Evaluator.java:241-242) changes R, the lambdas and the per-round ERR"""
    X, lab, qoff = make(3000, 8, "mslr", 31)
    lab = np.minimum(lab, 3).astype(np.float32)
    try:
        O.set_err_max(8.0); N.set_err_max(8.0)
        o = O.Oracle(X, lab, qoff, n_trees=3, n_leaves=8, metric="ERR", k=10)
        g = N.Trainer(n_trees=3, n_leaves=8, metric="ERR", metric_k=10)
        g.set_train(X, lab, qoff)
        o.init(); g.init()
        for r in range(3):
            to, tmo, _, _ = o.round()
            tg, tmg, _, _ = g.boost_round()
            assert np.array_equal(g.array("LAMBDA").view(np.int64), o.lambdas().view(np.int64)), r
            assert_same_tree(to, tg, X, "round %d" % r)
            assert tmo == tmg
    finally:
        O.set_err_max(16.0); N.set_err_max(16.0)
    o2 = O.Oracle(X, lab, qoff, n_trees=1, n_leaves=8, metric="ERR", k=10)
    o2.init(); o2.round()
    assert not np.array_equal(o2.lambdas(), o.lambdas())          # MAX = 16 gives other lambdas


@pytest.mark.parametrize("metric,n_long", [("NDCG", 12000), ("MAP", 5600)])
def test_ranked_lists_beyond_5000_documents(metric, n_long):
    """RankLib puts no limit on a ranked list's length (LambdaMART.java:361-396): lists past the LDS capacity of the block-per-query
    ranking kernel go through k_rank_huge (tiled rank by counting); the lambda kernels tile over the columns anyway."""
    rng = np.random.default_rng(12)
    sizes = [n_long, 40, 700, 9, 5100]
    qoff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    n = int(qoff[-1])
    X = rng.random((n, 5)).astype(np.float32)
    X[:, 0] = np.floor(X[:, 0] * 12)
    lab = np.clip(np.floor(X[:, 0] / 3 + X[:, 1] * 2 + rng.random(n)), 0, 4).astype(np.float32)
    k = 0 if metric == "MAP" else 10
    o = O.Oracle(X, lab, qoff, n_trees=3, n_leaves=8, metric=metric, k=k)
    g = N.Trainer(n_trees=3, n_leaves=8, metric=metric, metric_k=k)
    g.set_train(X, lab, qoff)
    o.init(); g.init()
    for r in range(3):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert np.array_equal(g.array("LAMBDA").view(np.int64), o.lambdas().view(np.int64)), r
        assert np.array_equal(g.array("WEIGHT").view(np.int64), o.weights().view(np.int64)), r
        assert_same_tree(to, tg, X, "round %d" % r)
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r
        assert tmo == tmg, r
    so, _ = o.finish()
    sg, _ = g.finish()
    assert so == sg


@pytest.mark.parametrize("mls,n_docs,strict", [(1, 300, True), (7, 2000, True), (40, 6000, False)])
def test_unlimited_leaves(mls, n_docs, strict):
    """-leaf -1 (RegressionTree.java:72 `nodes == -1`): the tree grows until no leaf can be split; only -mls bounds it.  Trees this deep
    end in nodes of a few documents, where different partitions tie exactly all the time: those cases run under RL_FLAG_JAVA_ORDER,
    which resolves ties as the Java does, and must give IDENTICAL trees."""
    X, lab, qoff = make(n_docs, 6, "ns", 51)
    o = O.Oracle(X, lab, qoff, n_trees=3, n_leaves=-1, mls=mls)
    g = N.Trainer(n_trees=3, n_leaves=-1, min_leaf_support=mls, flags=N.RL_FLAG_JAVA_ORDER if strict else 0)
    g.set_train(X, lab, qoff)
    o.init(); g.init()
    for r in range(3):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        ties = assert_equivalent(to, tg, X, "round %d" % r)
        assert ties == 0 or not strict
        assert to.n_nodes > 2 * 10 - 1                       # more leaves than the default budget
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r
        assert tmo == tmg, r
    assert "## No. of leaves = -1" in g.model_text()


def test_ensemble_eval_with_1200_trees_streams_many_lds_tiles():
    """configs[4] in small: a model of 1200 trees (40 trained rounds tiled 30 x, as bench.py --workload infer builds its 10 000) is scored
    through ~38 LDS tiles of 32 trees per block; Ensemble.eval's float accumulation in tree order must survive the tile boundaries
    (learning/tree/Ensemble.java:110-116): scores equal the oracle's flat evaluation of the same trees, bit for bit."""
    X, lab, qoff = make(4000, 24, "mslr", 61)
    g = N.Trainer(n_trees=40, n_leaves=31)
    g.set_train(X, lab, qoff); g.init()
    g.boost_rounds_async(40); g.sync(); g.finish()
    trees = [g.get_tree(i).trimmed() for i in range(40)]
    text = g.model_text()
    head, body = text.split("<ensemble>\n", 1)
    blocks = body.rsplit("</ensemble>", 1)[0].split("\t</tree>\n")[:-1]
    tiled = [b.split(">", 1)[1] for _ in range(30) for b in blocks]
    text = head + "<ensemble>\n" + "".join("\t<tree id=\"%d\" weight=\"0.1\">%s\t</tree>\n" % (i + 1, t) for i, t in enumerate(tiled)) + "</ensemble>\n"
    m = N.Model(text)
    assert m.num_trees() == 1200
    n = 3001                                         # partial last document tile
    rows = np.zeros((n, 25), np.float32)
    rows[:, 1:] = X[:n]
    got = m.predict_rows(rows)
    want = O.eval_flat_model([trees[i % 40] for i in range(1200)], rows, n_threads=8)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    m.close()


# ---- -qrel: external relevance judgments (eval/Evaluator.java:580-591; NDCGScorer.java:50-96, APScorer.java:45-66, :86-94, :124-143) -------
@pytest.mark.gpu
@pytest.mark.parametrize("metric,k", [("NDCG", 10), ("MAP", 0)])
def test_external_relevance_judgments_match_the_oracle(metric, k):
    """NDCG: lists whose qid has an idealGains entry from the judgment file never compute their own ideal DCG (some entries here are larger than
    the list's own, one is 0: the list then scores 0 and gets no lambdas); MAP: rdCount comes from the file, 0 for qids that are not in it.
    Training set AND validation set, lambdas / weights / trees / scores / both metrics against the oracle, early stopping included."""
    rng = np.random.default_rng(77)
    X, lab, qoff = synth.make_dataset(5000, 10, "mslr", seed_offset=91)
    Xv, labv, qoffv = synth.make_dataset(2500, 10, "mslr", seed_offset=92)
    Q, Qv = len(qoff) - 1, len(qoffv) - 1
    ideal = idealv = rd = rdv = None
    if metric == "NDCG":
        ideal = np.where(rng.random(Q) < 0.6, 5.0 + 40.0 * rng.random(Q), np.nan); ideal[3] = 0.0
        idealv = np.where(rng.random(Qv) < 0.5, 5.0 + 40.0 * rng.random(Qv), np.nan)
    else:
        rd = rng.integers(0, 60, Q).astype(np.int32); rd[rng.random(Q) < 0.2] = 0
        rdv = rng.integers(0, 60, Qv).astype(np.int32)
    o = O.Oracle(X, lab, qoff, n_trees=12, n_leaves=8, metric=metric, k=k, early_stop=3)
    # strict mode: identical trees, so the validation rows take the same branches and the validation metric can be compared unconditionally
    g = N.Trainer(n_trees=12, n_leaves=8, metric=metric, metric_k=k, early_stop_rounds=3, flags=N.RL_FLAG_JAVA_ORDER)
    g.set_train(X, lab, qoff)
    o.set_validation(Xv, labv, qoffv); g.set_validation(Xv, labv, qoffv)
    o.set_external(False, ideal, rd); o.set_external(True, idealv, rdv)
    g.set_external_judgments(False, ideal, rd); g.set_external_judgments(True, idealv, rdv)
    o.init(); g.init()
    for r in range(12):
        to, tmo, vmo, so = o.round()
        tg, tmg, vmg, sg = g.boost_round()
        assert np.array_equal(g.array("LAMBDA").view(np.int64), o.lambdas().view(np.int64)), r
        assert np.array_equal(g.array("WEIGHT").view(np.int64), o.weights().view(np.int64)), r
        assert_equivalent(to, tg, X, ctx="round %d" % r)
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r
        assert np.float32(tmg).view(np.uint32) == np.float32(tmo).view(np.uint32) and np.float32(vmg).view(np.uint32) == np.float32(vmo).view(np.uint32), r
        assert so == sg
        if sg:
            break
    assert o.finish() == g.finish()
    # and the judgments matter: the same run without them ends elsewhere
    g2 = N.Trainer(n_trees=2, n_leaves=8, metric=metric, metric_k=k)
    g2.set_train(X, lab, qoff); g2.init(); g2.boost_round()
    o2 = O.Oracle(X, lab, qoff, n_trees=2, n_leaves=8, metric=metric, k=k); o2.set_external(False, ideal, rd); o2.init(); o2.round()
    assert not np.array_equal(g2.array("LAMBDA"), o2.lambdas())


@pytest.mark.gpu
@pytest.mark.parametrize("n_docs,n_qlevel,leaves", [(9001, 5, 12), (30011, 11, 31)])
def test_query_level_columns_take_the_quad_folded_histogram_path(n_docs, n_qlevel, leaves, monkeypatch):
    """columns that hold one value per query come in runs of equal bins: rl_init flags them (k_run_stats) and k_hist<.., RUNS> folds the four
    samples of a quad that agree into one LDS atomic (DPP quad permutes).  Integer sums, so nothing may change: histograms, trees, scores
    against the oracle, and against the same run with the path switched off (RLHIP_RUNS_OFF)."""
    rng = np.random.default_rng(n_docs)
    X, lab, qoff = synth.make_dataset(n_docs, 24, "mslr", seed_offset=33)
    X = X.copy()
    nq = len(qoff) - 1
    for j in range(n_qlevel):           # query-level columns, some with few distinct values (one of them becomes the mode bin), spread over both groups
        vals = rng.random(nq).astype(np.float32) if j % 2 else np.floor(rng.random(nq) * 3).astype(np.float32)
        X[:, 2 * j + 1] = np.repeat(vals, np.diff(qoff))
    runs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("RLHIP_RUNS_OFF", "1")
        g = N.Trainer(n_trees=4, n_leaves=leaves)
        g.set_train(X, lab, qoff)
        g.init()
        rec = []
        for _ in range(4):
            t, tm, _, _ = g.boost_round()
            rec.append((t.trimmed(), g.array("SCORE").copy(), tm))
        runs.append(rec)
    o = O.Oracle(X, lab, qoff, n_trees=4, n_leaves=leaves, n_threads=4)
    o.init()
    for r in range(4):
        to, tmo, _, _ = o.round()
        for rec in runs:
            assert np.array_equal(rec[r][1].view(np.int64), o.scores().view(np.int64)), r
            assert np.float32(rec[r][2]).view(np.uint32) == np.float32(tmo).view(np.uint32)
        for key in ("feature", "left", "right", "count"):
            assert np.array_equal(runs[0][r][0][key], runs[1][r][0][key]), (r, key)
        assert np.array_equal(runs[0][r][0]["threshold"].view(np.uint32), runs[1][r][0]["threshold"].view(np.uint32))
        assert np.array_equal(runs[0][r][0]["deviance"].view(np.int64), runs[1][r][0]["deviance"].view(np.int64))      # exact fixed-point sums either way


@pytest.mark.gpu
@pytest.mark.parametrize("F,n,leaves", [(3000, 4000, 15), (136, 8000, 128), (17, 20000, 300)])
def test_wide_and_deep_shapes_match_the_oracle(F, n, leaves):
    """thousands of columns (188 feature groups; the growth bookkeeping's per-feature reductions) and hundreds of leaves (node records beyond
    the LDS copy, the serial queue of select_step): lambdas and scores bit for bit, trees equivalent"""
    X, lab, qoff = synth.make_dataset(n, F, "mslr", seed_offset=5)
    g = N.Trainer(n_trees=3, n_leaves=leaves)
    g.set_train(X, lab, qoff)
    g.init()
    o = O.Oracle(X, lab, qoff, n_trees=3, n_leaves=leaves, n_threads=os.cpu_count() or 8)
    o.init()
    for r in range(3):
        to, tmo, _, _ = o.round()
        tg, tmg, _, _ = g.boost_round()
        assert np.array_equal(g.array("LAMBDA"), o.lambdas())
        assert_equivalent(to, tg, X, "round %d" % r)
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r
        assert np.float32(tmg).view(np.uint32) == np.float32(tmo).view(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("wide", ["1", "0"])
def test_bookkeeping_kernel_of_wide_data_gives_the_oracles_trees(wide, monkeypatch):
    """161 .. 768 histogram features (the Yahoo-set1 shape): k_select2<true> (rl_step2.inc: the gains of a lane's share in registers, its best record and
    the tie pass from memory, DPP reductions; RLHIP_SELECT2_WIDE=1, the default) or round 4's k_select -- against the oracle, stored (feature,
    threshold) pairs included, on data with duplicated and mirrored columns spread over the lanes' shares (exact ties and mirrored cuts of OTHER
    features: the stalled / deferred tie-break re-enters the bookkeeping), with a validation set"""
    monkeypatch.setenv("RLHIP_SELECT2_WIDE", wide)
    X, lab, qoff = make(5000, 400, "mslr", 91)
    X[:, 37] = X[:, 5]; X[:, 170] = X[:, 5]; X[:, 399] = X[:, 5]         # the same column in three other shares
    X[:, 222] = -X[:, 9]; X[:, 310] = X[:, 64] * 2.0                        # a mirrored cut, a rescaled copy
    Xv, lv, qv = make(1500, 400, "mslr", 92)
    Xv[:, 37] = Xv[:, 5]; Xv[:, 170] = Xv[:, 5]; Xv[:, 399] = Xv[:, 5]; Xv[:, 222] = -Xv[:, 9]; Xv[:, 310] = Xv[:, 64] * 2.0
    o, g = pair(X, lab, qoff, n_trees=5, n_leaves=20)
    o.set_validation(Xv, lv, qv); g.set_validation(Xv, lv, qv)
    o.init(); g.init()
    for r in range(5):
        to, tmo, vmo, _ = o.round()
        tg, tmg, vmg, _ = g.boost_round()
        assert_same_tree(to, tg, X, "round %d" % r)
        assert np.float32(tmo).view(np.uint32) == np.float32(tmg).view(np.uint32)
        assert np.float32(vmo).view(np.uint32) == np.float32(vmg).view(np.uint32)
        assert np.array_equal(g.array("SCORE").view(np.int64), o.scores().view(np.int64)), r


@pytest.mark.gpu
@pytest.mark.parametrize("step2,split,extras", [("1", "0", "1"), ("1", "0", "0"), ("0", "0", "0"), ("0", "1", "0")])
def test_fused_and_two_launch_finish_give_the_same_trees(step2, split, extras, monkeypatch):
    """The second half of a growth step: round 5's two short launches k_fin2 + k_select2 (rl_step2.inc; RLHIP_STEP2=1, the default) or round 4's kernels --
    one fused launch whose last block runs the bookkeeping or, on wide data, two (k_hist_finish_wide + k_select, HISTORY.md 4.10; RLHIP_FIN_SPLIT forces
    either).  `extras` switches the other round-5 paths with them: the skipped child histograms of the split that fills the leaf budget and the
    streaming score update.  All against the oracle, with a validation set, on data whose trees need ties resolved (the stalled / deferred paths
    re-enter the bookkeeping from another kernel)"""
    monkeypatch.setenv("RLHIP_STEP2", step2)
    monkeypatch.setenv("RLHIP_FIN_SPLIT", split)
    monkeypatch.setenv("RLHIP_SKIP_LAST", extras)
    monkeypatch.setenv("RLHIP_SCORE_STREAM", extras)
    X, lab, qoff = make(5000, 24, "mslr", 77)
    Xv, lv, qv = make(1500, 24, "mslr", 78)
    o, g = pair(X, lab, qoff, n_trees=6, n_leaves=20)
    o.set_validation(Xv, lv, qv); g.set_validation(Xv, lv, qv)
    o.init(); g.init()
    for r in range(6):
        to, tmo, vmo, _ = o.round()
        tg, tmg, vmg, _ = g.boost_round()
        assert_same_tree(to, tg, X, "round %d" % r)
        assert np.float32(tmo).view(np.uint32) == np.float32(tmg).view(np.uint32)
        assert np.float32(vmo).view(np.uint32) == np.float32(vmg).view(np.uint32)
