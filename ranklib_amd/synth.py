"""Deterministic synthetic LETOR-shaped data.

Counter-based: every value is splitmix64(seed, stream, index), so any slice can
be regenerated independently and the generator is trivially shardable by query
(rank r of R generates only its own query range).  No dataset ships with the
repo and there is no network, so MSLR-WEB10K/30K and Yahoo-set1 are replaced by
data of the same SHAPE:

* F dense float features, four families by (column index mod 4):
    0: small integer counts 0..20      (<= 256 distinct -> exact-value thresholds,
                                        learning/tree/LambdaMART.java:135-140)
    1: continuous uniform [0,1)         (> 256 distinct -> 256-step thresholds, :141-149)
    2: heavy-tailed exp(4u)             (skewed bin occupancy)
    3: 70 % exact zeros, else uniform   (zero-dominated bin, like MSLR stream features)
* integer relevance labels 0..4 with MSLR-like marginals, made learnable through a
  latent linear score over 8 of the features,
* docs/query either MSLR-like (~120, log-normal, clipped to [1,1251]) or the
  north-star's "~10 docs/query" (uniform 5..15).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
SEED_DATA, SEED_LABEL, SEED_QSIZE = 20240601, 20240602, 20240603
LABEL_MARGINALS = (0.52, 0.32, 0.13, 0.02, 0.01)


def splitmix64(x):
    """vectorised splitmix64 finaliser on uint64 arrays"""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def uniform01(seed, stream, index):
    """u in [0,1) with 24 bits (exact in float32): (splitmix64(...) >> 40) * 2^-24"""
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95)))
        x = splitmix64(base + np.asarray(index, dtype=np.uint64))
    return ((x >> np.uint64(40)).astype(np.float64) * (2.0 ** -24)).astype(np.float32)


def query_sizes(n_docs, kind="mslr", seed=SEED_QSIZE):
    """query offsets (int32, Q+1) whose sizes sum to exactly n_docs"""
    mean = {"mslr": 120.0, "yahoo": 24.0}.get(kind, 10.0)
    est = int(n_docs / mean * 1.6) + 64
    while True:
        i = np.arange(est, dtype=np.uint64)
        if kind == "mslr":
            u1 = uniform01(seed, 1, i).astype(np.float64)
            u2 = uniform01(seed, 2, i).astype(np.float64)
            z = np.sqrt(-2.0 * np.log(np.maximum(u1, 2.0 ** -24))) * np.cos(2 * np.pi * u2)
            n = np.clip(np.rint(120.0 * np.exp(0.5 * z) / np.exp(0.125)), 1, 1251).astype(np.int64)
        elif kind == "ns":
            n = 5 + (uniform01(seed, 1, i).astype(np.float64) * 11).astype(np.int64)
        elif kind == "yahoo":             # Yahoo-set1 shape: ~24 documents per query
            n = 5 + (uniform01(seed, 1, i).astype(np.float64) * 39).astype(np.int64)
        else:
            raise ValueError("kind must be 'mslr', 'ns' or 'yahoo'")
        c = np.cumsum(n)
        if c[-1] >= n_docs:
            break
        est *= 2
    q = int(np.searchsorted(c, n_docs, side="left")) + 1
    off = np.zeros(q + 1, dtype=np.int64)
    off[1:] = c[:q]
    off[-1] = n_docs
    return off.astype(np.int32)


def features(n_docs, n_features, doc_start=0, seed=SEED_DATA, out=None, sparse=False):
    """row-major float32 [n_docs, n_features] for docs doc_start..doc_start+n_docs-1.
    sparse (Yahoo-set1 shape): 181 of every 700 columns are all zero, the others are 85 % zeros."""
    X = out if out is not None else np.empty((n_docs, n_features), dtype=np.float32)
    docs = np.arange(doc_start, doc_start + n_docs, dtype=np.uint64)
    for f in range(n_features):
        u = uniform01(seed, 1000 + f, docs)
        fam = f % 4
        if sparse:
            if (f * 2654435761 % 700) < 181 and f >= 20:          # the first 20 columns stay informative (labels use some)
                X[:, f] = 0
                continue
            gate = uniform01(seed, 9000 + f, docs)
            dense = u if fam != 2 else np.exp(4.0 * u.astype(np.float64)).astype(np.float32)
            X[:, f] = np.where((gate < np.float32(0.85)) & (f >= 20), np.float32(0), dense)
            continue
        if fam == 0:
            col = np.floor(u.astype(np.float64) ** 2 * 21.0).astype(np.float32)
        elif fam == 1:
            col = u
        elif fam == 2:
            col = np.exp(4.0 * u.astype(np.float64)).astype(np.float32)
        else:
            gate = uniform01(seed, 5000 + f, docs)
            col = np.where(gate < np.float32(0.7), np.float32(0), u)
        X[:, f] = col
    return X


def labels_from(X, doc_start=0, seed=SEED_LABEL, cuts=None):
    """integer labels 0..4 (float32) from a latent score; `cuts` (4 thresholds on the
    latent) may be passed so that shards of one data set label consistently."""
    n, F = X.shape
    use = [f for f in (1, 2, 5, 9, 0, 13, 3, 17) if f < F] or [0]
    z = np.zeros(n, dtype=np.float64)
    for j, f in enumerate(use):
        col = X[:, f].astype(np.float64)
        if f % 4 == 2:
            col = np.log(col) / 4.0
        elif f % 4 == 0:
            col = col / 20.0
        z += (1.0 - 0.08 * j) * col
    docs = np.arange(doc_start, doc_start + n, dtype=np.uint64)
    z += 0.5 * len(use) ** 0.5 * (uniform01(seed, 7, docs).astype(np.float64) - 0.5)
    if cuts is None:
        qs = np.cumsum(LABEL_MARGINALS)[:-1]
        cuts = np.quantile(z, qs)
    lab = np.searchsorted(np.asarray(cuts), z, side="right").astype(np.float32)
    return lab, np.asarray(cuts)


def make_dataset(n_docs, n_features=136, kind="mslr", seed_offset=0):
    """(X float32 [N,F] row-major, labels float32 [N], qoff int32 [Q+1])"""
    qoff = query_sizes(n_docs, kind, SEED_QSIZE + seed_offset)
    X = features(n_docs, n_features, 0, SEED_DATA + seed_offset, sparse=(kind == "yahoo"))
    lab, _ = labels_from(X, 0, SEED_LABEL + seed_offset)
    return X, lab, qoff


SHAPES = {
    # name: (n_docs, n_features, docs/query kind, trees, leaves)   BASELINE.json configs
    "c0": (10_000, 136, "ns", 50, 10),
    "c1": (1_200_000, 136, "mslr", 1000, 31),
    "c1ns": (1_200_000, 136, "ns", 1000, 31),
    "c2": (3_770_000, 136, "mslr", 1000, 31),
    "c2ns": (3_770_000, 136, "ns", 1000, 31),     # the north star's "~10 docs/query" at the WEB30K size (SURVEY.md 8d asks for both list-length variants)
    "c3": (473_000, 700, "yahoo", 1000, 31),      # Yahoo-set1 shape: sparse, 700 features (181 empty)
}


def make_shard(n_docs, n_features=136, kind="mslr", rank=0, world=1, seed_offset=0, cut_sample=262144):
    """rank's contiguous shard of the SAME global data set for any `world` (strong scaling): only the shard's
    feature rows are generated.  Label cuts come from a fixed prefix sample so every rank labels identically."""
    from . import dist as D
    qoff = query_sizes(n_docs, kind, SEED_QSIZE + seed_offset)
    qb, qe = D.partition_queries(qoff, world)[rank]
    d0, d1 = int(qoff[qb]), int(qoff[qe])
    ns = min(cut_sample, n_docs)
    sp = (kind == "yahoo")
    Xs = features(ns, n_features, 0, SEED_DATA + seed_offset, sparse=sp)
    _, cuts = labels_from(Xs, 0, SEED_LABEL + seed_offset)
    if d0 == 0 and d1 <= ns:
        X = Xs[:d1]
    elif d0 == 0:
        X = np.empty((d1, n_features), dtype=np.float32)
        X[:ns] = Xs
        features(d1 - ns, n_features, ns, SEED_DATA + seed_offset, out=X[ns:], sparse=sp)
    else:
        X = features(d1 - d0, n_features, d0, SEED_DATA + seed_offset, sparse=sp)
    lab, _ = labels_from(X, d0, SEED_LABEL + seed_offset, cuts=cuts)
    return X, lab, (qoff[qb:qe + 1] - qoff[qb]).astype(np.int32), int(len(qoff) - 1)


def make_heldout(shape, frac=0.2):
    """A held-out set for a SHAPES entry: documents n_docs .. n_docs (1 + frac) of the shape's own generator (the stream simply continues past the
    training documents), labelled with the cuts of the training prefix (as make_shard labels), lists of the shape's kind from another size stream.
    Used by tools/long_parity.py and bench.py (NDCG@10 on train AND held-out, GPU vs oracle)."""
    n_docs, n_feat, kind, _, _ = SHAPES[shape]
    nv = int(n_docs * frac)
    sp = kind == "yahoo"
    ns = min(262144, n_docs)
    _, cuts = labels_from(features(ns, n_feat, 0, SEED_DATA, sparse=sp), 0, SEED_LABEL)
    Xv = features(nv, n_feat, n_docs, SEED_DATA, sparse=sp)
    labv, _ = labels_from(Xv, n_docs, SEED_LABEL, cuts=cuts)
    qv = query_sizes(nv, kind, SEED_QSIZE + 77)
    return Xv, labv, qv
