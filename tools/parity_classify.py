"""Why two trees part where they first do -- shared by tools/fuzz_parity.py and tools/long_parity.py (test infrastructure)."""
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tree_equiv import node_members  # noqa: E402


def exact_sums(x):
    """(sum, sum of squares) of the float64 array x as exact Fractions: mantissas as Python integers on a common exponent (a Fraction per
    document costs microseconds; a c2 node holds millions)"""
    x = np.asarray(x, np.float64)
    x = x[x != 0.0]
    if len(x) == 0:
        return Fraction(0), Fraction(0)
    m, e = np.frexp(x)
    mi = np.ldexp(m, 53).astype(np.int64)
    e = e.astype(np.int64) - 53
    emin = int(e.min())
    sh = (e - emin).tolist()
    ints = [a << b for a, b in zip(mi.tolist(), sh)]
    s1 = sum(ints)
    s2 = sum(a * a for a in ints)
    two = Fraction(2)
    return Fraction(s1) * two ** emin, Fraction(s2) * two ** (2 * emin)


def classify(to, tg, X, lam, sampling):
    """Why the two trees part where they first do.  Returns a reason if it is one of the documented tie effects (the
    Java resolves exact ties by the rounding noise of its running f64 sums, the GPU's exact sums see a tie and take the first
    candidate), else None:
      * two different partitions of a node with EXACTLY equal S = sl^2/cl + sr^2/cr (rational arithmetic on the f64 lambdas)
      * a node whose exact deviance is 0 (the Java's noise makes it > 0 and splits it) or equals another node's deviance (order
        in the best-first queue, i.e. who gets the last leaves)
      * with feature sampling: the draw of a node depends on its path, and an earlier tie mirrored the sides of a split
      * a node whose lambdas all lie below 2^-49 of the round's largest |lambda|: its deviance is exactly 0 in the GPU's fixed
        point and rounding noise of either sign in the Java's doubles (which then splits a node whose outputs are 0 anyway)"""
    a, b = to.trimmed(), tg.trimmed()
    ma, mb = node_members(a, X, None), node_members(b, X, None)

    def dev(docs):
        if len(docs) == 0:
            return Fraction(0)
        tot, sq = exact_sums(lam[docs])
        return sq - tot * tot / len(docs)
    # walk the whole pairing and collect EVERY place where the trees part: a tie anywhere also moves the leaf budget elsewhere
    found, unexplained = [], []
    all_devs = None
    stack = [(0, 0, False)]
    while stack:
        na, nb, mirrored = stack.pop()
        if not np.array_equal(ma[na], mb[nb]):
            unexplained.append("different members")
            continue
        la, lb = a["feature"][na] == -1, b["feature"][nb] == -1
        if la != lb:
            if sampling and mirrored:
                found.append("feature draw after a mirrored tie")      # a node's draw decides whether it can be split at all
                continue
            d0 = dev(ma[na])
            if d0 == 0:
                found.append("exact deviance 0")
                continue
            res = Fraction(float(np.abs(lam).max())) / (1 << 48)           # the GPU's fixed-point lambdas resolve 2^-49 of max |lambda|
            res2 = Fraction(float(np.abs(lam).max())) ** 2 * len(lam) / (1 << 58)      # ... and lambda^2 resolves 2^(lg N - 61) of its maximum
            if d0 <= len(ma[na]) * max(res * res, res2):
                found.append("deviance below the fixed-point resolution")  # 0 on the GPU, rounding noise (of either sign) in the Java
                continue
            if all_devs is None:
                all_devs = [(0, i, dev(m)) for i, m in ma.items() if len(m)] + [(1, i, dev(m)) for i, m in mb.items() if len(m)]
            if any(abs(o - d0) <= abs(d0) * Fraction(1, 10 ** 12) for w, i, o in all_devs if (w, i) not in ((0, na), (1, nb))):
                found.append("equal deviances in the queue")
            else:
                found.append("leaf budget")            # a consequence if something else explains the divergence
            continue
        if la:
            continue
        al, ar, bl, br = int(a["left"][na]), int(a["right"][na]), int(b["left"][nb]), int(b["right"][nb])
        if np.array_equal(ma[al], mb[bl]):
            stack += [(al, bl, mirrored), (ar, br, mirrored)]
        elif np.array_equal(ma[al], mb[br]):
            stack += [(al, br, True), (ar, bl, True)]
        else:
            docs = ma[na]
            tot = exact_sums(lam[docs])[0]

            def S(left):
                sl = exact_sums(lam[left])[0]
                return sl * sl / len(left) + (tot - sl) * (tot - sl) / (len(docs) - len(left))
            sa, sb = S(ma[al]), S(mb[bl])
            res = Fraction(float(np.abs(lam).max())) / (1 << 48)
            if abs(sa - sb) <= max(abs(sa), abs(sb)) * Fraction(1, 10 ** 13):      # below what the Java's f64 running sums resolve
                found.append("exact tie of S")
            elif abs(sa - sb) <= 4 * res * exact_sums(np.abs(lam[docs]))[0] + len(docs) * res * res:
                found.append("gain difference below the fixed-point resolution")   # node of lambdas tiny against the round's largest
            elif sampling and mirrored:
                found.append("feature draw after a mirrored tie")
            else:
                unexplained.append("different S at node %d/%d: %.17g vs %.17g (%d docs)" % (na, nb, float(sa), float(sb), len(docs)))
    if to.n_nodes != tg.n_nodes and not found:
        return None
    real = [f for f in found if f != "leaf budget"]
    if unexplained or not real:
        classify.last = (found, unexplained)
        return None
    return real[0]
