"""LETOR input: mirrors features/FeatureManager.java (readInput :187-245, readFeature :267-292,
getFeatureFromSampleVector :303-322) and utilities/FileUtils.smartReader (.gz by extension)."""
import gzip
import logging
import os

import numpy as np

from ._native import RankLibError
from .learning import DataPoint, RankList

logger = logging.getLogger("ranklib_amd")


def smart_reader(path):
    return gzip.open(path, "rt", encoding="utf-8") if path.endswith(".gz") else open(path, "r", encoding="utf-8")


class FeatureManager:
    @staticmethod
    def readInput(inputFile, mustHaveRelDoc=False, useSparseRepresentation=False):
        samples = []
        countEntries = 0
        try:
            fast = FeatureManager._read_native(inputFile, mustHaveRelDoc)
            if fast is not None:
                samples, countEntries = fast
                logger.info("(%d ranked lists, %d entries read)", len(samples), countEntries)
                return samples
            with smart_reader(inputFile) as f:
                lastID, hasRel, rl = "", False, []
                for content in f:
                    content = content.strip()
                    if not content or content[0] == "#":
                        continue
                    qp = DataPoint(content)          # sparse only changes row storage in the reference
                    if lastID and lastID != qp.getID():
                        if not mustHaveRelDoc or hasRel:
                            samples.append(RankList(rl))
                        rl, hasRel = [], False
                    if qp.getLabel() > 0:
                        hasRel = True
                    lastID = qp.getID()
                    rl.append(qp)
                    countEntries += 1
                if rl and (not mustHaveRelDoc or hasRel):
                    samples.append(RankList(rl))
            logger.info("(%d ranked lists, %d entries read)", len(samples), countEntries)
        except RankLibError:
            raise
        except Exception as ex:       # noqa: BLE001
            raise RankLibError("Error in FeatureManager::readInput(): %s" % ex)
        return samples

    native = True                     # parse with librlhip's rl_letor_* (host code); False = the pure-Python reader below

    @staticmethod
    def _read_native(inputFile, mustHaveRelDoc):
        """The same lists as the loop in readInput, with the lines parsed by rl_letor_parse on all host threads.  Lines the native
        parser flags (anything but plain `number qid:token (digits:number)*`) go through DataPoint's own parser, in file order,
        so malformed input raises exactly what it raised before."""
        if not FeatureManager.native:
            return None
        try:
            from . import _native as N
            N.lib()
        except Exception:             # noqa: BLE001 -- no library (e.g. a CPU-only checkout): the Python reader still works
            return None
        cached = FeatureManager._cache_load(inputFile) if FeatureManager.cache else None
        if cached is not None:
            return FeatureManager._lists_from_arrays(cached, mustHaveRelDoc)
        raw = (gzip.open(inputFile, "rb") if inputFile.endswith(".gz") else open(inputFile, "rb")).read()
        try:
            raw.decode("ascii")
        except UnicodeDecodeError:    # non-ASCII text: offsets into the bytes would not be offsets into the str
            return None
        p = N.letor_parse(raw)
        text = raw.decode("ascii")     # ASCII: byte offsets are character offsets
        n, X, mf = p["n"], p["X"], p["max_fid"]
        rows = list(X)                 # row views, created at C speed
        labels, last, slow = p["labels"].tolist(), p["last_fid"].tolist(), p["slow"].tolist()
        qo, do, lo = p["qid_off"].tolist(), p["desc_off"].tolist(), p["line_off"].tolist()
        qe = (p["qid_off"] + p["qid_len"]).tolist()
        de = (p["desc_off"] + p["desc_len"]).tolist()
        le = (p["line_off"] + p["line_len"]).tolist()
        if FeatureManager.cache and not any(slow):
            FeatureManager._cache_save(inputFile, dict(
                X=X, labels=p["labels"], last_fid=p["last_fid"], max_fid=np.int32(mf),
                qids=np.array([text[qo[i]:qe[i]] for i in range(n)], dtype="S"), descs=np.array([text[do[i]:de[i]] for i in range(n)], dtype="S")))
        from_parsed = DataPoint.from_parsed
        samples, rl, lastID, hasRel = [], [], "", False
        for i in range(n):
            if slow[i]:
                qp = DataPoint(text[lo[i]:le[i]])
            else:
                row = rows[i]
                qp = from_parsed(labels[i], text[qo[i]:qe[i]], text[do[i]:de[i]], row if last[i] == mf else row[:last[i] + 1])
            if lastID and lastID != qp.id:
                if not mustHaveRelDoc or hasRel:
                    samples.append(RankList(rl))
                rl, hasRel = [], False
            if qp.label > 0:
                hasRel = True
            lastID = qp.id
            rl.append(qp)
        if rl and (not mustHaveRelDoc or hasRel):
            samples.append(RankList(rl))
        return samples, n

    # ---- binary cache of a parsed LETOR file (the text of an MSLR-WEB30K fold is 5 GB) -------------------------------
    # `<file>.rlcache.npz` next to the text: the dense row matrix, labels, largest feature id per line, qid and description strings.
    # Written after a native parse without irregular lines, used when it is newer than the text.  Off by default (the reference has
    # no such file); `python -m ranklib_amd.evaluator -cache ...` or FeatureManager.cache = True turn it on.
    cache = False

    @staticmethod
    def _cache_path(inputFile):
        return inputFile + ".rlcache.npz"

    @staticmethod
    def _cache_save(inputFile, arrays):
        try:
            tmp = FeatureManager._cache_path(inputFile) + ".tmp.npz"
            st = os.stat(inputFile)
            np.savez(tmp, src_size=np.int64(st.st_size), src_mtime_ns=np.int64(st.st_mtime_ns), **arrays)
            os.replace(tmp, FeatureManager._cache_path(inputFile))
        except OSError as ex:          # read-only directory: the cache is an optimisation
            logger.info("no LETOR cache written: %s", ex)

    @staticmethod
    def _cache_load(inputFile):
        path = FeatureManager._cache_path(inputFile)
        try:
            st = os.stat(inputFile)
            with np.load(path) as z:
                # the cache names the text it was made from (size and mtime in ns): a file rewritten within the timestamp granularity, or
                # restored with an older mtime (cp -p, rsync -t, git checkout), never yields stale rows
                if int(z["src_size"]) != st.st_size or int(z["src_mtime_ns"]) != st.st_mtime_ns:
                    return None
                return {k: z[k] for k in ("X", "labels", "last_fid", "max_fid", "qids", "descs")}
        except (OSError, KeyError, ValueError):
            return None

    @staticmethod
    def _lists_from_arrays(a, mustHaveRelDoc):
        X, mf = a["X"], int(a["max_fid"])
        rows, labels, last = list(X), a["labels"].tolist(), a["last_fid"].tolist()
        qids, descs = [q.decode("ascii") for q in a["qids"]], [d.decode("ascii") for d in a["descs"]]
        from_parsed = DataPoint.from_parsed
        samples, rl, lastID, hasRel = [], [], "", False
        for i in range(len(labels)):
            row = rows[i]
            qp = from_parsed(labels[i], qids[i], descs[i], row if last[i] == mf else row[:last[i] + 1])
            if lastID and lastID != qp.id:
                if not mustHaveRelDoc or hasRel:
                    samples.append(RankList(rl))
                rl, hasRel = [], False
            if qp.label > 0:
                hasRel = True
            lastID = qp.id
            rl.append(qp)
        if rl and (not mustHaveRelDoc or hasRel):
            samples.append(RankList(rl))
        return samples, len(labels)

    @staticmethod
    def readFeature(featureDefFile):
        fids = []
        with smart_reader(featureDefFile) as f:
            for content in f:
                content = content.strip()
                if not content or content[0] == "#":
                    continue
                fids.append(int(content.split("\t")[0].strip()))
        return fids

    @staticmethod
    def getFeatureFromSampleVector(samples):
        if not samples:
            raise RankLibError("Error in FeatureManager::getFeatureFromSampleVector(): There are no training samples.")
        fc = max(rl.getFeatureCount() for rl in samples)
        return list(range(1, fc + 1))

    @staticmethod
    def prepareSplit(samples, percentTrain):           # features/FeatureManager.java:432-443 (no shuffling: file order)
        size = int(len(samples) * percentTrain)
        return [RankList(rl) for rl in samples[:size]], [RankList(rl) for rl in samples[size:]]

    @staticmethod
    def prepareCV(samples, nFold, tvs=-1.0):           # :349-413: contiguous folds, the last one takes the remainder
        size = len(samples) // nFold
        folds, start, total = [], 0, 0
        for _ in range(nFold):
            t = [start + i for i in range(size) if start + i < len(samples)]
            folds.append(t)
            total += len(t)
            start += size
        while total < len(samples):
            folds[-1].append(total)
            total += 1
        trainingData, validationData, testData = [], [], []
        for t in folds:
            ts = set(t)
            train = [RankList(samples[j]) for j in range(len(samples)) if j not in ts]
            test = [RankList(samples[j]) for j in range(len(samples)) if j in ts]
            vali = []
            if tvs > 0:                                # the LAST lists of the training part, taken back to front (:391-397)
                validationSize = int(len(train) * (1.0 - tvs))
                for _ in range(validationSize):
                    vali.append(train.pop())
            trainingData.append(train)
            testData.append(test)
            if tvs > 0:
                validationData.append(vali)
        return trainingData, validationData, testData
