"""create / train / destroy in a loop and watch the device memory in use: the library must give back everything it took
(python tools/leak_check.py [iterations])"""
import re
import subprocess
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
from ranklib_amd import _native as N  # noqa: E402
from ranklib_amd import synth  # noqa: E402


def used():
    txt = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True, timeout=60).stdout
    return int(re.search(r"VRAM Total Used Memory \(B\): (\d+)", txt).group(1))


X, lab, qoff = synth.make_dataset(300000, 40, "mslr", seed_offset=1)
Xv, lv, qv = synth.make_dataset(50000, 40, "mslr", seed_offset=2)
base = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    for flags, metric in ((0, "NDCG"), (N.RL_FLAG_JAVA_ORDER, "NDCG"), (0, "ERR"), (0, "MAP")):
        g = N.Trainer(n_trees=3, n_leaves=8, flags=flags, metric=metric, metric_k=0 if metric == "MAP" else 10)
        g.set_train(X, lab, qoff)
        g.set_validation(Xv, lv, qv)
        g.init()
        for _ in range(3):
            g.boost_round()
        g.finish()
        m = N.Model(g.model_text())
        m.predict_rows(np.concatenate([np.zeros((1000, 1), np.float32), X[:1000]], axis=1))
        m.close()
        g.close()
    u = used()
    if base is None:
        base = u
    print("iteration %d: %.1f MB in use (%+.1f MB against the first)" % (it, u / 1e6, (u - base) / 1e6), flush=True)
assert u - base < 64e6, "device memory grows"
print("no leak")
