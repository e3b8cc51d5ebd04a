import sys, numpy as np
sys.path.insert(0,'/root/repo')
from ranklib_amd import _native as N, synth
X, lab, qoff, qt = synth.make_shard(3770000, 136, "mslr", 0, 1)
g = N.Trainer(n_trees=20, n_leaves=31, device=0, flags=0)
g.set_train(X, lab, qoff); g.init()
g.boost_rounds_async(20); g.sync()
out = np.zeros(16, np.int64)
N.check(N.lib().rl_get_array(g.h, 99, out.ctypes.data, 128))
n = out[6]; m = out[13]
print("select calls", n, "avg slots %.2f" % (out[7] / n), "per-call us: A %.1f B %.1f C %.1f D %.1f E %.1f" % tuple(out[i] / n / 100.0 for i in (0,1,2,3,4)))
print("finish (non-root, last block) us: setup %.1f chunks %.1f mode+prefix %.1f main loop %.1f block_best %.1f publish+arrive %.1f" % tuple(out[i] / m / 100.0 for i in (14,8,9,10,11,12)))
