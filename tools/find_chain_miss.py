import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import numpy as np
from ranklib_amd import _native as N, synth
X, lab, qoff, Q = synth.make_shard(3_770_000, 136, 'mslr', 0, 1)
g = N.Trainer(n_trees=30, n_leaves=31); g.set_train(X, lab, qoff); g.init()
prev = 0
for r in range(30):
    t, tm, _, _ = g.boost_round()
    st = g.array('CHAIN_STATS')
    if st[1] + st[2] != prev:
        miss = g.array('CHAIN_MISS')
        tr = t.trimmed()
        leaves = [i for i in range(t.n_nodes) if tr['feature'][i] == -1]
        print('round', r, 'stats', st, 'miss lambda leaves', np.nonzero(miss[0])[0], 'miss weight leaves', np.nonzero(miss[1])[0], 'leaf counts', tr['count'][leaves])
        prev = st[1] + st[2]
print('final stats', g.array('CHAIN_STATS'))
