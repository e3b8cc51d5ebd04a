"""GPU box: per-launch k_hist<false> durations (rocprofv3 trace) against the sizes of the accumulated (smaller) children"""
import os, sys, sqlite3, subprocess, json
import numpy as np
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R)
if len(sys.argv) > 1 and sys.argv[1] == 'run':
    from ranklib_amd import _native as N, synth
    X, lab, qoff, Q = synth.make_shard(3_770_000, 136, 'mslr', 0, 1)
    g = N.Trainer(n_trees=4, n_leaves=31); g.set_train(X, lab, qoff); g.init()
    for r in range(4):
        t, tm, _, _ = g.boost_round()
    tr = t.trimmed()
    sizes = sorted([int(min(tr['count'][tr['left'][i]], tr['count'][tr['right'][i]])) for i in range(t.n_nodes) if tr['feature'][i] != -1], reverse=True)
    print('SIZES', json.dumps(sizes))
else:
    os.chdir('/tmp'); os.environ['TMPDIR'] = '/tmp'
    out = subprocess.run(['rocprofv3', '--kernel-trace', '-d', '/tmp/hv', '-o', 'hv', '--', sys.executable, os.path.abspath(__file__), 'run'], capture_output=True, text=True)
    sizes = json.loads([l for l in out.stdout.splitlines() if l.startswith('SIZES')][0][6:])
    con = sqlite3.connect('/tmp/hv/hv_results.db')
    rows = list(con.execute("select (end-start)/1e3 from kernels where name like '%k_hist<false%' order by start"))
    d = sorted([r[0] for r in rows[-30:]], reverse=True)
    for s, t in zip(sizes, d):
        print("%9d docs  %8.1f us   %6.1f Gpairs/s" % (s, t, s * 136 / t / 1e3))
    for nm in ('k_part_scatter', 'k_part_count', 'k_hist_finish<false', 'k_select'):
        rows = [r[0] for r in con.execute("select (end-start)/1e3 from kernels where name like '%%%s%%' order by start" % nm)][-30:]
        print(nm, ' '.join('%.0f' % v for v in rows))
