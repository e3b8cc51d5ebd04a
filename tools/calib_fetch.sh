#!/bin/bash
# usage (GPU box): tools/calib_fetch.sh <tag> -- FETCH_SIZE / WRITE_SIZE of the membench kernels (known bytes) in two separate PMC passes
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python $R/tools/calib_fetch.py > $R/gpurun_out/calib_${tag}_rates.json 2>/dev/null
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$ctr
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/cal_$ctr -o p -- python $R/tools/calib_fetch.py > $R/gpurun_out/calib_${tag}_$ctr.log 2>&1
done
python - <<PY > $R/gpurun_out/calib_${tag}.txt
import sqlite3, glob, json
rates = json.load(open('$R/gpurun_out/calib_${tag}_rates.json'))
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only) on rl_debug_membench kernels with KNOWN byte counts")
print("# rates without the profiler (HIP events, 5 launches after a warm-up):")
for k, v in rates.items():
    print("#   %-22s %10.1f GB/s  (%.0f bytes per launch, %.3f ms)" % (k, v["GBps"], v["alg_bytes"], v["avg_ms"]))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob('/tmp/cal_%s/*.db' % ctr)[0]
    con = sqlite3.connect(db)
    print("# %s per dispatch in launch order (KB): calib_fetch.py launches copy x6, read x6, write x6, then [fill_idx, gather32 x6] for strides 1, 2, 4, 16" % ctr)
    rows = list(con.execute("select kernel_name, dispatch_id, value from counters_collection where counter_name=? and kernel_name like '%k_mb_%' order by dispatch_id", (ctr,)))
    i = 0
    while i < len(rows):
        j = i
        while j < len(rows) and rows[j][0] == rows[i][0]:
            j += 1
        vals = [r[2] for r in rows[i:j]]
        print("%-60s n=%2d avg=%14.1f min=%14.1f max=%14.1f" % (rows[i][0][:60], len(vals), sum(vals) / len(vals), min(vals), max(vals)))
        i = j
PY
cat $R/gpurun_out/calib_${tag}.txt
