#!/bin/bash
# usage (GPU box): tools/gpu_pmc.sh <tag> [bench args]   -- two separate PMC passes (FETCH_SIZE, WRITE_SIZE) with kernel trace only
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_$ctr -o p -- python $R/bench.py --cpu-rounds 0 --no-timing "$@" > $R/gpurun_out/pmc_${tag}_$ctr.log 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pmc_$ctr/*.db')[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
print("# $ctr  (rocprofv3 --pmc $ctr --kernel-trace), columns:", cols)
q = "select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection where counter_name='$ctr' group by kernel_name order by 3 desc"
try:
    for r in con.execute(q):
        print("%-72s n=%5d avg=%14.1f min=%14.1f max=%14.1f" % (r[0][:72], r[1], r[2], r[3], r[4]))
except Exception as e:
    print("query failed:", e, tabs)
PY
done
