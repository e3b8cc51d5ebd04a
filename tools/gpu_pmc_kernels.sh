#!/bin/bash
# usage (GPU box): tools/gpu_pmc_kernels.sh <kernel name substring> "<counters of pass 1>" ["<counters of pass 2>" ...] [-- bench args]
# SQ counters of the kernels whose name contains the substring, one rocprofv3 --pmc pass per quoted counter set (kernel trace only, never with
# other trace domains).  SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked: s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stalls) + SQ_ACTIVE_INST_ANY, in
# quad-cycles (MI355X_MICROARCH.md, counters).
filt=$1; shift
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
[ "$1" = "--" ] && shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for ctr in "${sets[@]}"; do
  d=/tmp/pmc_k_$(echo $ctr | tr ' ' '_')
  rm -rf $d
  rocprofv3 --pmc $ctr --kernel-trace -d $d -o p -- python $R/bench.py --cpu-rounds 0 --no-timing --plain "$@" > /dev/null 2>&1
  FILT="$filt" python - <<PY
import sqlite3, glob, os
db = glob.glob('$d/**/*.db', recursive=True)[0]
con = sqlite3.connect(db)
for name, ctr, n, avg in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name order by kernel_name"):
    if os.environ["FILT"] in name:
        print("%-72s %-22s launches %5d  avg per launch %16.1f" % (name[:72], ctr, n, avg))
PY
done
