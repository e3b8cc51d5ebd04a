#!/bin/bash
# usage (GPU box): tools/gpu_pmc_lds.sh <tag> [bench args]  -- LDS counters of the histogram kernels (kernel trace only, never with other trace domains):
# SQ_LDS_IDX_ACTIVE = all LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra cycles of bank / same-address conflicts (MI355X_MICROARCH.md, LDS),
# next to the kernels' durations: LDS_IDX_ACTIVE / (CUs x duration x clock) is how busy the LDS arrays are.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  d=/tmp/pmc_lds_$(echo $ctr | tr ' ' '_')
  rm -rf $d
  rocprofv3 --pmc $ctr --kernel-trace -d $d -o p -- python $R/bench.py --cpu-rounds 0 --no-timing --plain "$@" > /dev/null 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob('$d/**/*.db', recursive=True)[0]
con = sqlite3.connect(db)
for name, ctr, n, avg in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name order by kernel_name"):
    if "k_hist<" in name:
        print("%-64s %-22s launches %5d  avg per launch %16.1f" % (name[:64], ctr, n, avg))
PY
done
