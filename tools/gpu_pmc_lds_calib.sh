#!/bin/bash
# usage (GPU box): tools/gpu_pmc_lds_calib.sh  -- LDS bank-conflict counters of the CALIBRATION kernel (rl_debug_membench modes 4 / 5 / 7: 64 consecutive
# bins per wavefront -- the conflict-free pattern of a 64-bit atomic --, a random bin of 256 per lane, random + the count atomic), one rocprofv3 --pmc
# run per pattern (kernel trace only), next to the rate each pattern reaches without the profiler: do the conflicts bound the LDS atomics?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for mode in 4 5 7; do
  d=/tmp/pmc_ldscal_$mode
  rm -rf $d
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d $d -o p -- python -c "
import sys; sys.path.insert(0, '$R')
from ranklib_amd import _native as N
N.membench($mode, 4096, 1, 3)" > /dev/null 2>&1
  MODE=$mode python - <<PY
import sqlite3, glob, os
db = glob.glob('$d/**/*.db', recursive=True)[0]
con = sqlite3.connect(db)
vals = {}
for name, ctr, n, avg in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    if "k_mb_lds_atomic" in name: vals[ctr] = (n, avg)
a = vals.get("SQ_LDS_IDX_ACTIVE", (0, 0.0)); b = vals.get("SQ_LDS_BANK_CONFLICT", (0, 0.0))
print("pattern mode %s: launches %d  SQ_LDS_IDX_ACTIVE %.0f  SQ_LDS_BANK_CONFLICT %.0f  conflict share %.3f" % (os.environ["MODE"], a[0], a[1], b[1], b[1] / a[1] if a[1] else 0.0))
PY
done
python -c "
import sys; sys.path.insert(0, '$R')
from ranklib_amd import _native as N
for mode, name in ((4, 'consecutive bins'), (5, 'random bins'), (7, 'random bins + count')):
    ms, groups = N.membench(mode, 4096, 1, 10)
    print('%-22s %.3f ms per launch, %.2f atomic groups per CU and clock (256 CUs x 2.4 GHz)' % (name, ms, groups / (ms * 1e-3) / (256 * 2.4e9)))"
