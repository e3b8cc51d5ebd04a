#!/bin/bash
# usage (GPU box): tools/gpu_pmc_infer.sh <tag> [docs]   -- kernel stats and SQ counters of the inference kernel (bench.py --workload infer)
# separate rocprofv3 passes: --kernel-trace --stats, then one --pmc pass per counter set (kernel trace only, never with other trace domains)
tag=$1; docs=${2:-4000000}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--workload infer --docs $docs --steps 3 --warmup 1 --cpu-rounds 0"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py $ARGS > $R/gpurun_out/prof_$tag.log 2>&1
python $R/tools/rocprof_summary.py $R/gpurun_out/prof_$tag/${tag}_results.db "$tag: rocprofv3 --kernel-trace --stats -- python bench.py $ARGS" > $R/gpurun_out/prof_${tag}_kernel_stats.txt
rm -f $R/gpurun_out/prof_$tag/*.db
out=$R/gpurun_out/prof_${tag}_pmc.txt
echo "# $tag: rocprofv3 --pmc <set> --kernel-trace -- python bench.py $ARGS  (one pass per set; averages per launch of k_model_eval_tiled)" > $out
for ctr in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  d=/tmp/pmc_i_$(echo $ctr | tr ' ' '_')
  rm -rf $d
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $d -o p -- python $R/bench.py $ARGS > /dev/null 2>&1
  python - >> $out <<PY
import sqlite3, glob
dbs = glob.glob('$d/**/*.db', recursive=True)
if dbs:
    con = sqlite3.connect(dbs[0])
    for name, ctr, n, avg in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name order by kernel_name"):
        if "k_model_eval" in name:
            print("%-40s %-22s launches %3d  avg per launch %18.1f" % (name[:40], ctr, n, avg))
else:
    print("(no counters for: $ctr)")
PY
done
cat $out
