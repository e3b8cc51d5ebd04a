#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_profile.sh <tag> [bench args...]
# runs bench.py under rocprofv3 --kernel-trace --stats and leaves gpurun_out/prof_<tag>/ + summary text
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py --cpu-rounds 0 --no-timing "$@" > $R/gpurun_out/prof_$tag.log 2>&1
grep '"metric"' $R/gpurun_out/prof_$tag.log | cut -c1-400
python $R/tools/rocprof_summary.py $R/gpurun_out/prof_$tag/${tag}_results.db "$tag: rocprofv3 --kernel-trace --stats -- python bench.py --cpu-rounds 0 --no-timing $*" > $R/gpurun_out/prof_${tag}_kernel_stats.txt
head -30 $R/gpurun_out/prof_${tag}_kernel_stats.txt | cut -c1-200
python $R/tools/rocprof_timeline.py $R/gpurun_out/prof_$tag/${tag}_results.db > $R/gpurun_out/prof_${tag}_timeline.txt
rm -f $R/gpurun_out/prof_$tag/*.db
