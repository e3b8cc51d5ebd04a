#!/bin/bash
# usage (on the GPU box, via gpurun): tools/ab_env.sh <out.txt> <rounds> <shape> VAR "v1 v2 .."     -- same-box A/B of one run-time knob, three interleaved repetitions
out=$1; rounds=$2; shape=$3; var=$4; vals=$5
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for v in $vals; do
    env $var=$v python $R/bench.py --plain --shape $shape --steps $rounds --warmup 5 2>/dev/null | grep '"metric"' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-6s %s=%-4s rep $rep  %8.2f rounds/s  %7.4f ms' % ('$shape', '$var', '$v', d['value'], d['ms_per_step']))"
  done
done | tee -a $out
